// Which fp16 MFMA shape does an MI355X sustain more FLOP/s on under its power limit?  A wavefront owns a 128 x 64 (or 64 x 64) output block
// and, per step of K = 32, reads its hi / lo fragments of both operands from LDS (random fp16 data, ds_read_b128) and issues the three
// partial products of the fp16-split arithmetic (hi*lo, lo*hi, hi*hi) either as v_mfma_f32_32x32x16_f16 (2 x K 16) or as
// v_mfma_f32_16x16x32_f16.  Same FLOPs, same LDS bytes, same MFMA cycles by the book -- what differs is the clock the chip holds.
//     hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_shape_probe tools/mfma_shape_probe.hip && /tmp/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// LDS image: 384 rows x 128 B (K = 32 of a pair-format slab: [hi 0..15 | lo 0..15 | hi 16..31 | lo 16..31]); plain linear reads with a row
// XOR swizzle so that ds_read_b128 groups do not collide (the exact fragment meaning is irrelevant here: only bytes and instructions count)
template <int SHAPE, int TM64>      // SHAPE 32: 32x32x16; 16: 16x16x32.  TM64: rows of the wavefront's block / 64 (2: 128 x 64, 1: 64 x 64)
__global__ __launch_bounds__(512, 2) void probe(const uint4* __restrict__ src, float* __restrict__ out, int iters, unsigned long long* __restrict__ clk) {
    extern __shared__ char smem[];
    for (int i = threadIdx.x; i < 384 * 128 / 16; i += blockDim.x) reinterpret_cast<uint4*>(smem)[i] = src[(blockIdx.x % 64) * 3072 + i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int ROWS = TM64 * 64;
    const int wm0 = (wave & 1) * ROWS, wn0 = 256 + (wave >> 1) % 2 * 64;
    unsigned long long t0 = 0, r0 = 0;
    if (threadIdx.x == 0) { r0 = __builtin_amdgcn_s_memrealtime(); t0 = __builtin_amdgcn_s_memtime(); }
    if (SHAPE == 32) {
        constexpr int TM = ROWS / 32, TN = 2;
        f32x16 acc[TM][TN];
        for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        const int lr = lane & 31, lh = lane >> 5;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {          // two K = 16 granules
                f16x8 a[TM][2], b[TN][2];
#pragma unroll
                for (int p = 0; p < 2; ++p) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[j][p] = *reinterpret_cast<const f16x8*>(smem + (wn0 + j * 32 + lr) * 128 + ((half * 4 + 2 * p + lh) ^ ((lr >> 1) & 7)) * 16);
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[i][p] = *reinterpret_cast<const f16x8*>(smem + ((wm0 + i * 32 + lr) % 256) * 128 + ((half * 4 + 2 * p + lh) ^ ((lr >> 1) & 7)) * 16);
                }
#pragma unroll
                for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[j][pr == 0], a[i][pr == 1], acc[i][j], 0, 0, 0);
            }
        }
        float s = 0.f;
        for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else {
        constexpr int TM = ROWS / 16, TN = 4;
        f32x4 acc[TM][TN];
        for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
        const int lr = lane & 15, lg = lane >> 4;           // row, K group (8 of the 32)
        for (int it = 0; it < iters; ++it) {
            f16x8 a[TM][2], b[TN][2];
            // hi fragment: K groups 0, 1 = granule 0's hi chunks, 2, 3 = granule 1's; lo fragment likewise
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int chunk = (lg >> 1) * 4 + 2 * p + (lg & 1);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j][p] = *reinterpret_cast<const f16x8*>(smem + (wn0 + j * 16 + lr) * 128 + (chunk ^ ((lr >> 1) & 7)) * 16);
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i][p] = *reinterpret_cast<const f16x8*>(smem + ((wm0 + i * 16 + lr) % 256) * 128 + (chunk ^ ((lr >> 1) & 7)) * 16);
            }
#pragma unroll
            for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j][pr == 0], a[i][pr == 1], acc[i][j], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int e = 0; e < 4; ++e) s += acc[i][j][e];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
        clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

template <typename K>
static void run(const char* name, K kern, int threads, int wgs, const uint4* src, float* out, unsigned long long* clk, int iters, double flop_per_wg_iter) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 384 * 128);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads), 384 * 128, 0, src, out, iters, clk);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 10;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads), 384 * 128, 0, src, out, iters, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    std::vector<unsigned long long> h(2 * wgs);
    hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, ghz = 0;
    for (int i = 0; i < wgs; ++i) { cyc += (double)h[2 * i]; ghz += (double)h[2 * i] / ((double)h[2 * i + 1] * 10.0); }
    printf("%-58s %8.3f ms  %8.1f TFLOP/s fp16 (%6.1f fp32-equivalent)  cycles/wg %10.0f  clock %.2f GHz\n", name, ms, flop_per_wg_iter * iters * wgs / (ms * 1e-3) / 1e12,
           flop_per_wg_iter * iters * wgs / (ms * 1e-3) / 1e12 / 3.0, cyc / wgs, ghz / wgs);
}

int main(int argc, char** argv) {
    const int iters = 3000;
    const bool zeros = argc > 1 && atoi(argv[1]) == 0;
    std::vector<unsigned short> h(64 * 3072 * 8);
    srand(1);
    for (auto& v : h) {
        // random fp16 bit patterns of moderate magnitude: sign, exponent 8..20 (biased), 10 mantissa bits
        v = zeros ? 0 : (unsigned short)(((rand() & 1) << 15) | ((8 + rand() % 12) << 10) | (rand() & 1023));
    }
    uint4* src; float* out; unsigned long long* clk;
    hipMalloc(&src, h.size() * 2); hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&clk, 4096 * 16);
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    printf("operands: %s\n", zeros ? "zeros" : "random fp16");
    // per workgroup and iteration (K = 32): waves x rows x 64 x 32 x 2 x 3 products
    for (int rep = 0; rep < 2; ++rep) {
        run("32x32x16, 8 waves of 128 x 64 (512 thr, 1 wg/CU)", probe<32, 2>, 512, 256, src, out, clk, iters, 8.0 * 128 * 64 * 32 * 2 * 3);
        run("16x16x32, 8 waves of 128 x 64 (512 thr, 1 wg/CU)", probe<16, 2>, 512, 256, src, out, clk, iters, 8.0 * 128 * 64 * 32 * 2 * 3);
        run("32x32x16, 8 waves of 64 x 64  (512 thr, 1 wg/CU)", probe<32, 1>, 512, 256, src, out, clk, iters, 8.0 * 64 * 64 * 32 * 2 * 3);
        run("16x16x32, 8 waves of 64 x 64  (512 thr, 1 wg/CU)", probe<16, 1>, 512, 256, src, out, clk, iters, 8.0 * 64 * 64 * 32 * 2 * 3);
        run("32x32x16, 4 waves of 128 x 64 (256 thr, 2 wg/CU)", probe<32, 2>, 256, 512, src, out, clk, iters, 4.0 * 128 * 64 * 32 * 2 * 3);
        run("16x16x32, 4 waves of 128 x 64 (256 thr, 2 wg/CU)", probe<16, 2>, 256, 512, src, out, clk, iters, 4.0 * 128 * 64 * 32 * 2 * 3);
    }
    return 0;
}
