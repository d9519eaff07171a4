// How fast can ONE workgroup per CU take data in on an MI355X?  Each workgroup streams `bytes_per_wg` from a buffer (shared by all: L2 /
// Infinity Cache resident, or private slices of a large one: HBM) with 16-byte loads, UNROLL loads in flight per lane, into registers
// (global_load_dwordx4) or into LDS (global_load_lds_dwordx4).  Prints GB/s per CU and chip-wide.
//     hipcc --offload-arch=gfx950 -O2 -o /tmp/cu_ingest_probe tools/cu_ingest_probe.hip && /tmp/cu_ingest_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int UNROLL>
__global__ __launch_bounds__(256) void reg_stream(const uint4* __restrict__ src, size_t wg_stride_vec, size_t vec_per_wg, uint4* __restrict__ sink) {
    const uint4* p = src + (size_t)blockIdx.x * wg_stride_vec;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t i = threadIdx.x; i + (UNROLL - 1) * 256 < vec_per_wg; i += (size_t)UNROLL * 256) {
        uint4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = p[i + (size_t)u * 256];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345679u) sink[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int UNROLL>
__global__ __launch_bounds__(256) void lds_stream(const char* __restrict__ src, size_t wg_stride, size_t bytes_per_wg, uint4* __restrict__ sink) {
    extern __shared__ char smem[];                       // UNROLL x 4 KB per round (256 lanes x 16 B)
    const char* p = src + (size_t)blockIdx.x * wg_stride;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (size_t off = 0; off + (size_t)UNROLL * 4096 <= bytes_per_wg; off += (size_t)UNROLL * 4096) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + off + (size_t)u * 4096 + wave * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(smem + u * 4096 + wave * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const uint4 v = reinterpret_cast<const uint4*>(smem)[threadIdx.x];
    if ((v.x ^ v.y) == 0x12345679u) sink[blockIdx.x * 256 + threadIdx.x] = v;
}

// the operand pattern of a K-contiguous GEMM tile: ROWS rows of ROWBYTES, taken in as K slabs of GRAN bytes per row (GRAN = 64: a slab of 16
// pair-format k, half a 128-byte line per row and instruction; 128: whole lines).  One LDS-DMA instruction of a wavefront covers 1024 / GRAN rows.
template <int GRAN, int INFLIGHT>
__global__ __launch_bounds__(256) void lds_rows(const char* __restrict__ src, size_t wg_stride, int rows, int rowbytes, int rowstride, uint4* __restrict__ sink, int reps = 16) {
    extern __shared__ char smem[];
    const char* p = src + (size_t)blockIdx.x * wg_stride;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int RPI = 1024 / GRAN, CH = GRAN / 16;       // rows per instruction, 16-byte chunks per row
    const int r_in = lane / CH, c = lane % CH;
    const int nslab = rowbytes / GRAN, npiece = rows / RPI / 4;      // pieces per wavefront and slab
    int q = 0;
    for (int rep = 0; rep < reps; ++rep)                   // (the tile again and again: a launch long enough that its ramp does not count)
    for (int s = 0; s < nslab; ++s) {
        for (int i = 0; i < npiece; ++i) {
            const int row = (wave + 4 * i) * RPI + r_in;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + (size_t)row * rowstride + (size_t)s * GRAN + c * 16),
                                             (__attribute__((address_space(3))) void*)(smem + (q % INFLIGHT) * 4096 + wave * 1024), 16, 0, 0);
            if (++q % INFLIGHT == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT / 2) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const uint4 v = reinterpret_cast<const uint4*>(smem)[threadIdx.x];
    if ((v.x ^ v.y) == 0x12345679u) sink[blockIdx.x * 256 + threadIdx.x] = v;
}

template <typename F>
static double time_ms(F f, int reps = 20) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    const size_t total = (size_t)2 << 30;                 // 2 GiB
    char* buf; uint4* sink;
    hipMalloc(&buf, total); hipMalloc(&sink, 1 << 24);
    hipMemset(buf, 1, total);
    const size_t per_wg = (size_t)4 << 20;                // every workgroup streams 4 MiB
    struct Case { const char* name; size_t stride; int wgs; } cases[] = {
        {"shared 2 MiB window (L2), 256 wgs", 0, 256}, {"private 4 MiB slices of 1 GiB (HBM), 256 wgs", per_wg, 256},
        {"shared window, 32 wgs", 0, 32}, {"shared window, 512 wgs (2 per CU)", 0, 512}};
    for (auto& c : cases) {
        const size_t bytes = c.stride ? per_wg : (size_t)2 << 20;      // shared: a 2 MiB window (fits an XCD L2) read twice
        const int passes = c.stride ? 1 : 2;
        auto report = [&](const char* kind, int unroll, double ms) {
            const double gb = (double)bytes * passes * c.wgs / 1e9;
            printf("%-46s %-8s %2d x 16 B in flight per lane: %7.1f GB/s per workgroup  %8.1f GB/s chip\n", c.name, kind, unroll, gb / c.wgs / (ms * 1e-3),
                   gb / (ms * 1e-3));
        };
#define RUN_REG(U) report("regs", U, time_ms([&] { for (int q = 0; q < passes; ++q) reg_stream<U><<<c.wgs, 256>>>((const uint4*)buf, c.stride / 16, bytes / 16, sink); }) )
#define RUN_LDS(U) report("lds-dma", U, time_ms([&] { for (int q = 0; q < passes; ++q) lds_stream<U><<<c.wgs, 256, U * 4096>>>(buf, c.stride, bytes, sink); }) )
        RUN_REG(1); RUN_REG(4); RUN_REG(8); RUN_REG(16);
        RUN_LDS(1); RUN_LDS(4); RUN_LDS(8); RUN_LDS(16);
    }
    // GEMM-tile pattern: 384 rows (a 256 x 128 tile's operands) of 2,048 bytes (K = 512 in the pair format) at a row stride of 2,048 bytes
    // (a [rows][512] fp32 tensor) or padded; all workgroups the same rows (L2) or rows of their own (a 1 GiB matrix: HBM)
    for (int own = 0; own < 2; ++own) {
        for (int rowstride : {2048, 2048 + 128, 8192}) {
            const int rows = 384, rowbytes = 2048, wgs = 256;
            const size_t stride = own ? (size_t)rows * rowstride + 4096 : 0;
            auto rep = [&](int gran, int inflight, double ms) {
                const double gb = 16.0 * rows * rowbytes * wgs / 1e9;
                printf("tile rows (%s), row stride %5d B, slabs of %3d B per row, %2d instructions in flight per wavefront: %7.1f GB/s per workgroup (%5.1f us per 768 KB tile)\n",
                       own ? "own rows: HBM" : "shared rows: L2", rowstride, gran, inflight, gb / wgs / (ms * 1e-3), ms * 1e3 / 16);
            };
#define RUN_ROWS(G, F) rep(G, F, time_ms([&] { lds_rows<G, F><<<wgs, 256, F * 4096>>>(buf, stride, rows, rowbytes, rowstride, sink); }))
            RUN_ROWS(64, 12); RUN_ROWS(128, 12);
        }
    }
    // harness check: the same kernel on CONTIGUOUS bytes (a "row" is one granule), and whole 1,024-byte row halves per instruction
    {
        const int wgs = 256;
        auto rep2 = [&](const char* what, int rows, int rowbytes, double ms) {
            const double gb = 16.0 * rows * rowbytes * wgs / 1e9;
            printf("%-70s %7.1f GB/s per workgroup\n", what, gb / wgs / (ms * 1e-3));
        };
        rep2("contiguous 768 KB as 12,288 'rows' of 64 B (shared: L2)", 12288, 64, time_ms([&] { lds_rows<64, 12><<<wgs, 256, 12 * 4096>>>(buf, 0, 12288, 64, 64, sink); }));
        rep2("contiguous 768 KB as 6,144 'rows' of 128 B (shared: L2)", 6144, 128, time_ms([&] { lds_rows<128, 12><<<wgs, 256, 12 * 4096>>>(buf, 0, 6144, 128, 128, sink); }));
        rep2("384 rows of 2,048 B, 1,024 B of ONE row per instruction (shared: L2)", 384 * 2, 1024, time_ms([&] { lds_rows<1024, 12><<<wgs, 256, 12 * 4096>>>(buf, 0, 768, 1024, 1024, sink); }));
        rep2("384 rows of 2,048 B in slabs of 512 B per row (shared: L2)", 384, 2048, time_ms([&] { lds_rows<512, 12><<<wgs, 256, 12 * 4096>>>(buf, 0, 384, 2048, 2048, sink); }));
        rep2("3,072 rows of 256 B in slabs of 64 B per row, row stride 256 (shared: L2)", 3072, 256, time_ms([&] { lds_rows<64, 12><<<wgs, 256, 12 * 4096>>>(buf, 0, 3072, 256, 256, sink); }));
        rep2("1,536 rows of 512 B in slabs of 64 B per row, row stride 512 (shared: L2)", 1536, 512, time_ms([&] { lds_rows<64, 12><<<wgs, 256, 12 * 4096>>>(buf, 0, 1536, 512, 512, sink); }));
    }
    return 0;
}
