// The fused writer-head layer (C ABI 11: grappa_writer_head_fwd, include/grappa_hip.h): ONE kernel per transformer layer of a writer head
//
//     x1 = LN(x); qkv = x1 W_in^T + b_in; a = MHA(q, k, v) over the s tokens of a tuple; x2 = drop(a W_o^T + b_o) + x1;
//     x3 = LN(x2); u = ELU(x3 W_1^T + b_1); out = drop(u W_2^T + b_2) + x3
//
// (reference models/network_utils.py:112-133 DottedAttWithMLP with :44-54 FeedForwardLayer; the stack is models/perm_equiv_transformer.py:121-151).
// A workgroup owns a tile of 64 token rows = all s tokens of 32 / 21 / 16 tuples and walks the whole layer on it: the 512-wide
// activations never leave the CU.  bf16 storage configuration (BASELINE configs[2]).
//
//  * LDS (138 KB): image A [64][512] bf16 (x1, later x3), a staging area for the q, k, v of TWO heads [64][384] and their attention
//    output [64][128]; the ELU output u [64][512] overlays the staging area once the attention is done.  Rows are 32 bytes longer than
//    their data (conflict-free ds_read_b128 of the fragments, see WL_LDA).
//  * every product is computed TRANSPOSED, out^T = W act^T, on v_mfma_f32_16x16x32_bf16: the weight is the A operand, read straight from
//    HBM / L2 into registers from a copy packed in fragment order (grappa_writer_pack_weight: one contiguous KB per wave-instruction, every
//    weight byte is loaded once per workgroup -- the eight wavefronts own disjoint 64-feature slices); the activation is the B operand,
//    read from the LDS image.  An accumulator then holds 4 CONSECUTIVE features of one token per lane: results go back into the next
//    image with one 8-byte ds_write per fragment, row statistics are sums over registers and 2 lane exchanges.
//  * the out-projection is accumulated head pair by head pair (K = 128 at a time) in registers while the q, k, v of the next pair are
//    being produced, so the full attention output never exists.
//  * training (save pointers non-NULL): every tensor the unfused backward pass reads is written as a by-product, never re-read.
#include "common.h"

namespace {

typedef __bf16 wl_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wl_bf16x4 __attribute__((ext_vector_type(4)));
typedef float wl_f32x4 __attribute__((ext_vector_type(4)));

// diagnostic build only (tools/writer_layer_lab.sh -DWL_LAB_STAMP): wavefront 0 of the first 2,048 workgroups of the forward kernel leaves the clock at
// its phase boundaries in a device array that grappa_debug_writer_stamps() copies out.  No stamp executes in the shipped library.
#ifdef WL_LAB_STAMP
constexpr int WL_STAMP_WORDS = 48, WL_STAMP_WGS = 2048;
__device__ unsigned long long g_wl_stamps[WL_STAMP_WGS * WL_STAMP_WORDS];
#define WL_STAMP(k) do { if (tid == 0 && blockIdx.x < WL_STAMP_WGS) g_wl_stamps[blockIdx.x * WL_STAMP_WORDS + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WL_STAMP(k) do { } while (0)
#endif
constexpr int WL_F = 512, WL_ROWS = 64, WL_THREADS = 512;
// row strides = 32 bytes more than the data, i.e. 32 (mod 256): the 16 token rows x 4 k-chunks of a ds_read_b128 fragment read then touch
// every bank once per lane group (MI355X LDS: groups {0-3, 12-15, 20-27}, ... : 8 rows on the even 16-byte slots, the other 8 on the odd ones)
constexpr int WL_LDA = 528;      // elements per row of a full-width image (1,056 bytes)
constexpr int WL_LDQ = 400;      // q | k | v of two heads: 384 + 16
constexpr int WL_LDT = 144;      // attention output of two heads: 128 + 16
constexpr int WL_OFF_A = 0;
constexpr int WL_OFF_Q = WL_ROWS * WL_LDA * 2;
constexpr int WL_OFF_T = WL_OFF_Q + WL_ROWS * WL_LDQ * 2;
constexpr int WL_OFF_B = WL_OFF_Q;
constexpr int WL_OFF_RED = WL_OFF_T + WL_ROWS * WL_LDT * 2;
constexpr int WL_OFF_PAR = WL_OFF_RED + 2 * 8 * WL_ROWS * 4;      // forward kernel: b_in (1536) | b_o | b_1 | b_2 | nf gamma | nf beta (512 each) as fp32
constexpr int WL_PAR_BIN = 0, WL_PAR_BO = 1536, WL_PAR_B1 = 2048, WL_PAR_B2 = 2560, WL_PAR_GF = 3072, WL_PAR_BF = 3584, WL_PAR_N = 4096;
constexpr int WL_OFF_IDX = WL_OFF_PAR + WL_PAR_N * 4;             // gather mode: the table row of each of the tile's 64 token rows (or -1)
constexpr int WL_SMEM = WL_OFF_IDX + WL_ROWS * 4;
static_assert(WL_OFF_B + WL_ROWS * WL_LDA * 2 <= WL_OFF_RED, "the u image must fit into the staging area");
static_assert(WL_SMEM <= 160 * 1024, "LDS of one CU");

__device__ __forceinline__ void wl_unpack8(const uint4& u, float (&v)[8]) {
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
    v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
    v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 wl_pack8(const float (&v)[8]) {
    wl_bf16x8 h;
#pragma unroll
    for (int e = 0; e < 8; ++e) h[e] = (__bf16)v[e];
    return __builtin_bit_cast(uint4, h);
}
__device__ __forceinline__ uint2 wl_pack4(const float (&v)[4]) {
    wl_bf16x4 h;
#pragma unroll
    for (int e = 0; e < 4; ++e) h[e] = (__bf16)v[e];
    return __builtin_bit_cast(uint2, h);
}
__device__ __forceinline__ void wl_unpack4(const uint2& u, float (&v)[4]) {
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
}
__device__ __forceinline__ float wl_round_bf16(float v) {
    const __bf16 h = (__bf16)v;
    return (float)h;
}

// acc[i][mb] += W[n-block i](:, ks0 ... ks0 + KSN) act^T: pa[i] = the lane's first fragment of n-block i (consecutive k-steps are 64 uint4
// apart), bimg = the lane's first fragment in the LDS image (row lr, k = 8 lq), rows `ldb` bytes apart.  PD k-steps of weights in flight.
// the first PD k-steps of a product's weight fragments -> registers.  Issued EARLY -- in front of the epilogue / barrier / attention that precedes
// the product -- so that their L2 round trip is over when the product starts (one exposed round trip per product was ~15 % of a tile's time)
template <int NB, int PD>
__device__ __forceinline__ void wl_ring_fill(const uint4* const (&pa)[NB], uint4 (&ring)[PD][NB]) {
#pragma unroll
    for (int d = 0; d < PD; ++d)
#pragma unroll
        for (int i = 0; i < NB; ++i) ring[d][i] = pa[i][d * 64];
}

// acc[i][mb] += W[n-block i](:, KSN k-steps) act^T with the ring already filled
template <int NB, int KSN, int PD>
__device__ __forceinline__ void wl_product_run(const uint4* const (&pa)[NB], uint4 (&ring)[PD][NB], const char* bimg, const int ldb, wl_f32x4 (&acc)[NB][4]) {
    static_assert(KSN % PD == 0, "whole rings");
    // every k-step is a scheduling region of its own (sched_barrier): left alone, hipcc sinks the refill loads of the ring to just in front of their
    // use -- one exposed L2 round trip per fragment -- instead of keeping PD k-steps of weights in flight
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int ks0 = 0; ks0 < KSN; ks0 += PD) {
#pragma unroll
        for (int d = 0; d < PD; ++d) {
            const int ks = ks0 + d;
            wl_bf16x8 b[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) b[mb] = __builtin_bit_cast(wl_bf16x8, *reinterpret_cast<const uint4*>(bimg + mb * 16 * ldb + ks * 64));
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const wl_bf16x8 a = __builtin_bit_cast(wl_bf16x8, ring[d][i]);
#ifdef WL_LAB_NO_MFMA      // tools/writer_layer_lab.sh: a build without the matrix instructions (what the rest of the kernel costs)
                {
                    const uint4 b0 = __builtin_bit_cast(uint4, b[0]), b1 = __builtin_bit_cast(uint4, b[1]), b2 = __builtin_bit_cast(uint4, b[2]), b3 = __builtin_bit_cast(uint4, b[3]);
                    asm volatile("" ::"v"(ring[d][i].x ^ ring[d][i].y ^ ring[d][i].z ^ ring[d][i].w), "v"(b0.x ^ b0.y ^ b0.z ^ b0.w), "v"(b1.x ^ b1.y ^ b1.z ^ b1.w),
                                 "v"(b2.x ^ b2.y ^ b2.z ^ b2.w), "v"(b3.x ^ b3.y ^ b3.z ^ b3.w));
                }
                (void)a;
#else
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) acc[i][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[mb], acc[i][mb], 0, 0, 0);
#endif
            }
            if (ks + PD < KSN) {
#pragma unroll
#ifdef WL_LAB_NO_WLOAD     // lab: every weight fragment from the same KB (served by the CU's L1): what streaming the weights from L2 costs
                for (int i = 0; i < NB; ++i) ring[d][i] = pa[i][0];
#else
                for (int i = 0; i < NB; ++i) ring[d][i] = pa[i][(ks + PD) * 64];
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// acc[i][mb] += W[n-block i](:, ks0 ... ks0 + KSN) act^T: pa[i] = the lane's first fragment of n-block i (consecutive k-steps are 64 uint4
// apart), bimg = the lane's first fragment in the LDS image (row lr, k = 8 lq), rows `ldb` bytes apart.  PD k-steps of weights in flight.
template <int NB, int KSN, int PD>
__device__ __forceinline__ void wl_product(const uint4* const (&pa)[NB], const char* bimg, const int ldb, wl_f32x4 (&acc)[NB][4]) {
    uint4 ring[PD][NB];
    wl_ring_fill<NB, PD>(pa, ring);
    wl_product_run<NB, KSN, PD>(pa, ring, bimg, ldb, acc);
}

// The dropout decision of csrc/common.h (grappa_keep: a 32-bit finaliser of the 64-bit element index, keyed by the seed) for elements
// row * 512 + n + e, e = 0 .. E - 1, with everything that does not depend on e hoisted -- the same bits for 3 instead of 5 quarter-rate 32-bit
// multiplies per element (the hash was a sixth of a training tile's cycles): idx = row * 512 + c has no carry from c < 512 into its high word,
// and (uint32) idx * K = lo(row * 512) * K + c * K (mod 2^32).
struct WlDropRow { uint32_t a, t; };          // per token row: lo(row * 512) * K1 + lo(seed), hi(row * 512) * K2
__device__ __forceinline__ WlDropRow wl_drop_row(uint64_t seed, long row) {
    const uint64_t base = (uint64_t)(row < 0 ? 0 : row) * WL_F;
    return WlDropRow{(uint32_t)base * 0x9E3779B1u + (uint32_t)seed, (uint32_t)(base >> 32) * 0x85EBCA77u};
}
__device__ __forceinline__ uint32_t wl_drop_threshold(float p) { return (uint32_t)ceilf(p * 16777216.0f); }      // keep iff (h >> 8) >= p * 2^24
template <int E>
__device__ __forceinline__ void wl_dropout(float (&v)[E], const WlDropRow& r, uint32_t col_k1, uint32_t seed_hi, uint32_t thr, float scale) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
        uint32_t h = (r.a + col_k1 + (uint32_t)e * 0x9E3779B1u) ^ r.t;      // col_k1 = n * K1 of the first element's column
        h ^= h >> 16;
        h *= 0x85EBCA6Bu;
        h ^= h >> 13;
        h *= 0xC2B2AE35u;
        h ^= h >> 16;
        h += seed_hi;
        h ^= h >> 15;
        h *= 0x2C1B3C6Du;
        h ^= h >> 12;
        v[e] = (h >> 8) >= thr ? v[e] * scale : 0.0f;
    }
}

// workgroup barrier for LDS hand-offs only: waits for this wavefront's LDS operations, NOT for its global loads / stores -- __syncthreads() makes
// hipcc drain vmcnt(0) first, i.e. every barrier would wait for the weight fragments requested ahead (wl_ring_fill) and for the by-product stores
__device__ __forceinline__ void wl_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ELU for a result that is rounded to bf16: exp(x) - 1 from v_exp_f32 below -1/16, its cubic Taylor polynomial above (relative error < 2e-5,
// far inside a bf16 step) -- expm1f() costs ~50 instructions per element, a tenth of the tile's time in the FF1 epilogue
__device__ __forceinline__ float wl_elu_bf16(float x) {
    const float t = x * (1.0f + x * (0.5f + x * 0.16666667f));
    return x > 0.0f ? x : (x > -0.0625f ? t : __expf(x) - 1.0f);
}

template <int NB>
__device__ __forceinline__ void wl_zero(wl_f32x4 (&acc)[NB][4]) {
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[i][mb] = wl_f32x4{0.f, 0.f, 0.f, 0.f};
}

// sums over aligned groups of 8 / 16 lanes on the DPP data path (no LDS crossbar round trips); every lane gets the sum
template <int CTRL>
__device__ __forceinline__ float wl_dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wl_sum8(float v) {
    v = wl_dpp_add<0xB1>(v);        // quad_perm [1, 0, 3, 2]
    v = wl_dpp_add<0x4E>(v);        // quad_perm [2, 3, 0, 1]
    return wl_dpp_add<0x141>(v);    // row_half_mirror
}
__device__ __forceinline__ float wl_sum64(float v) {
    v = wl_dpp_add<0x140>(wl_sum8(v));      // row_mirror: the 16 lanes of a row
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
// sum over the 16 lanes of a DPP row (the 16 tokens of an accumulator block); every lane gets the sum
__device__ __forceinline__ float wl_sum16(float v) { return wl_dpp_add<0x140>(wl_sum8(v)); }
// sum over the four lanes that hold the same token (l, l ^ 16, l ^ 32, l ^ 48)
__device__ __forceinline__ float wl_quad_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// GATHER (the first layer of a head whose LayerNorm and q | k | v were computed once per (atom, position) TABLE row, ops.ProjFirstLayerFn): the
// normalised rows x1 and q | k | v are not computed but gathered from those tables through gather_idx[t * s + pos] (the row of
// grappa_tuple_gather_fwd); everything behind them is the same kernel.
template <int S, bool GATHER>
__global__ __launch_bounds__(WL_THREADS) void writer_layer_fwd_bf16_kernel(const grappa_writer_layer_desc d) {
    constexpr int TT = WL_ROWS / S;                 // tuples per tile: 32, 21, 16
    extern __shared__ char smem[];
    uint16_t* imgA = reinterpret_cast<uint16_t*>(smem + WL_OFF_A);
    uint16_t* imgQ = reinterpret_cast<uint16_t*>(smem + WL_OFF_Q);
    uint16_t* imgT = reinterpret_cast<uint16_t*>(smem + WL_OFF_T);
    uint16_t* imgB = reinterpret_cast<uint16_t*>(smem + WL_OFF_B);
    float* red = reinterpret_cast<float*>(smem + WL_OFF_RED);          // [2][8 wavefronts][64 rows]
    // the epilogues' per-feature vectors live in LDS: a global load of them behind the weight fragments requested ahead would wait for all of those
    // (vector-memory results return in order)
    float* par = reinterpret_cast<float*>(smem + WL_OFF_PAR);
    int* trow = reinterpret_cast<int*>(smem + WL_OFF_IDX);

    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lr = l & 15, lq = l >> 4;
    const int T = d.T;
    const int t0 = blockIdx.x * TT;
    const int ntup = min(TT, T - t0);
    const uint16_t* __restrict__ x = reinterpret_cast<const uint16_t*>(d.x);
    uint16_t* __restrict__ out = reinterpret_cast<uint16_t*>(d.out);
    uint16_t* __restrict__ sv_x1 = reinterpret_cast<uint16_t*>(d.save_x1);
    uint16_t* __restrict__ sv_qkv = reinterpret_cast<uint16_t*>(d.save_qkv);
    uint16_t* __restrict__ sv_att = reinterpret_cast<uint16_t*>(d.save_att);
    uint16_t* __restrict__ sv_x2 = reinterpret_cast<uint16_t*>(d.save_x2);
    uint16_t* __restrict__ sv_x3 = reinterpret_cast<uint16_t*>(d.save_x3);
    uint16_t* __restrict__ sv_u = reinterpret_cast<uint16_t*>(d.save_u);
    const float drop_p = d.drop_p;
    const float drop_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;

    WL_STAMP(0);
    {
        const float* src[6] = {d.b_in, d.b_o, d.b1, d.b2, d.nf_gamma, d.nf_beta};
        const int beg[7] = {WL_PAR_BIN, WL_PAR_BO, WL_PAR_B1, WL_PAR_B2, WL_PAR_GF, WL_PAR_BF, WL_PAR_N};
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int i4 = (pass * WL_THREADS + threadIdx.x) * 4;      // 1,024 float4 = 4,096 floats
            int a = 0;
#pragma unroll
            for (int q = 1; q < 6; ++q) a += i4 >= beg[q] ? 1 : 0;
            const float* sp = a == 0 ? src[0] : (a == 1 ? src[1] : (a == 2 ? src[2] : (a == 3 ? src[3] : (a == 4 ? src[4] : src[5]))));
            const int b0 = a == 0 ? beg[0] : (a == 1 ? beg[1] : (a == 2 ? beg[2] : (a == 3 ? beg[3] : (a == 4 ? beg[4] : beg[5]))));
            *reinterpret_cast<float4*>(par + i4) = *reinterpret_cast<const float4*>(sp + (i4 - b0));
        }
    }
    if constexpr (GATHER) {
        // ---- phase 0 (gather): the table rows of the tile -> LDS; x1 = their normalised rows, as they stand -> image A
        const int* __restrict__ gidx = d.gather_idx;
        const uint16_t* __restrict__ x1t = reinterpret_cast<const uint16_t*>(d.x1_tab);
        if (tid < WL_ROWS) {
            const int p = tid / TT, j = tid - p * TT;
            trow[tid] = (p < S && j < ntup) ? gidx[(size_t)(t0 + j) * S + p] : -1;
        }
        const int rr = l >> 4, c = l & 15;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int r = w * 8 + ps * 4 + rr, p = r / TT, j = r - p * TT;
            const int tr = (p < S && j < ntup) ? gidx[(size_t)(t0 + j) * S + p] : -1;
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) {
                const int col = (c + 16 * kq) * 8;
                *reinterpret_cast<uint4*>(imgA + r * WL_LDA + col) = tr >= 0 ? *reinterpret_cast<const uint4*>(x1t + (size_t)tr * WL_F + col) : make_uint4(0u, 0u, 0u, 0u);
            }
        }
    } else
    // ---- phase 0: x1 = LN(x) -> image A; wavefront w owns tile rows 8 w ... 8 w + 7, FOUR rows at a time: a row per 16 lanes (its reductions
    // stay inside a DPP row), a lane the 16-byte chunks c, c + 16, c + 32, c + 48 of its row
    {
        const int rr = l >> 4, c = l & 15;
        uint4 raw[2][4];
        long grow2[2];
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int r = w * 8 + ps * 4 + rr, p = r / TT, j = r - p * TT;
            const bool ok = p < S && j < ntup;
            grow2[ps] = ok ? (long)p * T + t0 + j : -1;
#pragma unroll
            for (int kq = 0; kq < 4; ++kq)
                raw[ps][kq] = ok ? *reinterpret_cast<const uint4*>(x + (size_t)grow2[ps] * WL_F + (c + 16 * kq) * 8) : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int r = w * 8 + ps * 4 + rr;
            float v[4][8];
            float sm = 0.f;
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) {
                wl_unpack8(raw[ps][kq], v[kq]);
#pragma unroll
                for (int e = 0; e < 8; e += 2) sm += v[kq][e] + v[kq][e + 1];
            }
            const float mean = wl_sum16(sm) / (float)WL_F;
            float q = 0.f;
#pragma unroll
            for (int kq = 0; kq < 4; ++kq)
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const float a0 = v[kq][e] - mean, a1 = v[kq][e + 1] - mean;
                    q += a0 * a0 + a1 * a1;
                }
            const float rstd = 1.0f / sqrtf(wl_sum16(q) / (float)WL_F + 1e-5f);
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) {
                const int col = (c + 16 * kq) * 8;
                const float4 g0 = *reinterpret_cast<const float4*>(d.n1_gamma + col), g1 = *reinterpret_cast<const float4*>(d.n1_gamma + col + 4);
                const float4 b0 = *reinterpret_cast<const float4*>(d.n1_beta + col), b1 = *reinterpret_cast<const float4*>(d.n1_beta + col + 4);
                const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
                float y[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = grappa_ln_apply(v[kq][e], mean, rstd, g[e], b[e]);
                const uint4 pk = wl_pack8(y);
                *reinterpret_cast<uint4*>(imgA + r * WL_LDA + col) = pk;
                if (sv_x1 && grow2[ps] >= 0) *reinterpret_cast<uint4*>(sv_x1 + (size_t)grow2[ps] * WL_F + col) = pk;
            }
            if (d.save_mean1 && c == 0 && grow2[ps] >= 0) {
                d.save_mean1[grow2[ps]] = mean;
                d.save_rstd1[grow2[ps]] = rstd;
            }
        }
    }
    // the four tokens (one per 16-row block) this lane holds in every accumulator: global row or -1
    long grow[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        const int r = mb * 16 + lr, p = r / TT, j = r - p * TT;
        grow[mb] = (p < S && j < ntup) ? (long)p * T + t0 + j : -1;
    }
    const uint64_t seed1 = grappa_salted(d.seed1, d.drop_salt), seed2 = grappa_salted(d.seed2, d.drop_salt);
    const uint32_t drop_thr = wl_drop_threshold(drop_p);
    WlDropRow drow1[4], drow2[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        drow1[mb] = wl_drop_row(seed1, grow[mb]);
        drow2[mb] = wl_drop_row(seed2, grow[mb]);
    }
    const uint4* __restrict__ wq_in = reinterpret_cast<const uint4*>(d.w_in_pk);
    const uint4* __restrict__ wq_o = reinterpret_cast<const uint4*>(d.w_o_pk);
    const uint4* __restrict__ wq_1 = reinterpret_cast<const uint4*>(d.w1_pk);
    const uint4* __restrict__ wq_2 = reinterpret_cast<const uint4*>(d.w2_pk);
    const char* fragA = smem + WL_OFF_A + (lr * WL_LDA + 8 * lq) * 2;
    const char* fragT = smem + WL_OFF_T + (lr * WL_LDT + 8 * lq) * 2;
    const char* fragB = smem + WL_OFF_B + (lr * WL_LDA + 8 * lq) * 2;
    WL_STAMP(1);
    uint4 ring_c[4][3];                                  // the first q | k | v weights of head pair 0 travel under the barrier
    const uint4* pc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) pc[i] = wq_in + ((size_t)((((3 * w + i) >> 3) * 32 + ((3 * w + i) & 7)) * 16)) * 64 + l;
    if constexpr (!GATHER) wl_ring_fill<3, 4>(pc, ring_c);
    wl_barrier();
    WL_STAMP(2);

    // ---- phase 1: head pair by head pair: q, k, v of two heads -> staging; attention -> staging; x2 accumulator += a W_o^T (K = 128).
    // The weight fragments of every product are requested one phase early (wl_ring_fill).
    wl_f32x4 oacc[4][4];
    wl_zero<4>(oacc);
#pragma unroll 1
    for (int hp = 0; hp < 4; ++hp) {
        uint4 ring_p[2][4];
        const uint4* pp[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) pp[i] = wq_o + ((size_t)(4 * w + i) * 16 + hp * 4) * 64 + l;
        if constexpr (GATHER) {
            // the pair's q | k | v of every token row from its TABLE row: 64 rows x 3 segments of 256 bytes
            const uint16_t* __restrict__ qt = reinterpret_cast<const uint16_t*>(d.qkv_tab);
#pragma unroll
            for (int pass = 0; pass < 6; ++pass) {
                const int idx = pass * WL_THREADS + tid, r = idx / 48, c16 = idx - r * 48, seg = c16 >> 4, off = c16 & 15;
                const int tr = trow[r];
                *reinterpret_cast<uint4*>(imgQ + r * WL_LDQ + c16 * 8) =
                    tr >= 0 ? *reinterpret_cast<const uint4*>(qt + (size_t)tr * (3 * WL_F) + seg * WL_F + hp * 128 + off * 8) : make_uint4(0u, 0u, 0u, 0u);
            }
        } else {
            wl_f32x4 acc[3][4];
            wl_zero<3>(acc);
            int nbw[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) nbw[i] = ((3 * w + i) >> 3) * 32 + hp * 8 + ((3 * w + i) & 7);      // the chunk's 16-feature block in W_in's 1536 rows
            wl_product_run<3, 16, 4>(pc, ring_c, fragA, WL_LDA * 2, acc);
            WL_STAMP(3 + 6 * hp);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int n = nbw[i] * 16 + 4 * lq;
                const float4 bi = *reinterpret_cast<const float4*>(par + WL_PAR_BIN + n);
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    const float v[4] = {acc[i][mb][0] + bi.x, acc[i][mb][1] + bi.y, acc[i][mb][2] + bi.z, acc[i][mb][3] + bi.w};
                    *reinterpret_cast<uint2*>(imgQ + (mb * 16 + lr) * WL_LDQ + (3 * w + i) * 16 + 4 * lq) = wl_pack4(v);
                }
            }
        }
        wl_ring_fill<4, 2>(pp, ring_p);                  // the out-projection's first fragments travel while the attention runs
        WL_STAMP(4 + 6 * hp);
        wl_barrier();
        WL_STAMP(5 + 6 * hp);
        if (!GATHER && sv_qkv) {
            // the pair's q | k | v rows -> global in whole 256-byte runs (by-product for the backward pass): 64 rows x 48 chunks of 16 bytes
#pragma unroll
            for (int pass = 0; pass < 6; ++pass) {
                const int idx = pass * WL_THREADS + tid, r = idx / 48, c16 = idx - r * 48, seg = c16 >> 4, off = c16 & 15;
                const int p = r / TT, j = r - p * TT;
                if (p < S && j < ntup)
                    *reinterpret_cast<uint4*>(sv_qkv + ((size_t)p * T + t0 + j) * (3 * WL_F) + seg * WL_F + hp * 128 + off * 8) =
                        *reinterpret_cast<const uint4*>(imgQ + r * WL_LDQ + c16 * 8);
            }
        }
        // attention of (tuple j, head 2 hp + hh): 16 lanes, 4 features each (the arithmetic of csrc/tuples.hip seqattn_fwd_kernel_e)
#pragma unroll 1
        for (int pass = 0; pass < (2 * TT + 31) / 32; ++pass) {
            const int sub = tid & 15, pair = pass * 32 + (tid >> 4), hh = pair & 1, j = pair >> 1;
#ifdef WL_LAB_NO_ATT       // lab: without the attention arithmetic
            if (false) {
#else
            if (j < ntup) {
#endif
                float q[S][4], k[S][4], v[S][4];
#pragma unroll
                for (int i = 0; i < S; ++i) {
                    const uint16_t* row = imgQ + (i * TT + j) * WL_LDQ + hh * 64 + sub * 4;
                    wl_unpack4(*reinterpret_cast<const uint2*>(row), q[i]);
                    wl_unpack4(*reinterpret_cast<const uint2*>(row + 128), k[i]);
                    wl_unpack4(*reinterpret_cast<const uint2*>(row + 256), v[i]);
                }
#pragma unroll
                for (int i = 0; i < S; ++i) {
                    float sc[S], mx = -INFINITY;
#pragma unroll
                    for (int jj = 0; jj < S; ++jj) {
                        const float dt = (q[i][0] * k[jj][0] + q[i][1] * k[jj][1]) + (q[i][2] * k[jj][2] + q[i][3] * k[jj][3]);
                        sc[jj] = wl_sum16(dt) * 0.125f;
                        mx = fmaxf(mx, sc[jj]);
                    }
                    float den = 0.f;
#pragma unroll
                    for (int jj = 0; jj < S; ++jj) {
                        sc[jj] = expf(sc[jj] - mx);
                        den += sc[jj];
                    }
                    const float inv = 1.0f / den;
                    float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int jj = 0; jj < S; ++jj) {
                        const float pw = sc[jj] * inv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] += pw * v[jj][e];
                    }
                    const uint2 pk = wl_pack4(o);
                    *reinterpret_cast<uint2*>(imgT + (i * TT + j) * WL_LDT + hh * 64 + sub * 4) = pk;
                    if (sv_att) *reinterpret_cast<uint2*>(sv_att + ((size_t)i * T + t0 + j) * WL_F + (2 * hp + hh) * 64 + sub * 4) = pk;
                }
            }
        }
        WL_STAMP(6 + 6 * hp);
        wl_barrier();
        WL_STAMP(7 + 6 * hp);
        if (!GATHER && hp < 3) {                         // the next pair's q | k | v weights travel while the out-projection runs
#pragma unroll
            for (int i = 0; i < 3; ++i) pc[i] = wq_in + ((size_t)((((3 * w + i) >> 3) * 32 + (hp + 1) * 8 + ((3 * w + i) & 7)) * 16)) * 64 + l;
            wl_ring_fill<3, 4>(pc, ring_c);
        }
        wl_product_run<4, 4, 2>(pp, ring_p, fragT, WL_LDT * 2, oacc);
        WL_STAMP(8 + 6 * hp);
    }
    uint4 ring_f[4][4];                                  // FF1's first fragments travel under the LayerNorm (requested once the x2 values exist:
    const uint4* pf[4];                                  // before that the out-projection's accumulators are still live)
#pragma unroll
    for (int i = 0; i < 4; ++i) pf[i] = wq_1 + ((size_t)(4 * w + i) * 16) * 64 + l;

    // ---- phase 2: x2 = drop(acc + b_o) + x1 (rounded to bf16 like the stored tensor); x3 = LN(x2) -> image A
    {
        float x2v[4][4][4];
        float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = (4 * w + i) * 16 + 4 * lq;
            const float4 bo = *reinterpret_cast<const float4*>(par + WL_PAR_BO + n);
            const float bb[4] = {bo.x, bo.y, bo.z, bo.w};
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                float x1v[4];
                wl_unpack4(*reinterpret_cast<const uint2*>(imgA + (mb * 16 + lr) * WL_LDA + n), x1v);
                float v4[4] = {oacc[i][mb][0] + bb[0], oacc[i][mb][1] + bb[1], oacc[i][mb][2] + bb[2], oacc[i][mb][3] + bb[3]};
                if (drop_p > 0.f) wl_dropout<4>(v4, drow1[mb], (uint32_t)n * 0x9E3779B1u, (uint32_t)(seed1 >> 32), drop_thr, drop_scale);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = wl_round_bf16(v4[e] + x1v[e]);
                    x2v[i][mb][e] = v;
                    part[mb] += v;
                }
                if (sv_x2 && !d.x2_tiled && grow[mb] >= 0) *reinterpret_cast<uint2*>(sv_x2 + (size_t)grow[mb] * WL_F + n) = wl_pack4(x2v[i][mb]);
            }
        }
        wl_ring_fill<4, 4>(pf, ring_f);
        if (sv_x2 && d.x2_tiled) {
            // x2 for the fused backward kernel only (its LayerNorm backward reads it per accumulator element): the 16 quads of a lane behind one
            // another, 128 contiguous bytes per lane and tile -- whole lines for both kernels instead of 16 x 32-byte pieces per instruction
            uint4* xt = reinterpret_cast<uint4*>(sv_x2 + ((size_t)blockIdx.x * WL_THREADS + tid) * 64);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint2 a = grow[2 * h] >= 0 ? wl_pack4(x2v[i][2 * h]) : make_uint2(0u, 0u), c2 = grow[2 * h + 1] >= 0 ? wl_pack4(x2v[i][2 * h + 1]) : make_uint2(0u, 0u);
                    xt[i * 2 + h] = make_uint4(a.x, a.y, c2.x, c2.y);
                }
        }
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            part[mb] = wl_quad_sum(part[mb]);
            if (lq == 0) red[w * WL_ROWS + mb * 16 + lr] = part[mb];
        }
        wl_barrier();
        float mean[4], rstd[4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            float s = 0.f;
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) s += red[ww * WL_ROWS + mb * 16 + lr];
            mean[mb] = s / (float)WL_F;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = x2v[i][mb][e] - mean[mb];
                    q += a * a;
                }
            q = wl_quad_sum(q);
            if (lq == 0) red[(8 + w) * WL_ROWS + mb * 16 + lr] = q;
        }
        wl_barrier();
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            float q = 0.f;
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) q += red[(8 + ww) * WL_ROWS + mb * 16 + lr];
            rstd[mb] = 1.0f / sqrtf(q / (float)WL_F + 1e-5f);
            if (d.save_meanf && w == 0 && lq == 0 && grow[mb] >= 0) {
                d.save_meanf[grow[mb]] = mean[mb];
                d.save_rstdf[grow[mb]] = rstd[mb];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = (4 * w + i) * 16 + 4 * lq;
            const float4 g4 = *reinterpret_cast<const float4*>(par + WL_PAR_GF + n), b4 = *reinterpret_cast<const float4*>(par + WL_PAR_BF + n);
            const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                float y[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = grappa_ln_apply(x2v[i][mb][e], mean[mb], rstd[mb], gg[e], bb[e]);
                const uint2 pk = wl_pack4(y);
                *reinterpret_cast<uint2*>(imgA + (mb * 16 + lr) * WL_LDA + n) = pk;
            }
        }
    }
    wl_barrier();
    // a finished image -> global in whole rows (by-products for the backward pass): 64 rows x 64 chunks of 16 bytes, 8 per thread
    auto save_image = [&](const uint16_t* img, uint16_t* __restrict__ dst) {
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int idx = pass * WL_THREADS + tid, r = idx >> 6, c = idx & 63;
            const int p = r / TT, j = r - p * TT;
            if (p < S && j < ntup) *reinterpret_cast<uint4*>(dst + ((size_t)p * T + t0 + j) * WL_F + c * 8) = *reinterpret_cast<const uint4*>(img + r * WL_LDA + c * 8);
        }
    };
    WL_STAMP(27);
    if (sv_x3) save_image(imgA, sv_x3);

    // ---- phase 3: u = ELU(x3 W_1^T + b_1) -> image B
    uint4 ring_g[4][4];
    const uint4* pg[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) pg[i] = wq_2 + ((size_t)(4 * w + i) * 16) * 64 + l;
    {
        wl_f32x4 acc[4][4];
        wl_zero<4>(acc);
        wl_product_run<4, 16, 4>(pf, ring_f, fragA, WL_LDA * 2, acc);
        WL_STAMP(28);
        wl_ring_fill<4, 4>(pg, ring_g);                  // FF2's first fragments travel under the epilogue and the barrier
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = (4 * w + i) * 16 + 4 * lq;
            const float4 b4 = *reinterpret_cast<const float4*>(par + WL_PAR_B1 + n);
            const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                float y[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = wl_elu_bf16(acc[i][mb][e] + bb[e]);
                const uint2 pk = wl_pack4(y);
                *reinterpret_cast<uint2*>(imgB + (mb * 16 + lr) * WL_LDA + n) = pk;
            }
        }
    }
    wl_barrier();
    WL_STAMP(29);
    if (sv_u) save_image(imgB, sv_u);

    // ---- phase 4: out = drop(u W_2^T + b_2) + x3
    {
        wl_f32x4 acc[4][4];
        wl_zero<4>(acc);
        wl_product_run<4, 16, 4>(pg, ring_g, fragB, WL_LDA * 2, acc);
        WL_STAMP(30);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = (4 * w + i) * 16 + 4 * lq;
            const float4 b4 = *reinterpret_cast<const float4*>(par + WL_PAR_B2 + n);
            const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                if (grow[mb] < 0) continue;
                float x3v[4], y[4];
                wl_unpack4(*reinterpret_cast<const uint2*>(imgA + (mb * 16 + lr) * WL_LDA + n), x3v);
                float v4[4] = {acc[i][mb][0] + bb[0], acc[i][mb][1] + bb[1], acc[i][mb][2] + bb[2], acc[i][mb][3] + bb[3]};
                if (drop_p > 0.f) wl_dropout<4>(v4, drow2[mb], (uint32_t)n * 0x9E3779B1u, (uint32_t)(seed2 >> 32), drop_thr, drop_scale);
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = v4[e] + x3v[e];
                *reinterpret_cast<uint2*>(out + (size_t)grow[mb] * WL_F + n) = wl_pack4(y);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WL_STAMP(31);
}

// ---- the backward pass of the layer: the input-gradient chain on a tile of 64 token rows; the four weight gradients are ordinary grouped products
// over the operands this kernel writes as by-products (dz2, dz1, dzo, dqkv) and the activations the forward saved (u, x3, att, x1).
//   dz2 = dropmask2(dout)                      dz1 = (dz2 W_2) * ELU'(u)                 dx3 = dz1 W_1 + dout
//   dx2 = LN'(dx3; x2, nf)                     dzo = dropmask1(dx2)                      datt = dzo W_o
//   dqkv = attention'(qkv, datt)               dx1 = dqkv W_in + dx2                     dx  = LN'(dx1; x, n1)
// every tensor the unfused kernels store is rounded to bf16 at the same place.  Products as in the forward kernel, with the TRANSPOSED weights
// packed (grappa_writer_pack_weight transpose = 1): out^T = W^T act^T.  LDS: image A (dz2, then dzo, then the staging of one head pair's dq | dk | dv),
// image B (dz1, then datt).  LayerNorm parameter gradients: per-tile partial sums [tile][dgamma | dbeta][512], reduced by the caller.
// GATHER (see the forward kernel): q | k | v come from the table rows, the chain ends behind the attention -- dqkv (token rows) and the skip branch's
// gradient dx2 (`dx`) are this kernel's outputs, the caller sums them into the table rows (grappa_tuple_gather_bwd) and runs the table-level
// products and the first LayerNorm's backward there.
template <int S, bool GATHER>
__global__ __launch_bounds__(WL_THREADS) void writer_layer_bwd_bf16_kernel(const grappa_writer_layer_bwd_desc d) {
    constexpr int TT = WL_ROWS / S;
    extern __shared__ char smem[];
    uint16_t* imgA = reinterpret_cast<uint16_t*>(smem + WL_OFF_A);
    uint16_t* imgQ = reinterpret_cast<uint16_t*>(smem + WL_OFF_A);          // staging of a head pair's dq | dk | dv: inside image A (dzo is dead by then)
    uint16_t* imgB = reinterpret_cast<uint16_t*>(smem + WL_OFF_B);
    float* red = reinterpret_cast<float*>(smem + WL_OFF_RED);              // [2][8 wavefronts][64 rows]
    float* par = reinterpret_cast<float*>(smem + WL_OFF_PAR);              // nf gamma (512) | n1 gamma (512): see the forward kernel
    int* trow = reinterpret_cast<int*>(smem + WL_OFF_IDX);
    if (threadIdx.x < (GATHER ? 128 : 256)) {
        const int i4 = threadIdx.x * 4;
        *reinterpret_cast<float4*>(par + i4) = *reinterpret_cast<const float4*>((i4 < WL_F ? d.nf_gamma : d.n1_gamma) + (i4 & (WL_F - 1)));
    }
    if constexpr (GATHER) {
        if (threadIdx.x < WL_ROWS) {
            const int r = threadIdx.x, p = r / (WL_ROWS / S), j = r - p * (WL_ROWS / S);
            trow[r] = (p < S && j < min(WL_ROWS / S, d.T - (int)blockIdx.x * (WL_ROWS / S))) ? d.gather_idx[(size_t)(blockIdx.x * (WL_ROWS / S) + j) * S + p] : -1;
        }
    }

    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lr = l & 15, lq = l >> 4;
    const int T = d.T;
    const int t0 = blockIdx.x * TT;
    const int ntup = min(TT, T - t0);
    const uint16_t* __restrict__ dout = reinterpret_cast<const uint16_t*>(d.dout);
    const uint16_t* __restrict__ xin = reinterpret_cast<const uint16_t*>(d.x);
    const uint16_t* __restrict__ qkv = reinterpret_cast<const uint16_t*>(d.qkv);
    const uint16_t* __restrict__ x2s = reinterpret_cast<const uint16_t*>(d.x2);
    const uint16_t* __restrict__ us = reinterpret_cast<const uint16_t*>(d.u);
    uint16_t* __restrict__ dx = reinterpret_cast<uint16_t*>(d.dx);
    uint16_t* __restrict__ o_dz2 = reinterpret_cast<uint16_t*>(d.dz2);
    uint16_t* __restrict__ o_dz1 = reinterpret_cast<uint16_t*>(d.dz1);
    uint16_t* __restrict__ o_dzo = reinterpret_cast<uint16_t*>(d.dzo);
    uint16_t* __restrict__ o_dqkv = reinterpret_cast<uint16_t*>(d.dqkv);
    const float drop_p = d.drop_p;
    const float drop_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    const uint64_t seed1 = grappa_salted(d.seed1, d.drop_salt), seed2 = grappa_salted(d.seed2, d.drop_salt);

    const uint4* __restrict__ wq_in = reinterpret_cast<const uint4*>(d.w_in_tpk);
    const uint4* __restrict__ wq_o = reinterpret_cast<const uint4*>(d.w_o_tpk);
    const uint4* __restrict__ wq_1 = reinterpret_cast<const uint4*>(d.w1_tpk);
    const uint4* __restrict__ wq_2 = reinterpret_cast<const uint4*>(d.w2_tpk);
    // (the weight fragments of every product are requested one phase early, as in the forward kernel)
    uint4 ring_a[4][4];
    const uint4* pa2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) pa2[i] = wq_2 + ((size_t)(4 * w + i) * 16) * 64 + l;
    wl_ring_fill<4, 4>(pa2, ring_a);
    WL_STAMP(0);
    // ---- phase 0: dz2 = dropout mask of the forward's last dropout applied to dout -> image A (+ global: operand of dW_2)
    // (every tensor this kernel reads or writes per ELEMENT of an accumulator -- u, dout, x, dz1, dzo -- passes through an LDS image in whole rows:
    //  fragment-shaped global accesses, 16 rows x 32 bytes per instruction, cost a third of the tile's time in the first version)
    {
        uint4 rawd[8], rawu[8];
        long gr8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = w * 8 + i, p = r / TT, j = r - p * TT;
            gr8[i] = (p < S && j < ntup) ? (long)p * T + t0 + j : -1;
            rawd[i] = gr8[i] >= 0 ? *reinterpret_cast<const uint4*>(dout + (size_t)gr8[i] * WL_F + 8 * l) : make_uint4(0u, 0u, 0u, 0u);
            rawu[i] = gr8[i] >= 0 ? *reinterpret_cast<const uint4*>(us + (size_t)gr8[i] * WL_F + 8 * l) : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = w * 8 + i;
            uint4 pk = make_uint4(0u, 0u, 0u, 0u);
            if (gr8[i] >= 0) {
                float v[8];
                wl_unpack8(rawd[i], v);
                if (drop_p > 0.f) wl_dropout<8>(v, wl_drop_row(seed2, gr8[i]), (uint32_t)(8 * l) * 0x9E3779B1u, (uint32_t)(seed2 >> 32), wl_drop_threshold(drop_p), drop_scale);
                pk = wl_pack8(v);
                *reinterpret_cast<uint4*>(o_dz2 + (size_t)gr8[i] * WL_F + 8 * l) = pk;
            }
            *reinterpret_cast<uint4*>(imgA + r * WL_LDA + 8 * l) = pk;
            *reinterpret_cast<uint4*>(imgB + r * WL_LDA + 8 * l) = rawu[i];          // u: read back per element by phase 1's epilogue, which overwrites it with dz1
        }
    }
    long grow[4];
    float mean_f[4], rstd_f[4], mean_1[4], rstd_1[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        const int r = mb * 16 + lr, p = r / TT, j = r - p * TT;
        grow[mb] = (p < S && j < ntup) ? (long)p * T + t0 + j : -1;
        const bool ok = grow[mb] >= 0;
        mean_f[mb] = ok ? d.meanf[grow[mb]] : 0.f;
        rstd_f[mb] = ok ? d.rstdf[grow[mb]] : 0.f;
        mean_1[mb] = (!GATHER && ok) ? d.mean1[grow[mb]] : 0.f;
        rstd_1[mb] = (!GATHER && ok) ? d.rstd1[grow[mb]] : 0.f;
    }
    const char* fragA = smem + WL_OFF_A + (lr * WL_LDA + 8 * lq) * 2;
    const char* fragB = smem + WL_OFF_B + (lr * WL_LDA + 8 * lq) * 2;
    const char* fragQ = smem + WL_OFF_A + (lr * WL_LDQ + 8 * lq) * 2;
    auto save_image = [&](const uint16_t* img, uint16_t* __restrict__ dst) {           // a finished image -> global in whole rows
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int idx = pass * WL_THREADS + tid, r = idx >> 6, c = idx & 63;
            const int p = r / TT, j = r - p * TT;
            if (p < S && j < ntup) *reinterpret_cast<uint4*>(dst + ((size_t)p * T + t0 + j) * WL_F + c * 8) = *reinterpret_cast<const uint4*>(img + r * WL_LDA + c * 8);
        }
    };
    // whole rows of a global tensor -> an image (zeros beyond the tile's tuples), in two halves (request, LDS writes)
    auto fetch_rows = [&](const uint16_t* __restrict__ src, uint4 (&t)[8]) {
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int idx = pass * WL_THREADS + tid, r = idx >> 6, c = idx & 63;
            const int p = r / TT, j = r - p * TT;
            t[pass] = (p < S && j < ntup) ? *reinterpret_cast<const uint4*>(src + ((size_t)p * T + t0 + j) * WL_F + c * 8) : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto place_rows = [&](const uint4 (&t)[8], uint16_t* img) {
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int idx = pass * WL_THREADS + tid, r = idx >> 6, c = idx & 63;
            *reinterpret_cast<uint4*>(img + r * WL_LDA + c * 8) = t[pass];
        }
    };
    float* part_f = d.lnf_part + (size_t)blockIdx.x * 2 * WL_F;
    float* part_1 = d.ln1_part + (size_t)blockIdx.x * 2 * WL_F;
    wl_barrier();

    WL_STAMP(1);
    // ---- phase 1: dz1 = (dz2 W_2) * ELU'(u) -> image B (+ global: operand of dW_1)
    {
        wl_f32x4 acc[4][4];
        wl_zero<4>(acc);
        wl_product_run<4, 16, 4>(pa2, ring_a, fragA, WL_LDA * 2, acc);
        WL_STAMP(2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = (4 * w + i) * 16 + 4 * lq;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                float uv[4], y[4];
                wl_unpack4(*reinterpret_cast<const uint2*>(imgB + (mb * 16 + lr) * WL_LDA + n), uv);
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = acc[i][mb][e] * grappa_elu_grad_from_out(uv[e]);
                const uint2 pk = wl_pack4(y);
                *reinterpret_cast<uint2*>(imgB + (mb * 16 + lr) * WL_LDA + n) = pk;
            }
        }
    }

    WL_STAMP(3);
    // ---- phase 2: dx3 = dz1 W_1 + dout; dx2 = LN'(dx3; x2); dzo = dropout mask of the forward's first dropout applied to dx2 -> image A
    uint2 dx2p[4][4];                     // dx2 as stored (bf16): the skip branch into dx1, phase 5
    {
        uint2 dop[4][4], x2p[4][4];
        if (d.x2_tiled) {
            // x2 as the forward kernel left it for THIS kernel: per tile, the 16 accumulator quads of a lane behind one another (128 bytes per lane)
            const uint4* xt = reinterpret_cast<const uint4*>(x2s + ((size_t)blockIdx.x * WL_THREADS + tid) * 64);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint4 t4 = xt[i * 2 + h];
                    x2p[i][2 * h] = make_uint2(t4.x, t4.y);
                    x2p[i][2 * h + 1] = make_uint2(t4.z, t4.w);
                }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
                    x2p[i][mb] = grow[mb] >= 0 ? *reinterpret_cast<const uint2*>(x2s + (size_t)grow[mb] * WL_F + (4 * w + i) * 16 + 4 * lq) : make_uint2(0u, 0u);
        }
        wl_f32x4 acc[4][4];
        wl_zero<4>(acc);
        const uint4* pa[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) pa[i] = wq_1 + ((size_t)(4 * w + i) * 16) * 64 + l;
        uint4 ring_b[4][4];
        wl_ring_fill<4, 4>(pa, ring_b);
        wl_barrier();                                    // dz1 is complete in image B; image A (dz2) is dead
        WL_STAMP(4);
        save_image(imgB, o_dz1);                         // operand of dW_1
        {
            uint4 drows[8];                              // the skip branch's dout, per element below
            fetch_rows(dout, drows);
            place_rows(drows, imgA);
        }
        wl_product_run<4, 16, 4>(pa, ring_b, fragB, WL_LDA * 2, acc);
        wl_barrier();
        WL_STAMP(5);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) dop[i][mb] = *reinterpret_cast<const uint2*>(imgA + (mb * 16 + lr) * WL_LDA + (4 * w + i) * 16 + 4 * lq);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
                if (grow[mb] < 0) x2p[i][mb] = make_uint2(0u, 0u);
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = (4 * w + i) * 16 + 4 * lq;
            const float4 g4 = *reinterpret_cast<const float4*>(par + n);
            const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
            float dgam[4] = {0.f, 0.f, 0.f, 0.f}, dbet[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                float dov[4], xv[4], y[4];
                wl_unpack4(dop[i][mb], dov);
                wl_unpack4(x2p[i][mb], xv);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    y[e] = wl_round_bf16(acc[i][mb][e] + dov[e]);                  // dx3 as the unfused product stores it
                    const float xh = (xv[e] - mean_f[mb]) * rstd_f[mb], g = y[e] * gg[e];
                    s1[mb] += g;
                    s2[mb] += g * xh;
                    dgam[e] += y[e] * xh;
                    dbet[e] += y[e];
                }
                dop[i][mb] = wl_pack4(y);                                          // (dout is done: the registers now hold dx3)
            }
            // LayerNorm parameter gradients of this tile: sum over its 64 tokens = the 4 blocks (above) and the 16 lanes of a row
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dgam[e] = wl_sum16(dgam[e]);
                dbet[e] = wl_sum16(dbet[e]);
            }
            if (lr == 0) {
                *reinterpret_cast<float4*>(part_f + n) = make_float4(dgam[0], dgam[1], dgam[2], dgam[3]);
                *reinterpret_cast<float4*>(part_f + WL_F + n) = make_float4(dbet[0], dbet[1], dbet[2], dbet[3]);
            }
        }
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            s1[mb] = wl_quad_sum(s1[mb]);
            s2[mb] = wl_quad_sum(s2[mb]);
            if (lq == 0) {
                red[w * WL_ROWS + mb * 16 + lr] = s1[mb];
                red[(8 + w) * WL_ROWS + mb * 16 + lr] = s2[mb];
            }
        }
        wl_barrier();
        float m1[4], m2[4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) {
                a += red[ww * WL_ROWS + mb * 16 + lr];
                b += red[(8 + ww) * WL_ROWS + mb * 16 + lr];
            }
            m1[mb] = a / (float)WL_F;
            m2[mb] = b / (float)WL_F;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = (4 * w + i) * 16 + 4 * lq;
            const float4 g4 = *reinterpret_cast<const float4*>(par + n);
            const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                float dyv[4], xv[4], o[4], z[4];
                wl_unpack4(dop[i][mb], dyv);
                wl_unpack4(x2p[i][mb], xv);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xh = (xv[e] - mean_f[mb]) * rstd_f[mb];
                    o[e] = wl_round_bf16(rstd_f[mb] * (dyv[e] * gg[e] - m1[mb] - xh * m2[mb]));      // dx2 as stored
                    z[e] = o[e];
                }
                if (drop_p > 0.f) wl_dropout<4>(z, wl_drop_row(seed1, grow[mb]), (uint32_t)n * 0x9E3779B1u, (uint32_t)(seed1 >> 32), wl_drop_threshold(drop_p), drop_scale);
                dx2p[i][mb] = wl_pack4(o);
                const uint2 pk = wl_pack4(z);
                *reinterpret_cast<uint2*>(imgA + (mb * 16 + lr) * WL_LDA + n) = pk;
            }
        }
    }

    WL_STAMP(6);
    // ---- phase 3: datt = dzo W_o -> image B
    {
        wl_f32x4 acc[4][4];
        wl_zero<4>(acc);
        const uint4* pa[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) pa[i] = wq_o + ((size_t)(4 * w + i) * 16) * 64 + l;
        uint4 ring[4][4];
        wl_ring_fill<4, 4>(pa, ring);
        wl_barrier();
        WL_STAMP(7);
        save_image(imgA, o_dzo);                         // operand of dW_o
        wl_product_run<4, 16, 4>(pa, ring, fragA, WL_LDA * 2, acc);
        WL_STAMP(8);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = (4 * w + i) * 16 + 4 * lq;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const float y[4] = {acc[i][mb][0], acc[i][mb][1], acc[i][mb][2], acc[i][mb][3]};
                *reinterpret_cast<uint2*>(imgB + (mb * 16 + lr) * WL_LDA + n) = wl_pack4(y);
            }
        }
    }
    wl_barrier();

    WL_STAMP(9);
    // ---- phase 4: head pair by head pair: dq | dk | dv of two heads -> staging (+ global: operand of dW_in); dx1 accumulator += dqkv W_in (K = 384)
    wl_f32x4 xacc[4][4];
    wl_zero<4>(xacc);
#pragma unroll 1
    for (int hp = 0; hp < 4; ++hp) {
        // the backward of the attention of (tuple j, head 2 hp + hh): 16 lanes, 4 features each (the arithmetic of csrc/tuples.hip seqattn_bwd_kernel_e)
#pragma unroll 1
        for (int pass = 0; pass < (2 * TT + 31) / 32; ++pass) {
            const int sub = tid & 15, pair = pass * 32 + (tid >> 4), hh = pair & 1, j = pair >> 1;
            if (j < ntup) {
                float q[S][4], k[S][4], v[S][4], go[S][4], dq[S][4], dk[S][4], dv[S][4];
#pragma unroll
                for (int i = 0; i < S; ++i) {
                    const uint16_t* row = qkv + (GATHER ? (size_t)trow[i * TT + j] : (size_t)i * T + t0 + j) * (3 * WL_F) + (2 * hp + hh) * 64 + sub * 4;
                    wl_unpack4(*reinterpret_cast<const uint2*>(row), q[i]);
                    wl_unpack4(*reinterpret_cast<const uint2*>(row + WL_F), k[i]);
                    wl_unpack4(*reinterpret_cast<const uint2*>(row + 2 * WL_F), v[i]);
                    wl_unpack4(*reinterpret_cast<const uint2*>(imgB + (i * TT + j) * WL_LDA + (2 * hp + hh) * 64 + sub * 4), go[i]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dq[i][e] = dk[i][e] = dv[i][e] = 0.f;
                }
#pragma unroll
                for (int i = 0; i < S; ++i) {
                    float pr[S], dp[S], mx = -INFINITY;
#pragma unroll
                    for (int jj = 0; jj < S; ++jj) {
                        const float dt = (q[i][0] * k[jj][0] + q[i][1] * k[jj][1]) + (q[i][2] * k[jj][2] + q[i][3] * k[jj][3]);
                        pr[jj] = wl_sum16(dt) * 0.125f;
                        mx = fmaxf(mx, pr[jj]);
                    }
                    float den = 0.f;
#pragma unroll
                    for (int jj = 0; jj < S; ++jj) {
                        pr[jj] = expf(pr[jj] - mx);
                        den += pr[jj];
                    }
                    const float inv = 1.0f / den;
                    float dsum = 0.f;
#pragma unroll
                    for (int jj = 0; jj < S; ++jj) {
                        pr[jj] *= inv;
                        const float dt = (go[i][0] * v[jj][0] + go[i][1] * v[jj][1]) + (go[i][2] * v[jj][2] + go[i][3] * v[jj][3]);
                        dp[jj] = wl_sum16(dt);
                        dsum += pr[jj] * dp[jj];
                    }
#pragma unroll
                    for (int jj = 0; jj < S; ++jj) {
                        const float ds = pr[jj] * (dp[jj] - dsum) * 0.125f;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            dq[i][e] += ds * k[jj][e];
                            dk[jj][e] += ds * q[i][e];
                            dv[jj][e] += pr[jj] * go[i][e];
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < S; ++i) {
                    const uint2 pq = wl_pack4(dq[i]), pk = wl_pack4(dk[i]), pv = wl_pack4(dv[i]);
                    if constexpr (!GATHER) {
                        uint16_t* st = imgQ + (i * TT + j) * WL_LDQ + hh * 64 + sub * 4;
                        *reinterpret_cast<uint2*>(st) = pq;
                        *reinterpret_cast<uint2*>(st + 128) = pk;
                        *reinterpret_cast<uint2*>(st + 256) = pv;
                    }
                    uint16_t* gp = o_dqkv + ((size_t)i * T + t0 + j) * (3 * WL_F) + (2 * hp + hh) * 64 + sub * 4;
                    *reinterpret_cast<uint2*>(gp) = pq;
                    *reinterpret_cast<uint2*>(gp + WL_F) = pk;
                    *reinterpret_cast<uint2*>(gp + 2 * WL_F) = pv;
                }
            }
        }
        // dx1^T += W_in^T (:, the pair's q | k | v rows) dqkv^T: three runs of 4 k-steps (rows hp 128 ... of each of W_in's three blocks of 512)
        if constexpr (!GATHER) {
            const uint4* pq[3][4];
#pragma unroll
            for (int seg = 0; seg < 3; ++seg)
#pragma unroll
                for (int i = 0; i < 4; ++i) pq[seg][i] = wq_in + ((size_t)(4 * w + i) * 48 + seg * 16 + hp * 4) * 64 + l;
            uint4 r0[2][4], r1[2][4];
            WL_STAMP(10 + 3 * hp);
            wl_ring_fill<4, 2>(pq[0], r0);
            wl_barrier();
            WL_STAMP(11 + 3 * hp);
            wl_ring_fill<4, 2>(pq[1], r1);
            wl_product_run<4, 4, 2>(pq[0], r0, fragQ, WL_LDQ * 2, xacc);
            wl_ring_fill<4, 2>(pq[2], r0);
            wl_product_run<4, 4, 2>(pq[1], r1, fragQ + 256, WL_LDQ * 2, xacc);
            wl_product_run<4, 4, 2>(pq[2], r0, fragQ + 512, WL_LDQ * 2, xacc);
            WL_STAMP(12 + 3 * hp);
            wl_barrier();
        }
    }

    WL_STAMP(22);
    if constexpr (GATHER) {
        // the skip branch's gradient dx2 is the layer's second output (image A -- dzo -- is dead since phase 3's barrier)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) *reinterpret_cast<uint2*>(imgA + (mb * 16 + lr) * WL_LDA + (4 * w + i) * 16 + 4 * lq) = dx2p[i][mb];
        wl_barrier();
        save_image(imgA, dx);
    } else
    // ---- phase 5: dx1 = acc + dx2; dx = LN'(dx1; x)
    {
        uint2 xp[4][4], d1p[4][4];
        {
            uint4 xrows[8];                              // (datt is dead: the last attention pass lies behind the loop's last barrier)
            fetch_rows(xin, xrows);
            place_rows(xrows, imgB);
        }
        wl_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) xp[i][mb] = *reinterpret_cast<const uint2*>(imgB + (mb * 16 + lr) * WL_LDA + (4 * w + i) * 16 + 4 * lq);
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = (4 * w + i) * 16 + 4 * lq;
            const float4 g4 = *reinterpret_cast<const float4*>(par + WL_F + n);
            const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
            float dgam[4] = {0.f, 0.f, 0.f, 0.f}, dbet[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                float skip[4], xv[4], y[4];
                wl_unpack4(dx2p[i][mb], skip);
                wl_unpack4(xp[i][mb], xv);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // dx1 as the unfused product stores it; rows beyond the tile's tuples: their staging rows were never written (whatever the
                    // LDS held went through the product), they must not reach the parameter gradients
                    y[e] = grow[mb] >= 0 ? wl_round_bf16(xacc[i][mb][e] + skip[e]) : 0.f;
                    const float xh = (xv[e] - mean_1[mb]) * rstd_1[mb], g = y[e] * gg[e];
                    s1[mb] += g;
                    s2[mb] += g * xh;
                    dgam[e] += y[e] * xh;
                    dbet[e] += y[e];
                }
                d1p[i][mb] = wl_pack4(y);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dgam[e] = wl_sum16(dgam[e]);
                dbet[e] = wl_sum16(dbet[e]);
            }
            if (lr == 0) {
                *reinterpret_cast<float4*>(part_1 + n) = make_float4(dgam[0], dgam[1], dgam[2], dgam[3]);
                *reinterpret_cast<float4*>(part_1 + WL_F + n) = make_float4(dbet[0], dbet[1], dbet[2], dbet[3]);
            }
        }
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            s1[mb] = wl_quad_sum(s1[mb]);
            s2[mb] = wl_quad_sum(s2[mb]);
            if (lq == 0) {
                red[w * WL_ROWS + mb * 16 + lr] = s1[mb];
                red[(8 + w) * WL_ROWS + mb * 16 + lr] = s2[mb];
            }
        }
        wl_barrier();
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) {
                a += red[ww * WL_ROWS + mb * 16 + lr];
                b += red[(8 + ww) * WL_ROWS + mb * 16 + lr];
            }
            s1[mb] = a / (float)WL_F;
            s2[mb] = b / (float)WL_F;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = (4 * w + i) * 16 + 4 * lq;
            const float4 g4 = *reinterpret_cast<const float4*>(par + WL_F + n);
            const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                float dyv[4], xv[4], o[4];
                wl_unpack4(d1p[i][mb], dyv);
                wl_unpack4(xp[i][mb], xv);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xh = (xv[e] - mean_1[mb]) * rstd_1[mb];
                    o[e] = rstd_1[mb] * (dyv[e] * gg[e] - s1[mb] - xh * s2[mb]);
                }
                *reinterpret_cast<uint2*>(imgA + (mb * 16 + lr) * WL_LDA + n) = wl_pack4(o);      // (the staging of dq | dk | dv is dead)
            }
        }
        wl_barrier();
        save_image(imgA, dx);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WL_STAMP(23);
}

// W (N x K fp32, rows ldw apart; transpose: the operand is W^T, K x N) -> bf16 in fragment order: block (nb, ks) = 16 rows x 32 k is one
// contiguous KB, lane l of a wavefront holds rows nb 16 + (l & 15), k = ks 32 + 8 (l >> 4) ... + 7 as its 16 bytes
__global__ __launch_bounds__(256) void writer_pack_bf16_kernel(int N, int K, const float* __restrict__ W, int ldw, int transpose, uint16_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;          // one lane fragment (8 elements) per thread
    const int KS = K >> 5;
    if (i >= (size_t)(N >> 4) * KS * 64) return;
    const int l = (int)(i & 63);
    const size_t blk = i >> 6;
    const int ks = (int)(blk % KS), nb = (int)(blk / KS);
    const int n = nb * 16 + (l & 15), k0 = ks * 32 + 8 * (l >> 4);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = transpose ? W[(size_t)(k0 + e) * ldw + n] : W[(size_t)n * ldw + k0 + e];
    reinterpret_cast<uint4*>(out)[i] = wl_pack8(v);
}

}  // namespace

extern "C" size_t grappa_writer_pack_bytes(int N, int K, int dtype) {
    if (N <= 0 || K <= 0 || N % 16 || K % 32 || dtype != GRAPPA_WRITER_BF16) return 0;
    return (size_t)N * K * 2;
}

extern "C" int grappa_writer_pack_weight(void* stream, int N, int K, const float* W, int ldw, int transpose, int dtype, void* out) {
    if (N <= 0 || K <= 0 || N % 16 || K % 32 || !W || !out || dtype != GRAPPA_WRITER_BF16) return GRAPPA_ERR_ARG;
    if (ldw < (transpose ? N : K)) return GRAPPA_ERR_ARG;
    const size_t frags = (size_t)(N / 16) * (K / 32) * 64;
    GRAPPA_LAUNCH(writer_pack_bf16_kernel, dim3((unsigned)((frags + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), N, K, W, ldw,
                  transpose, reinterpret_cast<uint16_t*>(out));
    return grappa_launch_status();
}

extern "C" int grappa_writer_head_fwd(void* stream, const grappa_writer_layer_desc* d) {
    if (!d || d->dtype != GRAPPA_WRITER_BF16 || d->F != WL_F || d->nheads != 8 || d->s < 2 || d->s > 4 || d->T < 0) return GRAPPA_ERR_ARG;
    if (d->T == 0) return GRAPPA_OK;
    if (d->gather_idx) {          // gather mode: x1_tab / qkv_tab instead of x, the first LayerNorm and the q | k | v product (their parameters are not read)
        if (!d->x1_tab || !d->qkv_tab || d->save_x1 || d->save_qkv || d->save_mean1) return GRAPPA_ERR_ARG;
        if (((uintptr_t)d->x1_tab | (uintptr_t)d->qkv_tab) & 15) return GRAPPA_ERR_ARG;
        if (!d->out || !d->w_o_pk || !d->w1_pk || !d->w2_pk || !d->b_in || !d->b_o || !d->b1 || !d->b2 || !d->nf_gamma || !d->nf_beta) return GRAPPA_ERR_ARG;
    } else if (!d->x || !d->out || !d->w_in_pk || !d->w_o_pk || !d->w1_pk || !d->w2_pk || !d->b_in || !d->b_o || !d->b1 || !d->b2 || !d->n1_gamma ||
               !d->n1_beta || !d->nf_gamma || !d->nf_beta)
        return GRAPPA_ERR_ARG;
    if (!(d->drop_p >= 0.f && d->drop_p < 1.f)) return GRAPPA_ERR_ARG;
    if ((d->save_mean1 == nullptr) != (d->save_rstd1 == nullptr) || (d->save_meanf == nullptr) != (d->save_rstdf == nullptr)) return GRAPPA_ERR_ARG;
    const uintptr_t al = (uintptr_t)d->x | (uintptr_t)d->out | (uintptr_t)d->w_in_pk | (uintptr_t)d->w_o_pk | (uintptr_t)d->w1_pk | (uintptr_t)d->w2_pk |
                         (uintptr_t)d->b_in | (uintptr_t)d->b_o | (uintptr_t)d->b1 | (uintptr_t)d->b2 | (uintptr_t)d->n1_gamma | (uintptr_t)d->n1_beta |
                         (uintptr_t)d->nf_gamma | (uintptr_t)d->nf_beta | (uintptr_t)d->save_x1 | (uintptr_t)d->save_qkv | (uintptr_t)d->save_att |
                         (uintptr_t)d->save_x2 | (uintptr_t)d->save_x3 | (uintptr_t)d->save_u;
    if (al & 15) return GRAPPA_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int tt = WL_ROWS / d->s;
    const dim3 grid((unsigned)((d->T + tt - 1) / tt));
    const bool gather = d->gather_idx != nullptr;
    void (*kern)(const grappa_writer_layer_desc) =
        gather ? (d->s == 2 ? writer_layer_fwd_bf16_kernel<2, true> : (d->s == 3 ? writer_layer_fwd_bf16_kernel<3, true> : writer_layer_fwd_bf16_kernel<4, true>))
               : (d->s == 2 ? writer_layer_fwd_bf16_kernel<2, false> : (d->s == 3 ? writer_layer_fwd_bf16_kernel<3, false> : writer_layer_fwd_bf16_kernel<4, false>));
    static bool attr_set[6] = {false, false, false, false, false, false};          // (idempotent: two threads setting it at once set the same value)
    const int ki = (gather ? 3 : 0) + d->s - 2;
    if (!attr_set[ki]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, WL_SMEM) != hipSuccess) return GRAPPA_ERR_LAUNCH;
        attr_set[ki] = true;
    }
    GRAPPA_LAUNCH(kern, grid, dim3(WL_THREADS), WL_SMEM, st, *d);
    return grappa_launch_status();
}

extern "C" int grappa_writer_head_tiles(int s, int T) {
    if (s < 2 || s > 4 || T < 0) return -1;
    const int tt = WL_ROWS / s;
    return (T + tt - 1) / tt;
}

extern "C" int grappa_writer_head_bwd(void* stream, const grappa_writer_layer_bwd_desc* d) {
    if (!d || d->dtype != GRAPPA_WRITER_BF16 || d->F != WL_F || d->nheads != 8 || d->s < 2 || d->s > 4 || d->T < 0) return GRAPPA_ERR_ARG;
    if (d->T == 0) return GRAPPA_OK;
    // gather mode (gather_idx != NULL): `qkv` is the TABLE, `dx` receives dx2; x, the first LayerNorm's tensors and W_in^T are not read
    const void* always[] = {d->dout, d->qkv, d->x2, d->u, d->meanf, d->rstdf, d->nf_gamma, d->w_o_tpk, d->w1_tpk, d->w2_tpk, d->dx, d->dz2, d->dz1, d->dzo, d->dqkv,
                            d->lnf_part};
    const void* full[] = {d->x, d->mean1, d->rstd1, d->n1_gamma, d->w_in_tpk, d->ln1_part};
    uintptr_t al = 0;
    for (const void* q : always) {
        if (!q) return GRAPPA_ERR_ARG;
        al |= (uintptr_t)q;
    }
    if (!d->gather_idx)
        for (const void* q : full) {
            if (!q) return GRAPPA_ERR_ARG;
            al |= (uintptr_t)q;
        }
    if (al & 15) return GRAPPA_ERR_ARG;
    if (!(d->drop_p >= 0.f && d->drop_p < 1.f)) return GRAPPA_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int tt = WL_ROWS / d->s;
    const dim3 grid((unsigned)((d->T + tt - 1) / tt));
    const bool gather = d->gather_idx != nullptr;
    void (*kern)(const grappa_writer_layer_bwd_desc) =
        gather ? (d->s == 2 ? writer_layer_bwd_bf16_kernel<2, true> : (d->s == 3 ? writer_layer_bwd_bf16_kernel<3, true> : writer_layer_bwd_bf16_kernel<4, true>))
               : (d->s == 2 ? writer_layer_bwd_bf16_kernel<2, false> : (d->s == 3 ? writer_layer_bwd_bf16_kernel<3, false> : writer_layer_bwd_bf16_kernel<4, false>));
    static bool attr_set[6] = {false, false, false, false, false, false};
    const int ki = (gather ? 3 : 0) + d->s - 2;
    if (!attr_set[ki]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, WL_SMEM) != hipSuccess) return GRAPPA_ERR_LAUNCH;
        attr_set[ki] = true;
    }
    GRAPPA_LAUNCH(kern, grid, dim3(WL_THREADS), WL_SMEM, st, *d);
    return grappa_launch_status();
}

#ifdef WL_LAB_STAMP
extern "C" int grappa_debug_writer_stamps(unsigned long long* host, int nwgs) {
    if (nwgs > WL_STAMP_WGS) nwgs = WL_STAMP_WGS;
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wl_stamps), (size_t)nwgs * WL_STAMP_WORDS * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
