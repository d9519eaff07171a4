// "Weight pairs" GEMM with the pinned pipeline of gemm_pairs_il.hip: A = fp32 activations [M][K] as every producer writes them (raw rows by
// LDS-DMA, split into fp16 (hi, lo) in registers by the one wavefront that owns them), B = a weight matrix in the pair format.
//
// Why (profiles/r5_shapes_baseline.txt): at C2 the forward / input-gradient products whose A operand is NOT written as pairs by its producer
// -- the second product of every feed-forward (A = the first one's output), every input-gradient product, the GNN's 8,233-row products: 15 of the
// 30 ms of product time -- run the fp32-operand kernel (gemm_bf16x_impl.h MODE H3) at 145 - 190 TFLOP/s: ~200 vector instructions and the LDS
// stores of the split tile per 24 MFMAs, one 512-thread workgroup per CU.  Here: the pair kernel's traffic (24 KB per slab by LDS-DMA, no LDS
// stores), 64 vector instructions per slab -- the split of the NEXT slab's 64 x 16 block, issued between the MFMAs of the current one -- and
// the fixed issue pattern MFMA | reads | split | copies per slab (sched_group_barrier), two workgroups per CU.
// Per slab and wavefront (64 rows x 128 columns, 8 accumulators): 24 MFMAs, 12 ds_read_b128 (4 raw A, 4 + 4 B), 6 LDS-DMA pieces.  B's lo
// fragments are single-buffered (their only reader is the first product group), everything else alternates between two register sets.
// Same products in the same order as the pair kernels and MODE H3: bit-identical results for the same K split.
#include <cstdlib>
#include "gemm_pairs_impl.h"

namespace {

constexpr int SGW_MFMA = 0x008, SGW_VMEM_R = 0x020, SGW_DS_R = 0x100, SGW_VALU = 0x002;

struct WSet { f16x8 ah[WTM], al[WTM]; };      // split A of a slab (double-buffered); B's fragments are single-buffered

template <int PR>
__device__ __forceinline__ void wmfma_group(const f16x8 (&a)[WTM], const f16x8 (&b)[WTN], f32x16 (&acc)[WTM][WTN]) {
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[j], a[i], acc[i][j], 0, 0, 0);
}

__device__ __forceinline__ void wread_bh(const char* __restrict__ stage, const unsigned (&boff)[2], f16x8 (&bh)[WTN]) {
#pragma unroll
    for (int j = 0; j < WTN; ++j) bh[j] = *reinterpret_cast<const f16x8*>(stage + QA_BYTES + j * 32 * QROWB + boff[0]);
}
__device__ __forceinline__ void wread_bl(const char* __restrict__ stage, const unsigned (&boff)[2], f16x8 (&bl)[WTN]) {
#pragma unroll
    for (int j = 0; j < WTN; ++j) bl[j] = *reinterpret_cast<const f16x8*>(stage + QA_BYTES + j * 32 * QROWB + boff[1]);
}

// N times: one MFMA, then NO instructions of class MASK (NO may be 0), then up to NV vector instructions
template <int N, int MASK, int NO, int NV>
struct SgRepW {
    static __device__ __forceinline__ void emit() {
        __builtin_amdgcn_sched_group_barrier(SGW_MFMA, 1, 0);
        if (NO > 0) __builtin_amdgcn_sched_group_barrier(MASK, NO, 0);
        if (NV > 0) __builtin_amdgcn_sched_group_barrier(SGW_VALU, NV, 0);
        SgRepW<N - 1, MASK, NO, NV>::emit();
    }
};
template <int MASK, int NO, int NV>
struct SgRepW<0, MASK, NO, NV> {
    static __device__ __forceinline__ void emit() {}
};

__global__ __launch_bounds__(256, 2) void gemm_wpairs_il_kernel(GemmParams p) {
    constexpr int QBN = 128, QSTAGE = (QBM + QBN) * QROWB, QPIECES = 6;
    extern __shared__ char smem[];
    const grappa_gemm_desc& d = p.d;
    const TileCoord tc = map_workgroup(p);
    const int split = tc.split, tile_local = tc.tile_local;
    const int m0 = tc.tile_m * QBM, n0 = tc.tile_n * QBN;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wm0 = wave * 64;
    const int lr = lane & 31, lh = lane >> 5;
    const int kbeg = split * p.k_per_split;
    const int kend = min(d.K, kbeg + p.k_per_split);
    const int nslab = (kend - kbeg) / QSLAB;                  // even and >= 4 (host: K and the K cuts are multiples of 32)

    f32x16 acc[WTM][WTN];
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    int sh[WTM];
#pragma unroll
    for (int i = 0; i < WTM; ++i) {
        // the row's largest magnitude: one array, or the maximum over the per-segment partials its producer's epilogue left (a_amax_nseg, C ABI 8):
        // the combine launch between two products goes away
        sh[i] = amax_shift(a_row_amax(d, min(m0 + wm0 + i * 32 + lr, d.M - 1)));
    }

    // buffer resources: A rows are fp32 (lda floats apart), B rows pairs (ldb fp16 apart); rows beyond M / N read as zeros
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.A), 0, (int)((unsigned)d.M * (unsigned)d.lda * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.B), 0, (int)((unsigned)d.N * (unsigned)d.ldb * 2u), 0x00020000);
    const int rin = lane >> 2, ch = (lane & 3) ^ ((lane >> 4) & 3);
    const int voffA = ((wave * 16 + rin) * d.lda + 4 * ch) * 4, voffB = ((wave * 16 + rin) * d.ldb + 8 * ch) * 2;
    const int strideA = 4 * 16 * d.lda * 4, strideB = 4 * 16 * d.ldb * 2;      // bytes between a wavefront's pieces (4 wavefronts x 16 rows)
    const int sA0 = m0 * d.lda * 4 + kbeg * 4, sB0 = n0 * d.ldb * 2 + kbeg * 4;  // 16 k = 64 bytes of a row, fp32 and pairs alike
    auto issue = [&](int slab, char* stage) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(stage + (wave + 4 * q) * 1024), 16, voffA, sA0 + slab * QROWB + q * strideA, 0, 0);
#pragma unroll
        for (int q = 0; q < 2; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (__attribute__((address_space(3))) void*)(stage + QA_BYTES + (wave + 4 * q) * 1024), 16, voffB,
                                                     sB0 + slab * QROWB + q * strideB, 0, 0);
    };
    const unsigned swz = (lr >> 2) & 3;
    const unsigned aoff[2] = {lr * QROWB + (((2 * lh) ^ swz) << 4), lr * QROWB + (((2 * lh + 1) ^ swz) << 4)};     // floats 8 lh .. + 3, + 4 .. + 7
    const unsigned boff[2] = {lr * QROWB + ((lh ^ swz) << 4), lr * QROWB + (((2 + lh) ^ swz) << 4)};

    WSet s0, s1;
    f16x8 bl[WTN], bh[WTN];
    WRaw raw;
#pragma unroll
    for (int u = 0; u < QNSTAGE; ++u) issue(u, smem + u * QSTAGE);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * QPIECES) : "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    wread_a(smem, aoff, wm0, raw);
    wread_bh(smem, boff, bh);
    wread_bl(smem, boff, bl);
#pragma unroll
    for (int i = 0; i < WTM; ++i) wsplit(raw.a[i], sh[i], s0.ah[i], s0.al[i]);
    int st = 0;
    // one slab: SC holds slab T's split A and B hi, `bl` its B lo; SN receives slab T + 1's.  ISSUE: slab T + 3 exists; READ: slab T + 1 exists;
    // LAST2: slab T + 2 does not exist
#define GW_STEP(T, SC, SN, ISSUE, READ, LAST2)                                                                                         \
    do {                                                                                                                               \
        char* cur_ = smem + st * QSTAGE;                                                                                               \
        st = st == QNSTAGE - 1 ? 0 : st + 1;                                                                                           \
        if (READ) {                                                                                                                    \
            if (LAST2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                      \
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(QPIECES) : "memory");                                             \
            __builtin_amdgcn_s_barrier();                                                                                              \
        } else {                                                                                                                       \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                          \
        }                                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                                             \
        const char* nxt_ = smem + st * QSTAGE;                                                                                         \
        /* region 1: hi_a x lo_b (the only reader of bl) | the next slab's raw A */                                                     \
        wmfma_group<0>(SC.ah, bl, acc);                                                                                                \
        if (READ) {                                                                                                                    \
            wread_a(nxt_, aoff, wm0, raw);                                                                                             \
            SgRepW<4, SGW_DS_R, 1, 0>::emit();                                                                                         \
        }                                                                                                                              \
        __builtin_amdgcn_sched_group_barrier(SGW_MFMA, 8, 0);                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                                             \
        /* region 2: lo_a x hi_b | the next slab's B lo into the registers region 1 has just read | split of the first 32-row block */   \
        wmfma_group<1>(SC.al, bh, acc);                                                                                                \
        if (READ) {                                                                                                                    \
            wread_bl(nxt_, boff, bl);                                                                                                  \
            wsplit(raw.a[0], sh[0], SN.ah[0], SN.al[0]);                                                                               \
            SgRepW<4, SGW_DS_R, 1, 6>::emit();                                                                                         \
            SgRepW<4, 0, 0, 6>::emit();                                                                                                \
        }                                                                                                                              \
        __builtin_amdgcn_sched_group_barrier(SGW_MFMA, 8, 0);                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                                             \
        /* region 3: hi_a x hi_b | split of the second block | the copies of slab T + 3 | B hi behind its last reader */                 \
        wmfma_group<2>(SC.ah, bh, acc);                                                                                                \
        if (READ) wsplit(raw.a[1], sh[1], SN.ah[1], SN.al[1]);                                                                         \
        if (ISSUE) issue((T) + QNSTAGE, cur_);                                                                                         \
        if (READ) wread_bh(nxt_, boff, bh);                                                                                            \
        if (ISSUE) {                                                                                                                   \
            SgRepW<6, SGW_VMEM_R, 1, 6>::emit();                                                                                       \
            SgRepW<2, 0, 0, 6>::emit();                                                                                                \
        } else if (READ) {                                                                                                             \
            SgRepW<8, 0, 0, 6>::emit();                                                                                                \
        }                                                                                                                              \
        __builtin_amdgcn_sched_group_barrier(SGW_MFMA, 8, 0);                                                                          \
        if (READ) __builtin_amdgcn_sched_group_barrier(SGW_DS_R, 4, 0);                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                                             \
    } while (0)
    int t = 0;
    for (; t + 4 < nslab; t += 2) {
        GW_STEP(t, s0, s1, true, true, false);
        GW_STEP(t + 1, s1, s0, true, true, false);
    }
    GW_STEP(t, s0, s1, true, true, false);                   // four left
    GW_STEP(t + 1, s1, s0, false, true, false);
    GW_STEP(t + 2, s0, s1, false, true, true);
    GW_STEP(t + 3, s1, s0, false, false, true);
#undef GW_STEP
    wpairs_finish(p, acc, sh, smem, m0, n0, wm0, wave, lane, split, tile_local);
}

}  // namespace

// fp32 A + weight pairs, K and every K cut a multiple of 32, at least four slabs per workgroup (the caller checks)
int grappa_launch_gemm_wpairs_il(hipStream_t st, GemmParams& p) {
    constexpr size_t ring = (size_t)QNSTAGE * (QBM + 128) * QROWB, staging = 4 * (size_t)EPI_WAVE_BYTES;
    constexpr size_t smem = ring > staging ? ring : staging;
    static_assert(smem <= 80 * 1024, "two workgroups per CU");
    auto kern = gemm_wpairs_il_kernel;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return GRAPPA_ERR_LAUNCH;
        attr_set = true;
    }
    GRAPPA_LAUNCH(kern, dim3(p.ntiles_launch * p.nsplit), dim3(256), smem, st, p);
    return grappa_launch_status();
}
