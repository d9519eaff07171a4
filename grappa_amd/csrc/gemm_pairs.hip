// fp32-grade GEMM on the fp16 matrix cores from operands that are ALREADY split and scaled: the "pair format".
//
// Pair format of an fp32 matrix X[R][C] (include/grappa_hip.h): per row one fp32 bit pattern amax_r >= max_c |X[r][c]| and
//     s_r  = 141 - exponent_field(amax_r)
//     HI   = f16(X * 2^s_r),  LO = f16(X * 2^s_r - HI)         (round to nearest even; the residual is exact in fp32)
// so that X = (HI + LO) * 2^-s_r to 22 significant bits and more (11 + 11 and the sign of the residual), the row's largest element
// sitting in [2^14, 2^15).  The two fp16 values of an element live in the same 64-byte granule of its row: a row is a sequence of
// blocks of 16 consecutive k, each [16 x HI | 16 x LO] -- element (r, k): HI at fp16 index r * ld + 32 * (k / 16) + k % 16, LO 16 further.
// It is the operand the fp16-split arithmetic of gemm_bf16x_impl.h (MODE H3) builds in every workgroup of every launch from fp32
// rows, built ONCE by the kernel that produces the tensor (LayerNorm, the s <= 4 attention, dropout backward, the GEMM epilogue;
// weights once per optimiser step) at the same 4 bytes per element as fp32.
//
// What the first version of this kernel measured (profiles/r3_gemm_pairs_v1_knockouts.txt; 8 wavefronts, 256 x 128 tile, one workgroup
// per CU): with the split arithmetic gone the kernel was 5 % faster, not 50 % -- a CU takes in ~43 GB/s by LDS-DMA whatever the line
// usage, the MFMA-only loop runs at 1.17 us per 32-deep slab, and prologue + epilogue (a quarter of a K = 512 tile) overlap with nothing
// because a CU holds ONE workgroup.  Hence this shape:
//   * 256 threads = 4 wavefronts as 2 x 2, each 128 x 64 of the 256 x 128 tile (8 accumulators): 12 fragment reads per 24 MFMAs;
//   * K advances in slabs of 16: a stage = (256 + 128) rows x 64 B = 24 KB, a ring of THREE = 72 KB -> TWO workgroups per CU.  Their
//     phases drift apart: one's epilogue (vector-bound: scale-back, bias, ELU, dropout hash) and prologue run under the other's MFMAs;
//   * one barrier per slab: wait for this wavefront's pieces of slab t+1 (counted vmcnt: slab t+2 stays in flight) and for its
//     fragment reads of slab t | barrier | DMA of slab t+3 into the stage slab t occupied | fragment reads of slab t+1 | 24 MFMAs of
//     slab t.  A slab's DMA is in flight for two slab times.
// The main loop has no vector arithmetic: tiles go global -> LDS by LDS-DMA (16 B per lane, no registers), fragments are single
// ds_read_b128, products hi*lo, lo*hi, hi*hi (smallest first) as v_mfma_f32_32x32x16_f16 into fp32 accumulators, scaled back by
// 2^-(s_m + s_n) (v_ldexp_f32, exact) in front of the shared row epilogue.  Same products in the same order as MODE H3: bit-identical
// results for the same K split (tests/test_gpu_pairs.py).
#include <cstdlib>
#include "gemm_common.h"

using namespace grappa_gemm;

int grappa_launch_gemm_pairs_il(hipStream_t st, GemmParams& p);           // gemm_pairs_il.hip
int grappa_launch_gemm_wpairs_il(hipStream_t st, GemmParams& p);          // gemm_wpairs_il.hip

#include "gemm_pairs_impl.h"

namespace {

template <int QBN, int QBMt>
struct PairsBody {
static __device__ __forceinline__ void run(const GemmParams& p, int nwg, int wgid) {
    using S = QShape<QBN, QBMt>;
    constexpr int TM = S::TM;
    constexpr int QSTAGE = S::STAGE, QPIECES = S::PIECES;
    extern __shared__ char smem[];
    const grappa_gemm_desc& d = p.d;
    const TileCoord tc = map_logical(p, nwg, wgid);
    const int split = tc.split, tile_local = tc.tile_local;
    const int m0 = tc.tile_m * QBMt, n0 = tc.tile_n * QBN;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wm0 = (wave / S::NWN) * (QBMt / 2), wn0 = (wave % S::NWN) * 64;
    const int lr = lane & 31, lh = lane >> 5;
    const int kbeg = split * p.k_per_split;
    const int kend = min(d.K, kbeg + p.k_per_split);
    const int nslab = (kend - kbeg + QSLAB - 1) / QSLAB;      // the rows are zero beyond K up to the next multiple of 32
#if GQ_STAMP
    unsigned long long q_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q_wait = 0, q_ta = 0, q_tb = 0;
    q_t[0] = __builtin_amdgcn_s_memrealtime();
    q_t[1] = __builtin_amdgcn_s_memtime();
#endif

    f32x16 acc[TM][QTN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < QTN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    if (nslab > 0) {
        const char* A = reinterpret_cast<const char*>(d.A);
        const char* B = reinterpret_cast<const char*>(d.B);
        const size_t kb0 = (size_t)kbeg * 4;                  // 16 k = 64 bytes of a row
        const QLaneSrc<QBN, QBMt> src = qlane_sources<QBN, QBMt>(d, m0, n0, wave, lane);
        const unsigned swz = (lr >> 2) & 3;
        const unsigned off[2] = {lr * QROWB + ((lh ^ swz) << 4), lr * QROWB + (((2 + lh) ^ swz) << 4)};
        QFrags<TM> f0, f1;

#pragma unroll
        for (int u = 0; u < QNSTAGE; ++u)
            if (u < nslab) qissue_slab<QBN, QBMt>(A, B, kb0 + (size_t)u * QROWB, src, smem + u * QSTAGE, wave);
        if (nslab >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * QPIECES) : "memory");
        else if (nslab == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QPIECES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#if GQ_STAMP
        q_t[2] = __builtin_amdgcn_s_memtime();
#endif
        qread_frags<TM>(smem, S::A_BYTES, off, wm0, wn0, f0);
        int st = 0;                                          // t % 3
        // one slab: FC holds slab T's fragments; FN receives those of slab T + 1
#define GQ_STEP(T, FC, FN)                                                                                                             \
    do {                                                                                                                               \
        char* cur_ = smem + st * QSTAGE;                                                                                               \
        st = st == QNSTAGE - 1 ? 0 : st + 1;                                                                                           \
        if ((T) + 1 < nslab) {                                                                                                         \
            GQ_STAMP_A                                                                                                                 \
            /* this wavefront's pieces of slab T+1 have landed (slab T+2 may stay in flight) and its reads of slab T are in registers; */ \
            /* behind the barrier that holds for every wavefront: stage `cur_` is free, slab T+1 is readable */                          \
            if ((T) + 2 < nslab) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(QPIECES) : "memory");                             \
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                            \
            __builtin_amdgcn_s_barrier();                                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                                                         \
            GQ_STAMP_B                                                                                                                 \
            if ((T) + QNSTAGE < nslab && GQ_KNOCK != 1) qissue_slab<QBN, QBMt>(A, B, kb0 + (size_t)((T) + QNSTAGE) * QROWB, src, cur_, wave); \
            qread_frags<TM>(smem + st * QSTAGE, S::A_BYTES, off, wm0, wn0, FN);                                                                        \
        }                                                                                                                              \
        qmfma<TM>(FC, acc);                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                                             \
    } while (0)
#if GQ_STAMP
#define GQ_STAMP_A q_wait += q_tb - q_ta; q_ta = __builtin_amdgcn_s_memtime();
#define GQ_STAMP_B q_tb = __builtin_amdgcn_s_memtime();
#else
#define GQ_STAMP_A
#define GQ_STAMP_B
#endif
        int t = 0;
        for (; t + 1 < nslab; t += 2) {
            GQ_STEP(t, f0, f1);
            GQ_STEP(t + 1, f1, f0);
        }
        if (t < nslab) GQ_STEP(t, f0, f1);
#undef GQ_STEP
#if GQ_STAMP
        q_wait += q_tb - q_ta;
#endif
    }
#if GQ_STAMP
    q_t[3] = __builtin_amdgcn_s_memtime();
#endif

    // undo the row scales: accumulator element e of block (i, j) is (m, n) = (wm0 + 32 i + lr, wn0 + 32 j + 8 (e / 4) + 4 lh + e % 4)
    {
        int ea[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) ea[i] = amax_shift(d.a_amax[min(m0 + wm0 + i * 32 + lr, d.M - 1)]);
        const bool b_vec = (reinterpret_cast<uintptr_t>(d.b_amax) & 15) == 0;
#pragma unroll
        for (int j = 0; j < QTN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn0 + j * 32 + g * 8 + lh * 4;                 // four consecutive columns, n % 4 == 0
                int eb[4];
                if (b_vec && n + 3 < d.N) {
                    const uint4 u = *reinterpret_cast<const uint4*>(d.b_amax + n);
                    eb[0] = amax_shift(u.x); eb[1] = amax_shift(u.y); eb[2] = amax_shift(u.z); eb[3] = amax_shift(u.w);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) eb[q] = amax_shift(d.b_amax[min(n + q, d.N - 1)]);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int i = 0; i < TM; ++i) acc[i][j][4 * g + q] = __builtin_ldexpf(acc[i][j][4 * g + q], -(ea[i] + eb[q]));
            }
    }
    __syncthreads();                                         // the ring is dead: reuse as epilogue staging
#if GQ_STAMP
    q_t[4] = __builtin_amdgcn_s_memtime();
#endif
    if (GQ_KNOCK == 3) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < QTN; ++j) asm volatile("" ::"v"(acc[i][j]));
        return;
    }
    // the wavefront's 128 x 64 block as four 32-row bands of the shared row epilogue
    float* wave_buf = reinterpret_cast<float*>(smem + wave * EPI_WAVE_BYTES);
    const int n = n0 + wn0 + ((lane & 15) << 2);
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d.bias && p.nsplit == 1) {
        b4.x = n < d.N ? d.bias[n] : 0.f;
        b4.y = n + 1 < d.N ? d.bias[n + 1] : 0.f;
        b4.z = n + 2 < d.N ? d.bias[n + 2] : 0.f;
        b4.w = n + 3 < d.N ? d.bias[n + 3] : 0.f;
    }
    const int mb = m0 + wm0;
    if (p.epi_class != 0 && p.nsplit == 1) {
#define GQ_FAST(CLS, T)                                                                                            \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) epilogue_band_fast<QTN, CLS, T, 4>(p, acc[i], wave_buf, mb + 32 * i, n, lane, b4); \
    break
        switch (p.epi_class) {
            case 1: GQ_FAST(1, float);
            case 2: GQ_FAST(2, float);
            case 3: GQ_FAST(3, float);
            case 4: GQ_FAST(4, float);
            case 5: GQ_FAST(5, float);
            case 9: GQ_FAST(1, grappa_bf16_t);
            case 10: GQ_FAST(2, grappa_bf16_t);
            case 11: GQ_FAST(3, grappa_bf16_t);
            default: GQ_FAST(4, grappa_bf16_t);
        }
#undef GQ_FAST
    } else {
#pragma unroll
        for (int i = 0; i < TM; ++i) epilogue_band<QBMt, QBN, QTN>(p, acc[i], wave_buf, m0, n0, mb + 32 * i, n, lane, b4, split, tile_local, p.vec_io != 0);
    }
#if GQ_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the stores have left
    q_t[5] = __builtin_amdgcn_s_memtime();
    q_t[6] = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && wgid < QSTAMP_WGS) {
        unsigned long long* o = g_q_stamps + (size_t)wgid * QSTAMP_WORDS;
        unsigned hw = 0, xcc = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        o[0] = q_t[0]; o[1] = q_t[1]; o[2] = q_t[2]; o[3] = q_t[3]; o[4] = q_t[4]; o[5] = q_t[5]; o[6] = q_t[6]; o[7] = q_wait;
        o[8] = hw; o[9] = xcc; o[10] = (unsigned)tc.tile_m; o[11] = (unsigned)tc.tile_n; o[12] = (unsigned)nslab;
    }
#endif
}
};

template <int QBN>
__global__ __launch_bounds__(QShape<QBN>::NT, 2) void gemm_pairs_kernel(GemmParams p) { PairsBody<QBN, QBM>::run(p, gridDim.x, blockIdx.x); }
__global__ __launch_bounds__(256, 2) void gemm_pairs_small_kernel(GemmParams p) { PairsBody<128, 128>::run(p, gridDim.x, blockIdx.x); }

// up to four independent pair-format products (the same product of the four writer heads) in one grid
template <int QBN>
__global__ __launch_bounds__(QShape<QBN>::NT, 2) void gemm_pairs_group4_kernel(GemmGroup4 g) {
    const int i = group4_find(g, blockIdx.x);
    PairsBody<QBN, QBM>::run(g.p[i], g.wg_begin[i + 1] - g.wg_begin[i], blockIdx.x - g.wg_begin[i]);
}

// ------------------------------------------------------------------------------------------------------------------------------
// "Weight pairs": A = fp32 activations [M][K] AS EVERY PRODUCER WRITES THEM, B = a weight matrix in the pair format (split once per
// optimiser step).  Same ring, same two workgroups per CU; the A tile reaches the LDS as raw fp32 rows by LDS-DMA (16 k = one 64-byte
// granule per row, the bytes of a pair granule), the wavefronts stand 4 x 1 -- each owns 64 rows x all 128 columns -- so that every A
// fragment is read, scaled and split into its (hi, lo) by exactly ONE wavefront, in registers: 64 vector instructions per slab beside
// 24 MFMAs (the fp32-operand kernel: ~200 per 24 across its two wavefronts of a SIMD, plus the LDS stores of the split tile), nothing
// staged in registers, and the producer side of the model untouched.
struct WLaneSrc { unsigned a[4], b[2]; };
__device__ inline void wissue(const char* __restrict__ A, const char* __restrict__ B, size_t kb, const WLaneSrc& s, char* __restrict__ stage, int wave) {
#pragma unroll
    for (int q = 0; q < 4; ++q) glds16(A + kb + s.a[q], stage + (wave + 4 * q) * 1024);
#pragma unroll
    for (int q = 0; q < 2; ++q) glds16(B + kb + s.b[q], stage + QA_BYTES + (wave + 4 * q) * 1024);
}

__global__ __launch_bounds__(256, 2) void gemm_wpairs_kernel(GemmParams p) {
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
    const int nslab = (kend - kbeg + QSLAB - 1) / QSLAB;      // K % 16 == 0 (host-checked: fp32 rows carry no padding)

    f32x16 acc[WTM][WTN];
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    int sh[WTM];
#pragma unroll
    for (int i = 0; i < WTM; ++i) sh[i] = amax_shift(d.a_amax[min(m0 + wm0 + i * 32 + lr, d.M - 1)]);

    if (nslab > 0) {
        const char* A = reinterpret_cast<const char*>(d.A);
        const char* B = reinterpret_cast<const char*>(d.B);
        const size_t kb0 = (size_t)kbeg * 4;                  // 16 k = 64 bytes of a row, fp32 and pairs alike
        // LDS-DMA pieces of 16 rows x 64 B: A rows are fp32 (lda floats apart), B rows pairs (ldb fp16 apart)
        WLaneSrc src;
        {
            const int rin = lane >> 2, c = (lane & 3) ^ ((lane >> 4) & 3);
#pragma unroll
            for (int q = 0; q < 4; ++q) src.a[q] = ((unsigned)min(m0 + (wave + 4 * q) * 16 + rin, d.M - 1) * (unsigned)d.lda + 4u * c) * 4u;
#pragma unroll
            for (int q = 0; q < 2; ++q) src.b[q] = ((unsigned)min(n0 + (wave + 4 * q) * 16 + rin, d.N - 1) * (unsigned)d.ldb + 8u * c) * 2u;
        }
        const unsigned swz = (lr >> 2) & 3;
        const unsigned aoff[2] = {lr * QROWB + (((2 * lh) ^ swz) << 4), lr * QROWB + (((2 * lh + 1) ^ swz) << 4)};     // floats 8 lh .. + 3, + 4 .. + 7
        const unsigned boff[2] = {lr * QROWB + ((lh ^ swz) << 4), lr * QROWB + (((2 + lh) ^ swz) << 4)};
        WRaw r0, r1;              // raw A of even / odd slabs (the next slab's is read under this slab's MFMAs)
        WBFrags fb;               // B fragments of the current slab only: re-read behind the MFMAs that consumed them (registers: 128
                                  // accumulators + 32 + 2 x 16 + 16 split = 224; the other workgroup of the CU covers the read latency)

#pragma unroll
        for (int u = 0; u < QNSTAGE; ++u)
            if (u < nslab) wissue(A, B, kb0 + (size_t)u * QROWB, src, smem + u * QSTAGE, wave);
        if (nslab >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * QPIECES) : "memory");
        else if (nslab == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QPIECES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        wread_a(smem, aoff, wm0, r0);
        wread_b(smem, boff, fb);
        int st = 0;
#define GW_STEP(T, RC, RN)                                                                                                             \
    do {                                                                                                                               \
        char* cur_ = smem + st * QSTAGE;                                                                                               \
        st = st == QNSTAGE - 1 ? 0 : st + 1;                                                                                           \
        if ((T) + 1 < nslab) {                                                                                                         \
            if ((T) + 2 < nslab) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(QPIECES) : "memory");                             \
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                            \
            __builtin_amdgcn_s_barrier();                                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                                                         \
            if ((T) + QNSTAGE < nslab && GQ_KNOCK != 1) wissue(A, B, kb0 + (size_t)((T) + QNSTAGE) * QROWB, src, cur_, wave);           \
            wread_a(smem + st * QSTAGE, aoff, wm0, RN);                                                                                \
        }                                                                                                                              \
        f16x8 ah_[WTM], al_[WTM];                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < WTM; ++i) wsplit(RC.a[i], sh[i], ah_[i], al_[i]);                                        \
        _Pragma("unroll") for (int pr = 0; pr < 3; ++pr)                                                                               \
            _Pragma("unroll") for (int i = 0; i < WTM; ++i)                                                                            \
                _Pragma("unroll") for (int j = 0; j < WTN; ++j) {                                                                      \
                    if (GQ_KNOCK == 2) asm volatile("" ::"v"(fb.b[j][pr == 0 ? 1 : 0]), "v"(pr == 1 ? al_[i] : ah_[i]));                \
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb.b[j][pr == 0 ? 1 : 0], pr == 1 ? al_[i] : ah_[i], acc[i][j], 0, 0, 0); \
                }                                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                                             \
        if ((T) + 1 < nslab) wread_b(smem + st * QSTAGE, boff, fb);                                                                    \
    } while (0)
        int t = 0;
        for (; t + 1 < nslab; t += 2) {
            GW_STEP(t, r0, r1);
            GW_STEP(t + 1, r1, r0);
        }
        if (t < nslab) GW_STEP(t, r0, r1);
#undef GW_STEP
    }

    wpairs_finish(p, acc, sh, smem, m0, n0, wm0, wave, lane, split, tile_local);
}

int launch_wpairs(hipStream_t st, GemmParams& p) {
    constexpr size_t ring = (size_t)QNSTAGE * (QBM + 128) * QROWB, staging = 4 * (size_t)EPI_WAVE_BYTES;
    constexpr size_t smem = ring > staging ? ring : staging;
    static_assert(smem <= 80 * 1024, "two workgroups per CU");
    auto kern = gemm_wpairs_kernel;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return GRAPPA_ERR_LAUNCH;
        attr_set = true;
    }
    GRAPPA_LAUNCH(kern, dim3(p.ntiles_launch * p.nsplit), dim3(256), smem, st, p);
    return grappa_launch_status();
}

template <int BN>
int launch_pairs(hipStream_t st, GemmParams& p) {
    using S = QShape<BN>;
    constexpr size_t ring = (size_t)QNSTAGE * S::STAGE, staging = S::NW * (size_t)EPI_WAVE_BYTES;
    constexpr size_t smem = ring > staging ? ring : staging;
    static_assert(BN != 128 || smem <= 80 * 1024, "two workgroups per CU");
    auto kern = gemm_pairs_kernel<BN>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return GRAPPA_ERR_LAUNCH;
        attr_set = true;
    }
    GRAPPA_LAUNCH(kern, dim3(p.ntiles_launch * p.nsplit), dim3(S::NT), smem, st, p);
    return grappa_launch_status();
}

int launch_pairs_small(hipStream_t st, GemmParams& p) {
    using S = QShape<128, 128>;
    constexpr size_t ring = (size_t)QNSTAGE * S::STAGE, staging = S::NW * (size_t)EPI_WAVE_BYTES;
    constexpr size_t smem = ring > staging ? ring : staging;
    auto kern = gemm_pairs_small_kernel;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return GRAPPA_ERR_LAUNCH;
        attr_set = true;
    }
    GRAPPA_LAUNCH(kern, dim3(p.ntiles_launch * p.nsplit), dim3(S::NT), smem, st, p);
    return grappa_launch_status();
}

// fp32 X[R][C] -> pair format; one 32 x 32 tile per 256-thread workgroup, through LDS when transposing.  amax: bit patterns of the
// largest magnitude of every OUTPUT row (R values, or C values when transposing)
__device__ inline size_t pair_index(size_t row, int k, int ldo) { return row * ldo + 32 * (k >> 4) + (k & 15); }

__global__ __launch_bounds__(256) void split_pairs_kernel(int R, int C, const float* __restrict__ x, int ldx, const unsigned* __restrict__ amax,
                                                          uint16_t* __restrict__ out, int ldo, int transpose) {
    __shared__ uint16_t tile[2][32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tc = threadIdx.x & 31, tr = threadIdx.x >> 5;          // 8 rows per pass
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = r0 + tr + 8 * q, c = c0 + tc;
        const bool ok = r < R && c < C;
        float v = ok ? x[(size_t)r * ldx + c] : 0.f;
        v = __builtin_ldexpf(v, ok ? amax_shift(amax[transpose ? c : r]) : 0);
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        const uint16_t hb = __builtin_bit_cast(uint16_t, hi), lb = __builtin_bit_cast(uint16_t, lo);
        if (transpose) {
            tile[0][tr + 8 * q][tc] = hb;
            tile[1][tr + 8 * q][tc] = lb;
        } else if (ok) {
            const size_t o = pair_index(r, c, ldo);
            out[o] = hb;
            out[o + 16] = lb;
        }
    }
    if (!transpose) return;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = c0 + tr + 8 * q, r = r0 + tc;                  // out row c, k = r: consecutive threads -> consecutive k
        if (c < C && r < R) {
            const size_t o = pair_index(c, r, ldo);
            out[o] = tile[0][tc][tr + 8 * q];
            out[o + 16] = tile[1][tc][tr + 8 * q];
        }
    }
}

// many matrices in ONE launch (every weight of a model, both orientations, once per optimiser step): a table of items in device memory,
// tile_begin = prefix of 32 x 32 tiles; a workgroup finds its item by bisection
__global__ __launch_bounds__(256) void split_pairs_batched_kernel(int count, const grappa_split_pairs_item* __restrict__ items) {
    __shared__ uint16_t tile[2][32][33];
    int lo = 0, hi = count - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].tile_begin <= (int)blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const grappa_split_pairs_item it = items[lo];
    const int local = (int)blockIdx.x - it.tile_begin, tiles_x = (it.C + 31) / 32;
    const int r0 = (local / tiles_x) * 32, c0 = (local % tiles_x) * 32;
    const int tc = threadIdx.x & 31, tr = threadIdx.x >> 5;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = r0 + tr + 8 * q, c = c0 + tc;
        const bool ok = r < it.R && c < it.C;
        float v = ok ? it.x[(size_t)r * it.ldx + c] : 0.f;
        v = __builtin_ldexpf(v, ok ? amax_shift(it.amax[it.transpose ? c : r]) : 0);
        const _Float16 hi16 = (_Float16)v;
        const _Float16 lo16 = (_Float16)(v - (float)hi16);
        const uint16_t hb = __builtin_bit_cast(uint16_t, hi16), lb = __builtin_bit_cast(uint16_t, lo16);
        if (it.transpose) {
            tile[0][tr + 8 * q][tc] = hb;
            tile[1][tr + 8 * q][tc] = lb;
        } else if (ok) {
            const size_t o = pair_index(r, c, it.ldp);
            it.pairs[o] = hb;
            it.pairs[o + 16] = lb;
        }
    }
    if (!it.transpose) return;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = c0 + tr + 8 * q, r = r0 + tc;
        if (c < it.C && r < it.R) {
            const size_t o = pair_index(c, r, it.ldp);
            it.pairs[o] = tile[0][tc][tr + 8 * q];
            it.pairs[o + 16] = tile[1][tc][tr + 8 * q];
        }
    }
}

}  // namespace

extern "C" int grappa_split_pairs_f32_batched(void* stream, int count, int total_tiles, const grappa_split_pairs_item* items) {
    if (count < 0 || total_tiles < 0 || (count > 0 && !items)) return GRAPPA_ERR_ARG;
    if (count == 0 || total_tiles == 0) return GRAPPA_OK;
    GRAPPA_LAUNCH(split_pairs_batched_kernel, dim3(total_tiles), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), count, items);
    return grappa_launch_status();
}

extern "C" int grappa_split_pairs_f32(void* stream, int R, int C, const float* x, int ldx, const uint32_t* amax, uint16_t* pairs, int ldp,
                                      int transpose) {
    if (R < 0 || C < 0) return GRAPPA_ERR_ARG;
    if (R == 0 || C == 0) return GRAPPA_OK;
    const int kk = transpose ? R : C;
    if (!x || !amax || !pairs || ldx < C || ldp < 2 * ((kk + 15) / 16 * 16)) return GRAPPA_ERR_ARG;
    GRAPPA_LAUNCH(split_pairs_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), R, C, x, ldx,
                       amax, pairs, ldp, transpose);
    return grappa_launch_status();
}

#if GQ_STAMP
extern "C" int grappa_debug_pairs_stamps(unsigned long long* host, int nwgs) {
    if (nwgs > QSTAMP_WGS) nwgs = QSTAMP_WGS;
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_q_stamps), (size_t)nwgs * QSTAMP_WORDS * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

// called by grappa_gemm_f32_group: pair-format products, tile 256 x 128, two workgroups per CU
int grappa_launch_gemm_pairs_group4(hipStream_t st, const GemmGroup4& g) {
    using S = QShape<128>;
    constexpr size_t ring = (size_t)QNSTAGE * S::STAGE, staging = S::NW * (size_t)EPI_WAVE_BYTES;
    constexpr size_t smem = ring > staging ? ring : staging;
    auto kern = gemm_pairs_group4_kernel<128>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return GRAPPA_ERR_LAUNCH;
        attr_set = true;
    }
    GRAPPA_LAUNCH(kern, dim3(g.wg_begin[g.count]), dim3(S::NT), smem, st, g);
    return grappa_launch_status();
}

// called by grappa_gemm_f32 (gemm_f32.hip) when both operands are in the pair format (precision F32_F16X3); tile 256 x 128
int grappa_launch_gemm_pairs(hipStream_t st, GemmParams& p) {
    if (!p.d.a_planes) {                                     // fp32 A, weight pairs
        // the pinned pipeline of gemm_wpairs_il.hip wherever every K range of the launch holds its four slabs (grappa_wpairs_il_takes: the host's
        // check of a_amax_nseg uses the same predicate); the round-3 loop for the rest
        if (grappa_wpairs_il_takes(p.d.K, p.nsplit, p.k_per_split)) return grappa_launch_gemm_wpairs_il(st, p);
        if (p.d.a_amax_nseg > 1) return GRAPPA_ERR_ARG;          // (it reads ONE maximum per row; unreachable through grappa_gemm_f32)
        return launch_wpairs(st, p);
    }
    const int kk = p.d.K - (p.nsplit - 1) * p.k_per_split;       // the shortest K range of the launch (the last)
    // (the pinned pipeline walks PAIRS of slabs: a K range that is not a multiple of 32 is rounded up, which is harmless at the end of a row -- the
    //  pair format pads rows with zeros -- and wrong at the boundary between two split-K ranges, where the 16 extra columns belong to the next range)
    if (kk >= 4 * QSLAB && (p.nsplit == 1 || (p.k_per_split & 31) == 0)) return grappa_launch_gemm_pairs_il(st, p);      // the pinned pipeline (gemm_pairs_il.hip)
    if (p.bm == 128) return launch_pairs_small(st, p);
    return p.bn == 256 ? launch_pairs<256>(st, p) : launch_pairs<128>(st, p);
}
