// Pair-format GEMM, round 5: the main loop as ONE pinned instruction pipeline per slab.
//
// What the in-kernel stamps of the round-3 kernel (gemm_pairs.hip PairsBody) said (profiles/r5_pairs_stamps_*.txt):
//   * a workgroup ALONE on its CU spends 1,350 cycles per slab for 768 cycles of MFMAs (products of 8,233 .. 17,158 rows at C2: one round
//     of tiles, half of the chip's MFMA time idle): its instruction stream was [wait | barrier | 6 LDS-DMA | 12 ds_read | wait for them |
//     24 MFMAs] -- the matrix pipe idles while the wavefront issues its copies and reads (~60 cycles per LDS-DMA piece), and a scalar reload
//     inside the loop makes the compiler wait lgkmcnt(0) for the fragments just requested;
//   * two workgroups per CU cover each other's gaps only to 79 % (62 k cycles per tile against 49 k of MFMAs), and while one of them is in its
//     epilogue the other runs at the lone rate.
// Here: the 24 MFMAs of slab t are issued FIRST and the fragment reads of slab t + 1 and the LDS-DMA pieces of slab t + 3 are slotted between
// them in a fixed pattern (sched_group_barrier): reads in the first half of the MFMAs -- so that 12 more MFMAs cover their latency before the
// next slab's wait -- copies in the second half.  Nothing scalar is loaded inside the loop: operands are addressed through buffer resources
// (one tile-invariant address register per operand, everything that moves is an SGPR offset), the steps that issue / read nothing are peeled
// off so that every step is one basic block.  Same products in the same order: bit-identical to PairsBody.
#include <cstdlib>
#include "gemm_pairs_impl.h"

namespace {

constexpr int SG_MFMA = 0x008, SG_VMEM_R = 0x020, SG_DS_R = 0x100;

// N times: `NM` MFMAs then one instruction of class MASK
template <int N, int NM, int MASK>
struct SgRep {
    static __device__ __forceinline__ void emit() {
        __builtin_amdgcn_sched_group_barrier(SG_MFMA, NM, 0);
        __builtin_amdgcn_sched_group_barrier(MASK, 1, 0);
        SgRep<N - 1, NM, MASK>::emit();
    }
};
template <int NM, int MASK>
struct SgRep<0, NM, MASK> {
    static __device__ __forceinline__ void emit() {}
};

// ARITH 0: the pair format (fp16 hi / lo, three products per slab of 16 k).  ARITH 1 (round 5, the bf16 STORAGE configuration, BASELINE
// configs[2]): both operands plain bf16 rows (one plane), a 64-byte granule = 32 k, two v_mfma_f32_32x32x16_bf16 per accumulator and slab.
// Same tile, ring, LDS image and fragment addresses (the "hi" chunks are k 0 .. 15 of the granule, the "lo" chunks k 16 .. 31); no row scales.
// The plane kernel this replaces for forward / input-gradient products (gemm_planes.hip, 8 wavefronts of 64 x 64, two barriers per slab of
// 8 MFMAs) ran at 0.2 of the bf16 peak.
typedef __bf16 bf16x8_il __attribute__((ext_vector_type(8)));

template <int QBN, int QBMt, int ARITH = 0>
struct PairsIL {
    using S = QShape<QBN, QBMt>;
    static constexpr int TM = S::TM;
    static constexpr int KSLAB = ARITH == 1 ? 32 : QSLAB;          // k per 64-byte granule
    static constexpr int NMFMA = (ARITH == 1 ? 2 : 3) * TM * QTN, NREAD = 2 * (TM + QTN), NDMA = S::PIECES;
    // reads ride on the first MFMAs (one or two per MFMA), copies on the following ones
    static constexpr int READS_PER_MFMA = NREAD <= NMFMA / 2 ? 1 : 2;
    static constexpr int MFMA_WITH_READS = NREAD / READS_PER_MFMA;
    static constexpr int MFMA_PER_DMA = (NMFMA - MFMA_WITH_READS) / NDMA > 0 ? (NMFMA - MFMA_WITH_READS) / NDMA : 1;
    static_assert(NREAD % READS_PER_MFMA == 0 && MFMA_WITH_READS + NDMA * MFMA_PER_DMA <= NMFMA, "pipeline pattern");

    template <bool ISSUE, bool READ>
    static __device__ __forceinline__ void pattern() {
        if (READ) {
            if (READS_PER_MFMA == 1) SgRep<MFMA_WITH_READS, 1, SG_DS_R>::emit();
            else {
                // one MFMA, two reads
#pragma unroll
                for (int i = 0; i < MFMA_WITH_READS; ++i) {
                    __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(SG_DS_R, 2, 0);
                }
            }
        }
        if (ISSUE) SgRep<NDMA, MFMA_PER_DMA, SG_VMEM_R>::emit();
        __builtin_amdgcn_sched_group_barrier(SG_MFMA, NMFMA, 0);      // whatever MFMAs are left
    }

    static __device__ __forceinline__ void mfma_slab(const QFrags<TM>& f, f32x16 (&acc)[TM][QTN]) {
        if (ARITH == 0) {
            qmfma<TM>(f, acc);
        } else {
#pragma unroll
            for (int pk = 0; pk < 2; ++pk)                         // k 0 .. 15, then 16 .. 31 of the granule
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < QTN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_il, f.b[j][pk]), __builtin_bit_cast(bf16x8_il, f.a[i][pk]), acc[i][j], 0, 0, 0);
        }
    }

    static __device__ __forceinline__ void run(const GemmParams& p, int nwg, int wgid) {
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
        // an EVEN number of slabs >= 4 (host): the rows are zero beyond K up to the next multiple of 32 -- the step pair of the unrolled loop
        // then always leaves exactly four steps behind it, ONE tail path
        const int nslab = (kend - kbeg + 2 * KSLAB - 1) / (2 * KSLAB) * 2;

#if GQ_STAMP
        unsigned long long q_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q_wait = 0, q_ta = 0, q_tb = 0;
        q_t[0] = __builtin_amdgcn_s_memrealtime();
        q_t[1] = __builtin_amdgcn_s_memtime();
#define GI_STAMP_A q_wait += q_tb - q_ta; q_ta = __builtin_amdgcn_s_memtime();
#define GI_STAMP_B q_tb = __builtin_amdgcn_s_memtime();
#else
#define GI_STAMP_A
#define GI_STAMP_B
#endif
        f32x16 acc[TM][QTN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < QTN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

        // buffer resources: rows beyond M / N lie beyond num_records and read as zeros
        const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.A), 0, (int)((unsigned)d.M * (unsigned)d.lda * 2u), 0x00020000);
        const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.B), 0, (int)((unsigned)d.N * (unsigned)d.ldb * 2u), 0x00020000);
        const int rin = lane >> 2, ch = (lane & 3) ^ ((lane >> 4) & 3);
        const int voffA = ((wave * 16 + rin) * d.lda + 8 * ch) * 2, voffB = ((wave * 16 + rin) * d.ldb + 8 * ch) * 2;
        const int strideA = S::NW * 16 * d.lda * 2, strideB = S::NW * 16 * d.ldb * 2;      // bytes between a wavefront's pieces
        const int kb0 = kbeg * (64 / KSLAB);                                              // 16 k (pairs) or 32 k (bf16) = 64 bytes of a row
        const int sA0 = m0 * d.lda * 2 + kb0, sB0 = n0 * d.ldb * 2 + kb0;
        auto issue = [&](int slab, char* stage) {
#pragma unroll
            for (int q = 0; q < S::A_PIECES; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(stage + (wave + S::NW * q) * 1024), 16, voffA,
                                                         sA0 + slab * QROWB + q * strideA, 0, 0);
#pragma unroll
            for (int q = 0; q < S::B_PIECES; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (__attribute__((address_space(3))) void*)(stage + S::A_BYTES + (wave + S::NW * q) * 1024), 16, voffB,
                                                         sB0 + slab * QROWB + q * strideB, 0, 0);
        };
        const unsigned swz = (lr >> 2) & 3;
        const unsigned off[2] = {lr * QROWB + ((lh ^ swz) << 4), lr * QROWB + (((2 + lh) ^ swz) << 4)};
        QFrags<TM> f0, f1;

#pragma unroll
        for (int u = 0; u < QNSTAGE; ++u) issue(u, smem + u * QSTAGE);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * QPIECES) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#if GQ_STAMP
        q_t[2] = __builtin_amdgcn_s_memtime();
#endif
        qread_frags<TM>(smem, S::A_BYTES, off, wm0, wn0, f0);
        int st = 0;                                          // t % 3
        // one slab: FC holds slab T's fragments; FN receives those of slab T + 1.  ISSUE: slab T + 3 exists; READ: slab T + 1 exists;
        // LAST2: slab T + 2 does not exist (nothing may stay in flight behind slab T + 1)
#define GI_STEP(T, FC, FN, ISSUE, READ, LAST2)                                                                                         \
    do {                                                                                                                               \
        char* cur_ = smem + st * QSTAGE;                                                                                               \
        st = st == QNSTAGE - 1 ? 0 : st + 1;                                                                                           \
        GI_STAMP_A                                                                                                                     \
        if (READ) {                                                                                                                    \
            if (LAST2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                      \
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(QPIECES) : "memory");                                             \
            __builtin_amdgcn_s_barrier();                                                                                              \
        } else {                                                                                                                       \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                          \
        }                                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                                             \
        GI_STAMP_B                                                                                                                     \
        mfma_slab(FC, acc);                                                                                                            \
        if (READ) qread_frags<TM>(smem + st * QSTAGE, S::A_BYTES, off, wm0, wn0, FN);                                                  \
        if (ISSUE) issue((T) + QNSTAGE, cur_);                                                                                         \
        pattern<ISSUE, READ>();                                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                                             \
    } while (0)
        int t = 0;
        for (; t + 4 < nslab; t += 2) {                      // both steps have a slab t + 3 to issue
            GI_STEP(t, f0, f1, true, true, false);
            GI_STEP(t + 1, f1, f0, true, true, false);
        }
        GI_STEP(t, f0, f1, true, true, false);               // four left: the last slab is issued, then the ring drains
        GI_STEP(t + 1, f1, f0, false, true, false);
        GI_STEP(t + 2, f0, f1, false, true, true);
        GI_STEP(t + 3, f1, f0, false, false, true);
#undef GI_STEP
#if GQ_STAMP
        q_wait += q_tb - q_ta;
        q_t[3] = __builtin_amdgcn_s_memtime();
#endif
        pairs_finish<QBN, QBMt, ARITH == 0>(p, acc, smem, m0, n0, wm0, wn0, wave, lane, split, tile_local);
#if GQ_STAMP
        q_t[4] = __builtin_amdgcn_s_memtime();                    // (issue of the last store; scale-back and epilogue are one phase here)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        q_t[5] = __builtin_amdgcn_s_memtime();
        q_t[6] = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0 && wgid < QSTAMP_WGS) {
            unsigned long long* o = g_q_stamps + (size_t)wgid * QSTAMP_WORDS;
            unsigned hw = 0, xcc = 0;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            o[0] = q_t[0]; o[1] = q_t[1]; o[2] = q_t[2]; o[3] = q_t[3]; o[4] = q_t[3]; o[5] = q_t[5]; o[6] = q_t[6]; o[7] = q_wait;
            o[8] = hw; o[9] = xcc; o[10] = (unsigned)tc.tile_m; o[11] = (unsigned)tc.tile_n; o[12] = (unsigned)nslab; o[13] = q_t[4];
        }
#endif
    }
};

template <int QBN>
__global__ __launch_bounds__(QShape<QBN>::NT, 2) void gemm_pairs_il_kernel(GemmParams p) { PairsIL<QBN, QBM>::run(p, gridDim.x, blockIdx.x); }
__global__ __launch_bounds__(256, 2) void gemm_pairs_il_small_kernel(GemmParams p) { PairsIL<128, 128>::run(p, gridDim.x, blockIdx.x); }

template <int QBN>
__global__ __launch_bounds__(QShape<QBN>::NT, 2) void gemm_bf16_il_kernel(GemmParams p) { PairsIL<QBN, QBM, 1>::run(p, gridDim.x, blockIdx.x); }

__global__ __launch_bounds__(256, 2) void gemm_bf16_il_small_kernel(GemmParams p) { PairsIL<128, 128, 1>::run(p, gridDim.x, blockIdx.x); }

template <int BN, int BM, typename K>
int launch_il(hipStream_t st, GemmParams& p, K kern, bool& attr_set) {
    using S = QShape<BN, BM>;
    constexpr size_t ring = (size_t)QNSTAGE * S::STAGE, staging = S::NW * (size_t)EPI_WAVE_BYTES;
    constexpr size_t smem = ring > staging ? ring : staging;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return GRAPPA_ERR_LAUNCH;
        attr_set = true;
    }
    GRAPPA_LAUNCH(kern, dim3(p.ntiles_launch * p.nsplit), dim3(S::NT), smem, st, p);
    return grappa_launch_status();
}

}  // namespace

#if GQ_STAMP
extern "C" int grappa_debug_pairs_il_stamps(unsigned long long* host, int nwgs) {
    if (nwgs > QSTAMP_WGS) nwgs = QSTAMP_WGS;
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_q_stamps), (size_t)nwgs * QSTAMP_WORDS * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

// both operands in the pair format, >= 4 slabs per workgroup (the caller checks)
int grappa_launch_gemm_pairs_il(hipStream_t st, GemmParams& p) {
    static bool a0 = false, a1 = false, a2 = false;
    if (p.bm == 128) return launch_il<128, 128>(st, p, gemm_pairs_il_small_kernel, a0);
    if (p.bn == 256) return launch_il<256, QBM>(st, p, gemm_pairs_il_kernel<256>, a1);
    return launch_il<128, QBM>(st, p, gemm_pairs_il_kernel<128>, a2);
}

// both operands ONE bf16 plane, K-contiguous, K (and every K cut) a multiple of 64, at least four slabs of 32 per workgroup, tile 256 x 128
// (the caller checks): the bf16 storage configuration's forward / input-gradient products
int grappa_launch_gemm_bf16_il(hipStream_t st, GemmParams& p) {
    static bool a0 = false, a1 = false, a2 = false;
    if (p.bm == 128) return launch_il<128, 128>(st, p, gemm_bf16_il_small_kernel, a2);      // experiment (GRAPPA_BF16_TILE=128)
    if (p.bn == 256) return launch_il<256, QBM>(st, p, gemm_bf16_il_kernel<256>, a1);
    return launch_il<128, QBM>(st, p, gemm_bf16_il_kernel<128>, a0);
}
