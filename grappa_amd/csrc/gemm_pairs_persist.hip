// The persistent walk of the pair-format GEMM (round 5); see gemm_pairs.hip for the format and the one-tile-per-workgroup kernel.
#include <cstdlib>
#include "gemm_pairs_impl.h"

namespace {

// ------------------------------------------------------------------------------------------------------------------------------
// Round 5: the PERSISTENT walk.  What the one-tile-per-workgroup kernel above loses at K = 512 (a tile is 32 slabs = ~12 us of MFMAs):
//   * every tile's first slabs arrive at HBM latency with nothing to run meanwhile (prologue), and its 128 KB of results leave while no
//     MFMA of that workgroup runs (epilogue): on a chip whose workgroups all start together the whole grid alternates between a phase
//     that only reads + multiplies and a phase that only writes -- HBM time and MFMA time ADD (these products sit at the ridge: 4 KB of
//     traffic and 0.5 MFLOP per row);
//   * hence here: a workgroup walks tiles lid, lid + nwg, ...; the LDS-DMA of the NEXT tile's first slabs is issued from inside the last
//     steps of the current tile (the ring simply goes on), so they land under the epilogue; the epilogue stages one 32 x 32 accumulator
//     block at a time (LD 36) in the ring stage the next tile does not need yet (BN = 128, two workgroups per CU) or behind the ring
//     (BN = 256); and the second resident workgroup of every CU starts `stagger` sleeps late, so that one's epilogue runs under the
//     other's MFMAs.
// Same products in the same order per tile: bit-identical to PairsBody.  nsplit == 1 and >= 3 slabs only (the host falls back).
// A slab's fragments read in three parts, each behind the last MFMA that used the registers it lands in: the hi halves ahead of the slab's
// MFMAs (double-buffered), B's lo halves behind the first product group (hi_a x lo_b: the only reader of lo_b), A's lo halves behind the second
// (lo_a x hi_b).  24 of the 96 fragment registers are single-buffered that way: the loop fits its 256 without a spill (a spill's reload is an
// ordinary load: the compiler drains the LDS-DMA queue with vmcnt(0) in front of it).  The products keep their order hi*lo, lo*hi, hi*hi.
template <int TM>
__device__ __forceinline__ void qread_hi(const char* __restrict__ stage, int a_bytes, const unsigned (&off)[2], int wm0, int wn0, QFrags<TM>& f) {
#pragma unroll
    for (int j = 0; j < QTN; ++j) f.b[j][0] = *reinterpret_cast<const f16x8*>(stage + a_bytes + (wn0 + j * 32) * QROWB + off[0]);
#pragma unroll
    for (int i = 0; i < TM; ++i) f.a[i][0] = *reinterpret_cast<const f16x8*>(stage + (wm0 + i * 32) * QROWB + off[0]);
}
template <int TM>
__device__ __forceinline__ void qread_blo(const char* __restrict__ stage, int a_bytes, const unsigned (&off)[2], int wn0, QFrags<TM>& f) {
#pragma unroll
    for (int j = 0; j < QTN; ++j) f.b[j][1] = *reinterpret_cast<const f16x8*>(stage + a_bytes + (wn0 + j * 32) * QROWB + off[1]);
}
template <int TM>
__device__ __forceinline__ void qread_alo(const char* __restrict__ stage, const unsigned (&off)[2], int wm0, QFrags<TM>& f) {
#pragma unroll
    for (int i = 0; i < TM; ++i) f.a[i][1] = *reinterpret_cast<const f16x8*>(stage + (wm0 + i * 32) * QROWB + off[1]);
}
template <int TM, int PR>
__device__ __forceinline__ void qmfma_group(const QFrags<TM>& f, f32x16 (&acc)[TM][QTN]) {
    constexpr int pa = PR == 1 ? 1 : 0, pb = PR == 0 ? 1 : 0;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < QTN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.b[j][pb], f.a[i][pa], acc[i][j], 0, 0, 0);
}

constexpr int EPI1_LD = 36;                                     // staged row of ONE 32-column block + 4 pad
constexpr int EPI1_WAVE_BYTES = 32 * EPI1_LD * 4;               // 4,608 B
template <int QBN, int QBMt>
struct PairsPersist {
    using S = QShape<QBN, QBMt>;
    static constexpr bool STAGE_IN_RING = S::STAGE / S::NW >= EPI1_WAVE_BYTES;      // 24 KB / 4 wavefronts: yes; 32 KB / 8: no
    static constexpr size_t SMEM = (size_t)QNSTAGE * S::STAGE + (STAGE_IN_RING ? 0 : (size_t)S::NW * EPI1_WAVE_BYTES);

    // the wavefront's (32 TM) x 64 block, one 32 x 32 accumulator at a time through its private staging rows.  CLS: the straight-line fp32
    // classes of gemm_common.h; 0: the general walk.  Constant loop bounds, the class a template parameter: the accumulators stay in registers
    template <int CLS, int J>
    static __device__ __forceinline__ void epilogue_column(const GemmParams& p, const f32x16 (&acc)[S::TM][QTN], float* __restrict__ wave_buf, int m0, int n0,
                                                           int wm0, int wn0, int lane) {
        const grappa_gemm_desc& d = p.d;
        const int mb = m0 + wm0;
        const int n = n0 + wn0 + J * 32 + ((lane & 7) << 2);
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (d.bias) {
            b4.x = n < d.N ? d.bias[n] : 0.f;
            b4.y = n + 1 < d.N ? d.bias[n + 1] : 0.f;
            b4.z = n + 2 < d.N ? d.bias[n + 2] : 0.f;
            b4.w = n + 3 < d.N ? d.bias[n + 3] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < S::TM; ++i) {
            const f32x16 one[1] = {acc[i][J]};
            if (CLS != 0) epilogue_band_fast<1, CLS == 0 ? 1 : CLS, float, 4, EPI1_LD>(p, one, wave_buf, mb + 32 * i, n, lane, b4);
            else epilogue_band<QBMt, QBN, 1, EPI1_LD>(p, one, wave_buf, m0, n0, mb + 32 * i, n, lane, b4, 0, 0, p.vec_io != 0);
        }
    }
    template <int CLS>
    static __device__ __forceinline__ void epilogue_tile(const GemmParams& p, const f32x16 (&acc)[S::TM][QTN], float* __restrict__ wave_buf, int m0, int n0,
                                                         int wm0, int wn0, int lane) {
        static_assert(QTN == 2, "two 32-column accumulator blocks per wavefront");
        epilogue_column<CLS, 0>(p, acc, wave_buf, m0, n0, wm0, wn0, lane);
        epilogue_column<CLS, 1>(p, acc, wave_buf, m0, n0, wm0, wn0, lane);
    }

static __device__ __forceinline__ void run(const GemmParams& p, int nwg, int orig, int stagger) {
    constexpr int TM = S::TM;
    constexpr int QSTAGE = S::STAGE, QPIECES = S::PIECES;
    extern __shared__ char smem[];
    const grappa_gemm_desc& d = p.d;
    const int xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);      // consecutive ids run on one XCD (map_logical)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wm0 = (wave / S::NWN) * (QBMt / 2), wn0 = (wave % S::NWN) * 64;
    const int lr = lane & 31, lh = lane >> 5;
    const int nslab = (min(d.K, p.k_per_split) + QSLAB - 1) / QSLAB;       // >= 3 (host)
    const unsigned swz = (lr >> 2) & 3;
    const unsigned off[2] = {lr * QROWB + ((lh ^ swz) << 4), lr * QROWB + (((2 + lh) ^ swz) << 4)};
    // the second resident workgroup of a CU (dispatch order: ids 256 .. 511 of a grid of two per CU) starts late
    if (stagger > 0 && ((orig >> 8) & 1)) {
        for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(127);
    }
    int tile = lid;
    if (tile >= p.ntiles_launch) return;
    // Operands through buffer resources (buffer_load_dwordx4 ... lds): the per-lane part of an address is ONE tile-invariant register per
    // operand -- row (wave * 16 + lane / 4) of a piece, 16-byte chunk (lane & 3) ^ swizzle -- everything that moves (tile row, piece, slab) is
    // scalar (soffset); rows beyond M / N lie beyond num_records and read as zeros (no clamping).  No 64-bit vector address arithmetic, 2
    // address registers instead of 6 + 12 temporaries: what lets the walk live in the 256 registers of two workgroups per CU.
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.A), 0, (int)((unsigned)d.M * (unsigned)d.lda * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.B), 0, (int)((unsigned)d.N * (unsigned)d.ldb * 2u), 0x00020000);
    const int rin = lane >> 2, ch = (lane & 3) ^ ((lane >> 4) & 3);
    const int voffA = ((wave * 16 + rin) * d.lda + 8 * ch) * 2, voffB = ((wave * 16 + rin) * d.ldb + 8 * ch) * 2;
    const int strideA = S::NW * 16 * d.lda * 2, strideB = S::NW * 16 * d.ldb * 2;      // bytes between a wavefront's pieces
    auto issue = [&](int m0_, int n0_, int slab, char* stage) {
        const int sA = m0_ * d.lda * 2 + slab * QROWB, sB = n0_ * d.ldb * 2 + slab * QROWB;
#pragma unroll
        for (int q = 0; q < S::A_PIECES; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(stage + (wave + S::NW * q) * 1024), 16, voffA, sA + q * strideA, 0, 0);
#pragma unroll
        for (int q = 0; q < S::B_PIECES; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (__attribute__((address_space(3))) void*)(stage + S::A_BYTES + (wave + S::NW * q) * 1024), 16, voffB,
                                                     sB + q * strideB, 0, 0);
    };
    auto coords = [&](int t, int& m0, int& n0) {
        const int tt = p.tile_begin + t;
        m0 = (tt / p.tiles_n) * QBMt;
        n0 = (tt % p.tiles_n) * QBN;
    };
    int m0, n0;
    coords(tile, m0, n0);
#pragma unroll
    for (int u = 0; u < QNSTAGE; ++u) issue(m0, n0, u, smem + u * QSTAGE);
    int st = 0;                                                  // stage of the current tile's slab 0
    bool first = true;
    for (;;) {
        const int next = tile + nwg;
        const bool has_next = next < p.ntiles_launch;
        int m0n = 0, n0n = 0;
        if (has_next) coords(next, m0n, n0n);
        f32x16 acc[TM][QTN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < QTN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
        // slab 0 of this tile has landed: first tile -- two slabs behind it may fly; later tiles -- everything older than the last slab issued
        // (in the ring: slab 2, issued behind the epilogue; behind the ring: nothing) is complete, the epilogue's stores included
        if (first) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * QPIECES) : "memory");
        else if (STAGE_IN_RING) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QPIECES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        first = false;
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        QFrags<TM> f0, f1;
        qread_frags<TM>(smem + st * QSTAGE, S::A_BYTES, off, wm0, wn0, f0);
        // slabs this tile still issues for the next one: 2 (the third stage stages the epilogue) or 3
        constexpr int NEXT_EARLY = STAGE_IN_RING ? 2 : 3;
#define GP_STEP(T, FC, FN)                                                                                                             \
    do {                                                                                                                               \
        char* cur_ = smem + st * QSTAGE;                                                                                               \
        st = st == QNSTAGE - 1 ? 0 : st + 1;                                                                                           \
        const int ahead_ = (T) + QNSTAGE - nslab;              /* >= 0: the slab to issue belongs to the next tile */                   \
        const bool issue_next_ = has_next && ahead_ >= 0 && ahead_ < NEXT_EARLY;                                                       \
        if ((T) + 1 < nslab) {                                                                                                         \
            /* slab T+1 landed; one younger slab (T+2 of this tile, or the next tile's) may stay in flight */                            \
            if ((T) + 2 < nslab || (has_next && (T) + 2 - nslab < NEXT_EARLY)) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(QPIECES) : "memory"); \
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                            \
            __builtin_amdgcn_s_barrier();                                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                                                         \
            if (ahead_ < 0) issue(m0, n0, (T) + QNSTAGE, cur_);                                                                         \
            else if (issue_next_) issue(m0n, n0n, ahead_, cur_);                                                                        \
            qread_frags<TM>(smem + st * QSTAGE, S::A_BYTES, off, wm0, wn0, FN);                                                        \
        } else if (!STAGE_IN_RING && issue_next_) {                                                                                    \
            /* last step, staging behind the ring: the stage of this slab is free once every wavefront holds its fragments */            \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                          \
            __builtin_amdgcn_s_barrier();                                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                                                         \
            issue(m0n, n0n, ahead_, cur_);                                                                                              \
        }                                                                                                                              \
        qmfma<TM>(FC, acc);                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                                             \
    } while (0)
        int t = 0;
        for (; t + 1 < nslab; t += 2) {
            GP_STEP(t, f0, f1);
            GP_STEP(t + 1, f1, f0);
        }
        if (t < nslab) GP_STEP(t, f0, f1);
#undef GP_STEP
        // here st == stage of the NEXT tile's slab 0 ((old st + nslab) % 3); the stage before it (slab nslab - 1 of this tile) is free
        // undo the row scales
        {
            int ea[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) ea[i] = amax_shift(d.a_amax[min(m0 + wm0 + i * 32 + lr, d.M - 1)]);
            const bool b_vec = (reinterpret_cast<uintptr_t>(d.b_amax) & 15) == 0;
#pragma unroll
            for (int j = 0; j < QTN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = n0 + wn0 + j * 32 + g * 8 + lh * 4;
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
        const int free_stage = st == 0 ? QNSTAGE - 1 : st - 1;
        if (STAGE_IN_RING) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                        // every wavefront holds the fragments of the last slab: its stage stages the epilogue
        }
        float* wave_buf = reinterpret_cast<float*>(STAGE_IN_RING ? smem + free_stage * QSTAGE + wave * (QSTAGE / S::NW)
                                                                 : smem + QNSTAGE * QSTAGE + wave * EPI1_WAVE_BYTES);
        switch (p.epi_class) {
            case 1: epilogue_tile<1>(p, acc, wave_buf, m0, n0, wm0, wn0, lane); break;
            case 2: epilogue_tile<2>(p, acc, wave_buf, m0, n0, wm0, wn0, lane); break;
            case 3: epilogue_tile<3>(p, acc, wave_buf, m0, n0, wm0, wn0, lane); break;
            case 4: epilogue_tile<4>(p, acc, wave_buf, m0, n0, wm0, wn0, lane); break;
            case 5: epilogue_tile<5>(p, acc, wave_buf, m0, n0, wm0, wn0, lane); break;
            default: epilogue_tile<0>(p, acc, wave_buf, m0, n0, wm0, wn0, lane); break;      // the general walk (bf16 outputs, C2, pre, accumulate ...)
        }
        if (!has_next) return;
        if (STAGE_IN_RING) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                        // the staging reads are done: the stage takes the next tile's slab 2
            __builtin_amdgcn_sched_barrier(0);
        }
        tile = next;
        m0 = m0n;
        n0 = n0n;
        if (STAGE_IN_RING) issue(m0, n0, 2, smem + free_stage * QSTAGE);
    }
}
};

template <int QBN>
__global__ __launch_bounds__(QShape<QBN>::NT, 2) void gemm_pairs_persist_kernel(GemmParams p, int stagger) {
    PairsPersist<QBN, QBM>::run(p, gridDim.x, blockIdx.x, stagger);
}

// persistent walk (PairsPersist): grid = the resident workgroups of the chip (two per CU at BN = 128, one at BN = 256) or the tiles, whichever
// is smaller.  GRAPPA_PAIRS_PERSIST=0 keeps the one-tile-per-workgroup kernel (A/B); GRAPPA_PAIRS_STAGGER = sleeps (of 127 x 64 cycles) the
// second resident workgroup of a CU starts late
template <int BN>
int launch_pairs_persist(hipStream_t st, GemmParams& p) {
    using S = QShape<BN>;
    using PP = PairsPersist<BN, QBM>;
    constexpr size_t smem = PP::SMEM;
    static_assert(BN != 128 || smem <= 80 * 1024, "two workgroups per CU");
    static_assert(smem <= 160 * 1024, "LDS");
    auto kern = gemm_pairs_persist_kernel<BN>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return GRAPPA_ERR_LAUNCH;
        attr_set = true;
    }
    static const int stagger_env = getenv("GRAPPA_PAIRS_STAGGER") ? atoi(getenv("GRAPPA_PAIRS_STAGGER")) : -1;
    const int resident = 256 * (BN == 128 ? 2 : 1);
    const int grid = p.ntiles_launch < resident ? p.ntiles_launch : resident;
    // default stagger: half a tile's MFMA time (a slab = 24 MFMAs x 32 cycles per wavefront), only where a second round exists
    const int nslab = ((p.d.K < p.k_per_split ? p.d.K : p.k_per_split) + QSLAB - 1) / QSLAB;
    int stagger = stagger_env >= 0 ? stagger_env : (nslab * 768 / 2) / (127 * 64);
    if (BN != 128 || p.ntiles_launch <= resident) stagger = 0;
    GRAPPA_LAUNCH(kern, dim3(grid), dim3(S::NT), smem, st, p, stagger);
    return grappa_launch_status();
}

}  // namespace

int grappa_launch_gemm_pairs_persist(hipStream_t st, GemmParams& p) { return p.bn == 256 ? launch_pairs_persist<256>(st, p) : launch_pairs_persist<128>(st, p); }
