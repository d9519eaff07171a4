// Device pieces shared by the pair-format GEMM kernels (gemm_pairs.hip: one tile per workgroup; gemm_pairs_persist.hip: the persistent walk):
// tile shapes, LDS-DMA slab issue, fragment reads, the MFMA block.  The pair format itself: gemm_pairs.hip / include/grappa_hip.h.
#pragma once
#include "gemm_common.h"

using namespace grappa_gemm;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// timing experiments only (tools/pairs_knockouts.sh): 1 = no LDS-DMA after the prologue, 2 = no MFMAs, 3 = no epilogue
#ifndef GQ_KNOCK
#define GQ_KNOCK 0
#endif

namespace {

constexpr int QBM = 256, QSLAB = 16, QNSTAGE = 3;
constexpr int QROWB = 64;                       // bytes of a row in a stage: 16 k x (hi, lo)
constexpr int QA_BYTES = QBM * QROWB;           // 16 KB
constexpr int QTN = 2;                          // 32 x 32 accumulator columns per wavefront (64); rows: QShape::TM blocks
// two shapes: BN = 128 on 256 threads (24 KB stages, TWO workgroups per CU: the default) and BN = 256 on 512 threads (32 KB stages, one
// workgroup per CU, 21 instead of 32 operand bytes per MFMA-cycle: GRAPPA_PAIRS_TILE=256)
// a third shape since round 4: BM = 128 (BN = 128, 4 wavefronts of 64 x 64, 16 KB stages): products of a few hundred to a few thousand rows
// (one molecule, a batch of 32) are a handful of workgroups whose time is the padded tile's MFMAs on ONE CU -- half the tile, half the time
template <int BN, int BM = QBM> struct QShape {
    static constexpr int NT = BN * 2;                               // 4 or 8 wavefronts, two rows of them
    static constexpr int NW = NT / 64, NWN = BN / 64;
    static constexpr int TM = BM / 64;                              // 32-row accumulator blocks per wavefront (its rows: BM / 2)
    static constexpr int A_BYTES = BM * QROWB;
    static constexpr int STAGE = (BM + BN) * QROWB;
    static constexpr int A_PIECES = (BM / 16) / NW, B_PIECES = (BN / 16) / NW;      // 1 KB pieces (16 rows) per wavefront and slab
    static constexpr int PIECES = A_PIECES + B_PIECES;
};

__device__ inline int amax_shift(unsigned bits) { return 141 - (int)((bits >> 23) & 0xffu); }

__device__ inline void glds16(const char* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// per-lane source offsets (bytes, without the slab's uniform part) of a wavefront's pieces.  A piece = 16 rows x 64 B: lane -> (row =
// lane >> 2, physical chunk = lane & 3); logical chunk (0, 1: hi k 0..7, 8..15; 2, 3: lo) = physical ^ ((row >> 2) & 3), so that the 16
// lanes of a ds_read_b128 group (rows r..r+3, r+12.., r+20..) hit 16 distinct 16-byte slots of the 256-byte bank row
template <int BN, int BM = QBM> struct QLaneSrc { unsigned a[QShape<BN, BM>::A_PIECES], b[QShape<BN, BM>::B_PIECES]; };
template <int BN, int BM = QBM>
__device__ inline QLaneSrc<BN, BM> qlane_sources(const grappa_gemm_desc& d, int m0, int n0, int wave, int lane) {
    using S = QShape<BN, BM>;
    QLaneSrc<BN, BM> s;
    const int rin = lane >> 2, c = (lane & 3) ^ ((lane >> 4) & 3);
#pragma unroll
    for (int q = 0; q < S::A_PIECES; ++q) s.a[q] = ((unsigned)min(m0 + (wave + S::NW * q) * 16 + rin, d.M - 1) * (unsigned)d.lda + 8u * c) * 2u;
#pragma unroll
    for (int q = 0; q < S::B_PIECES; ++q) s.b[q] = ((unsigned)min(n0 + (wave + S::NW * q) * 16 + rin, d.N - 1) * (unsigned)d.ldb + 8u * c) * 2u;
    return s;
}

template <int BN, int BM = QBM>
__device__ inline void qissue_slab(const char* __restrict__ A, const char* __restrict__ B, size_t k_bytes, const QLaneSrc<BN, BM>& s, char* __restrict__ stage,
                                   int wave) {
    using S = QShape<BN, BM>;
#pragma unroll
    for (int q = 0; q < S::A_PIECES; ++q) glds16(A + k_bytes + s.a[q], stage + (wave + S::NW * q) * 1024);
#pragma unroll
    for (int q = 0; q < S::B_PIECES; ++q) glds16(B + k_bytes + s.b[q], stage + S::A_BYTES + (wave + S::NW * q) * 1024);
}

template <int TM> struct QFrags { f16x8 a[TM][2], b[QTN][2]; };      // [32-row block][hi / lo]

// fragments of a staged slab: 12 (8: TM = 2) ds_read_b128.  off[p] = lr * 64 + ((2 p + lh) ^ swizzle(lr)) * 16
template <int TM>
__device__ inline void qread_frags(const char* __restrict__ stage, int a_bytes, const unsigned (&off)[2], int wm0, int wn0, QFrags<TM>& f) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int j = 0; j < QTN; ++j) f.b[j][p] = *reinterpret_cast<const f16x8*>(stage + a_bytes + (wn0 + j * 32) * QROWB + off[p]);
#pragma unroll
        for (int i = 0; i < TM; ++i) f.a[i][p] = *reinterpret_cast<const f16x8*>(stage + (wm0 + i * 32) * QROWB + off[p]);
    }
}

// the 24 (12) MFMAs of a slab: hi*lo, lo*hi, hi*hi (smallest first), the accumulators innermost.  B fragment first: the accumulator holds
// the transposed tile (4 consecutive n per lane: the row epilogue)
template <int TM>
__device__ inline void qmfma(const QFrags<TM>& f, f32x16 (&acc)[TM][QTN]) {
#pragma unroll
    for (int pr = 0; pr < 3; ++pr) {
        const int pa = pr == 1 ? 1 : 0, pb = pr == 0 ? 1 : 0;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < QTN; ++j) {
                if (GQ_KNOCK == 2) asm volatile("" ::"v"(f.b[j][pb]), "v"(f.a[i][pa]));
                else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.b[j][pb], f.a[i][pa], acc[i][j], 0, 0, 0);
            }
    }
}

}  // namespace
