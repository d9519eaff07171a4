// Device pieces shared by the pair-format GEMM kernels (gemm_pairs.hip: one tile per workgroup; the pinned-pipeline kernels of gemm_pairs_il.hip / gemm_wpairs_il.hip):
// tile shapes, LDS-DMA slab issue, fragment reads, the MFMA block.  The pair format itself: gemm_pairs.hip / include/grappa_hip.h.
#pragma once
#include "gemm_common.h"

using namespace grappa_gemm;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// timing experiments only (tools/pairs_knockouts.sh): 1 = no LDS-DMA after the prologue, 2 = no MFMAs, 3 = no epilogue
#ifndef GQ_KNOCK
#define GQ_KNOCK 0
#endif

// diagnostic build only (tools/pairs_stamps.sh): -DGQ_STAMP=1 makes wavefront 0 of every workgroup of the one-tile kernel leave its timeline in
// a device array that grappa_debug_pairs_stamps() copies out.  No stamp executes in the shipped library.
#ifndef GQ_STAMP
#define GQ_STAMP 0
#endif
#if GQ_STAMP
constexpr int QSTAMP_WORDS = 16, QSTAMP_WGS = 16384;
static __device__ unsigned long long g_q_stamps[QSTAMP_WGS * QSTAMP_WORDS];      // one per translation unit
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

// largest magnitude of row m of A: one array, or the maximum over the per-segment partials its producer left (grappa_gemm_desc.a_amax_nseg);
// eight independent loads per trip (gemm_bf16x_impl.h a_row_amax)
__device__ inline unsigned a_row_amax(const grappa_gemm_desc& d, int m) {
    unsigned v = d.a_amax[m];
    for (int s0 = 1; s0 < d.a_amax_nseg; s0 += 8) {
        unsigned t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = d.a_amax[(size_t)min(s0 + u, d.a_amax_nseg - 1) * d.M + m];
#pragma unroll
        for (int u = 0; u < 8; ++u) v = max(v, t[u]);
    }
    return v;
}

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

// the tail of a tile, shared by the kernels that end with the ring dead: undo the row scales, then the row epilogue through the (reused)
// LDS.  acc element e of block (i, j) is (m, n) = (wm0 + 32 i + lr, wn0 + 32 j + 8 (e / 4) + 4 lh + e % 4)
// SCALED = false: the accumulators are final as they stand (plain bf16 operands: no row scales to undo)
template <int QBN, int QBMt, bool SCALED = true>
__device__ __forceinline__ void pairs_finish(const GemmParams& p, f32x16 (&acc)[QShape<QBN, QBMt>::TM][QTN], char* smem, int m0, int n0, int wm0, int wn0, int wave,
                                             int lane, int split, int tile_local) {
    using S = QShape<QBN, QBMt>;
    constexpr int TM = S::TM;
    const grappa_gemm_desc& d = p.d;
    const int lr = lane & 31, lh = lane >> 5;
    // undo the row scales: accumulator element e of block (i, j) is (m, n) = (wm0 + 32 i + lr, wn0 + 32 j + 8 (e / 4) + 4 lh + e % 4)
    if (SCALED) {
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
}

// ---- "weight pairs" pieces (A = fp32 activations as every producer writes them, B = a weight matrix in the pair format): the wavefronts stand
// 4 x 1, each owns 64 rows x all 128 columns, reads its raw fp32 fragments from the staged rows and splits them in registers
constexpr int WTM = 2, WTN = 4;                 // 32 x 32 accumulators per wavefront (64 x 128)
struct WRaw { float4 a[WTM][2]; };              // raw fp32 of a lane's 8 k per 32-row block
struct WBFrags { f16x8 b[WTN][2]; };            // [column block][hi / lo]

__device__ inline void wread_a(const char* __restrict__ stage, const unsigned (&aoff)[2], int wm0, WRaw& f) {
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int e = 0; e < 2; ++e)      // (read as the f16x8 every other fragment is read as: a float4-typed LDS read made the compiler drain the LDS-DMA queue, vmcnt(0), in front of it)
            f.a[i][e] = __builtin_bit_cast(float4, *reinterpret_cast<const f16x8*>(stage + (wm0 + i * 32) * QROWB + aoff[e]));
}
__device__ inline void wread_b(const char* __restrict__ stage, const unsigned (&boff)[2], WBFrags& f) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int j = 0; j < WTN; ++j) f.b[j][p] = *reinterpret_cast<const f16x8*>(stage + QA_BYTES + j * 32 * QROWB + boff[p]);
}

// 8 consecutive-k fp32 values of one row -> its hi / lo fp16 fragments, scaled by the row's power of two (exact) first
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ inline void wsplit(const float4 (&raw)[2], int shift, f16x8& hi, f16x8& lo) {
    const float r[8] = {__builtin_ldexpf(raw[0].x, shift), __builtin_ldexpf(raw[0].y, shift), __builtin_ldexpf(raw[0].z, shift), __builtin_ldexpf(raw[0].w, shift),
                        __builtin_ldexpf(raw[1].x, shift), __builtin_ldexpf(raw[1].y, shift), __builtin_ldexpf(raw[1].z, shift), __builtin_ldexpf(raw[1].w, shift)};
    unsigned uh[4], ul[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f16x2 h, l;
        h[0] = (_Float16)r[2 * e];                                // round to nearest even; |r| < 2^15 never overflows
        h[1] = (_Float16)r[2 * e + 1];
        l[0] = (_Float16)(r[2 * e] - (float)h[0]);
        l[1] = (_Float16)(r[2 * e + 1] - (float)h[1]);
        uh[e] = __builtin_bit_cast(unsigned, h);
        ul[e] = __builtin_bit_cast(unsigned, l);
    }
    hi = __builtin_bit_cast(f16x8, make_uint4(uh[0], uh[1], uh[2], uh[3]));
    lo = __builtin_bit_cast(f16x8, make_uint4(ul[0], ul[1], ul[2], ul[3]));
}

// CLS 1 .. 4: the straight-line fp32 classes of gemm_common.h; 0: the general walk (and split-K slabs).  Every loop has constant
// bounds and no branch on the class inside: the accumulators stay in registers (a runtime-indexed array would go to scratch)
template <int CLS>
__device__ __forceinline__ void wpairs_epilogue(const GemmParams& p, const f32x16 (&acc)[WTM][WTN], float* __restrict__ wave_buf, int m0, int n0, int wm0,
                                                int lane, int split, int tile_local) {
    const grappa_gemm_desc& d = p.d;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int n = n0 + 64 * h + ((lane & 15) << 2);
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (d.bias && p.nsplit == 1) {
            b4.x = n < d.N ? d.bias[n] : 0.f;
            b4.y = n + 1 < d.N ? d.bias[n + 1] : 0.f;
            b4.z = n + 2 < d.N ? d.bias[n + 2] : 0.f;
            b4.w = n + 3 < d.N ? d.bias[n + 3] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < WTM; ++i) {
            const f32x16 band[2] = {acc[i][2 * h], acc[i][2 * h + 1]};
            const int mb = m0 + wm0 + 32 * i;
            if (CLS != 0) epilogue_band_fast<2, CLS == 0 ? 1 : CLS, float, 4>(p, band, wave_buf, mb, n, lane, b4);
            else epilogue_band<QBM, 128, 2>(p, band, wave_buf, m0, n0, mb, n, lane, b4, split, tile_local, p.vec_io != 0);
        }
    }
}

// the tail of a weight-pairs tile: undo the row scales (sh: the shifts of this lane's two 32-row blocks), then the row epilogue
__device__ __forceinline__ void wpairs_finish(const GemmParams& p, f32x16 (&acc)[WTM][WTN], const int (&sh)[WTM], char* smem, int m0, int n0, int wm0, int wave, int lane,
                                              int split, int tile_local) {
    const grappa_gemm_desc& d = p.d;
    const int lh = lane >> 5;
    // undo the row scales: element e of block (i, j) is (m, n) = (wm0 + 32 i + lr, 32 j + 8 (e / 4) + 4 lh + e % 4)
    {
        const bool b_vec = (reinterpret_cast<uintptr_t>(d.b_amax) & 15) == 0;
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + j * 32 + g * 8 + lh * 4;
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
                    for (int i = 0; i < WTM; ++i) acc[i][j][4 * g + q] = __builtin_ldexpf(acc[i][j][4 * g + q], -(sh[i] + eb[q]));
            }
    }
    __syncthreads();                                         // the ring is dead: reuse as epilogue staging
    if (GQ_KNOCK == 3) {
#pragma unroll
        for (int i = 0; i < WTM; ++i)
#pragma unroll
            for (int j = 0; j < WTN; ++j) asm volatile("" ::"v"(acc[i][j]));
        return;
    }
    // the wavefront's 64 x 128 block as 2 x 2 bands of 32 rows x 64 columns of the shared row epilogue
    float* wave_buf = reinterpret_cast<float*>(smem + wave * EPI_WAVE_BYTES);
    const int cls = p.nsplit == 1 ? p.epi_class : 0;
    switch (cls) {
        case 1: wpairs_epilogue<1>(p, acc, wave_buf, m0, n0, wm0, lane, split, tile_local); break;
        case 2: wpairs_epilogue<2>(p, acc, wave_buf, m0, n0, wm0, lane, split, tile_local); break;
        case 3: wpairs_epilogue<3>(p, acc, wave_buf, m0, n0, wm0, lane, split, tile_local); break;
        case 4: wpairs_epilogue<4>(p, acc, wave_buf, m0, n0, wm0, lane, split, tile_local); break;
        case 5: wpairs_epilogue<5>(p, acc, wave_buf, m0, n0, wm0, lane, split, tile_local); break;      // residual = LayerNorm(res), recomputed (training: the rows were handed on as pairs only)
        default: wpairs_epilogue<0>(p, acc, wave_buf, m0, n0, wm0, lane, split, tile_local); break;
    }
}

}  // namespace
