// fp32-grade GEMM on the fp16 matrix cores from operands that are ALREADY split and scaled: the "pair format".
//
// Pair format of an fp32 matrix X[R][C] (include/grappa_hip.h): two fp16 matrices HI, LO of X's shape and one fp32 bit pattern per row,
//     s_r  = 141 - exponent_field(amax_r)             (amax_r >= max_c |X[r][c]|: the row's largest magnitude, or an upper bound of it)
//     HI   = f16(X * 2^s_r),  LO = f16(X * 2^s_r - HI)         (round to nearest even; the residual is exact in fp32)
// so that X = (HI + LO) * 2^-s_r to 22 significant bits and more (11 + 11 and the sign of the residual), the row's largest element
// sitting in [2^14, 2^15).  It is the operand the fp16-split arithmetic of gemm_bf16x_impl.h (MODE H3) builds in every workgroup of
// every launch from fp32 rows -- there 3,200 vector instructions per wavefront and tile beside 384 MFMAs at K = 512, and every row of A
// re-split by each of its 4 - 12 column tiles.  Here the tensor is split ONCE, by the kernel that produces it (LayerNorm, the s <= 4
// attention, dropout backward, the GEMM epilogue; weights once per optimiser step), at the same 4 bytes per element as fp32.
//
// The main loop has no vector arithmetic: tiles go global -> LDS by LDS-DMA (16 B per lane, no registers), fragments are single
// ds_read_b128, the wavefront issues hi*lo, lo*hi, hi*hi (smallest first) as v_mfma_f32_32x32x16_f16 into fp32 accumulators and
// scales them back by 2^-(s_m + s_n) (v_ldexp_f32, exact) in front of the shared row epilogue.  Same products in the same order as
// MODE H3: bit-identical results (tests/test_gpu_pairs.py).
//
// Tile 256 x 128, 512 threads = 8 wavefronts as 4 x 2, 64 x 64 per wavefront.  K advances in slabs of 32; a stage = 2 planes x
// (256 + 128) rows x 64 B = 48 KB; a ring of THREE stages (144 KB) and ONE barrier per slab:
//   iteration t: read the fragments of the slab's second half | 12 MFMAs on the first half | wait for this wavefront's pieces of slab
//   t+1 (counted vmcnt: slab t+2 stays in flight) | barrier: every wavefront holds all of slab t in registers and slab t+1 has landed |
//   DMA of slab t+3 into the stage slab t occupied | read the first half of slab t+1 | 12 MFMAs on the second half.
// A slab's DMA is in flight for two slab times (~3,000 cycles at the MFMA-bound pace).
#include "gemm_common.h"

using namespace grappa_gemm;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// timing experiments only (tools/gemm_pairs_check.py --variants): 1 = no LDS-DMA after the prologue, 2 = no MFMAs, 3 = no epilogue
#ifndef GQ_KNOCK
#define GQ_KNOCK 0
#endif

namespace {

constexpr int QBM = 256, QBN = 128, QSLAB = 32, QNT = 512;
constexpr int QA_PLANE = QBM * QSLAB * 2;       // bytes of one plane of the A tile in a stage (16 KB)
constexpr int QB_PLANE = QBN * QSLAB * 2;       // 8 KB
constexpr int QSTAGE = 2 * (QA_PLANE + QB_PLANE);
constexpr int QPIECES = 6;                      // LDS-DMA instructions per wavefront and slab: 2 planes x (2 of A + 1 of B)

__device__ inline int amax_shift(unsigned bits) { return 141 - (int)((bits >> 23) & 0xffu); }

__device__ inline void glds16(const char* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// per-lane source offsets (bytes, without the slab's uniform part) of a wavefront's three pieces per plane: two 1 KB pieces of A
// (tile row blocks `wave` and `wave + 8`), one of B.  A piece = 16 rows x 64 B: lane -> (row = lane >> 2, physical chunk = lane & 3),
// logical chunk = physical ^ ((row >> 2) & 3): the 16 lanes of a ds_read_b128 group then hit 16 distinct 16-byte slots of the bank row
struct QLaneSrc { unsigned a0, a1, b; };
__device__ inline QLaneSrc qlane_sources(const grappa_gemm_desc& d, int m0, int n0, int wave, int lane) {
    QLaneSrc s;
    if (GQ_KNOCK == 4) {
        // timing experiment: the same bytes fetched as WHOLE 128-byte lines (8 rows x 128 B per piece; qissue_slab walks k in steps of 128 B
        // over half the rows at a time).  Results are wrong by construction
        const int rin = lane >> 3, c = lane & 7;
        s.a0 = ((unsigned)min(m0 + wave * 8 + rin, d.M - 1) * (unsigned)d.lda + 8u * c) * 2u;
        s.a1 = ((unsigned)min(m0 + (wave + 8) * 8 + rin, d.M - 1) * (unsigned)d.lda + 8u * c) * 2u;
        s.b = ((unsigned)min(n0 + wave * 8 + rin, d.N - 1) * (unsigned)d.ldb + 8u * c) * 2u;
        return s;
    }
    const int rin = lane >> 2, c = (lane & 3) ^ ((lane >> 4) & 3);
    const int ra0 = min(m0 + wave * 16 + rin, d.M - 1), ra1 = min(m0 + (wave + 8) * 16 + rin, d.M - 1);
    const int rb = min(n0 + wave * 16 + rin, d.N - 1);
    s.a0 = ((unsigned)ra0 * (unsigned)d.lda + 8u * c) * 2u;
    s.a1 = ((unsigned)ra1 * (unsigned)d.lda + 8u * c) * 2u;
    s.b = ((unsigned)rb * (unsigned)d.ldb + 8u * c) * 2u;
    return s;
}

__device__ inline void qissue_slab(const char* __restrict__ A, const char* __restrict__ B, size_t a_plane_bytes, size_t b_plane_bytes, size_t k_bytes,
                                   const QLaneSrc& s, char* __restrict__ stage, int wave, size_t row_bytes = 0, size_t row_bytes_ld_a = 0,
                                   size_t row_bytes_ld_b = 0) {
    if (GQ_KNOCK == 4) {
        // slab t = k_bytes / 64 reads bytes [128 t, 128 t + 128) of rows 0..127 (A) / 0..63 (B) of the tile while that stays inside the
        // row (row_bytes = the K range), then the same of the other half of the rows
        const size_t kb = k_bytes * 2;
        const size_t ka = kb % row_bytes, ra = (kb / row_bytes) * 128 * (row_bytes_ld_a), rb = (kb / row_bytes) * 64 * (row_bytes_ld_b);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            glds16(A + p * a_plane_bytes + ra + ka + s.a0, stage + p * QA_PLANE + wave * 1024);
            glds16(A + p * a_plane_bytes + ra + ka + s.a1, stage + p * QA_PLANE + (wave + 8) * 1024);
            glds16(B + p * b_plane_bytes + rb + ka + s.b, stage + 2 * QA_PLANE + p * QB_PLANE + wave * 1024);
        }
        return;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const char* ap = A + p * a_plane_bytes + k_bytes;
        const char* bp = B + p * b_plane_bytes + k_bytes;
        glds16(ap + s.a0, stage + p * QA_PLANE + wave * 1024);
        glds16(ap + s.a1, stage + p * QA_PLANE + (wave + 8) * 1024);
        glds16(bp + s.b, stage + 2 * QA_PLANE + p * QB_PLANE + wave * 1024);
    }
}

struct QFrags { f16x8 a[2][2], b[2][2]; };      // [32-row block][plane]

// fragments of k-half kh (16 k) of a staged slab: 8 ds_read_b128
__device__ inline void qread_frags(const char* __restrict__ stage, unsigned off, int wm0, int wn0, QFrags& f) {
    const char* a_s = stage;
    const char* b_s = stage + 2 * QA_PLANE;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int i = 0; i < 2; ++i) f.a[i][p] = *reinterpret_cast<const f16x8*>(a_s + p * QA_PLANE + (wm0 + i * 32) * 64 + off);
#pragma unroll
        for (int j = 0; j < 2; ++j) f.b[j][p] = *reinterpret_cast<const f16x8*>(b_s + p * QB_PLANE + (wn0 + j * 32) * 64 + off);
    }
}

// the 12 MFMAs of one k-half: hi*lo, lo*hi, hi*hi (smallest first), the 2 x 2 accumulators innermost.  B fragment first: the
// accumulator holds the transposed tile (4 consecutive n per lane: tile_epilogue_rows)
__device__ inline void qmfma(const QFrags& f, f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int pr = 0; pr < 3; ++pr) {
        const int pa = pr == 1 ? 1 : 0, pb = pr == 0 ? 1 : 0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (GQ_KNOCK == 2) asm volatile("" ::"v"(f.b[j][pb]), "v"(f.a[i][pa]));
                else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.b[j][pb], f.a[i][pa], acc[i][j], 0, 0, 0);
            }
    }
}

template <int NSTAGE>
__global__ __launch_bounds__(QNT, 2) void gemm_pairs_kernel(GemmParams p) {
    static_assert(NSTAGE == 2 || NSTAGE == 3, "ring of two or three stages");
    extern __shared__ char smem[];
    const grappa_gemm_desc& d = p.d;
    const TileCoord tc = map_workgroup(p);
    const int split = tc.split, tile_local = tc.tile_local;
    const int m0 = tc.tile_m * QBM, n0 = tc.tile_n * QBN;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
    const int lr = lane & 31, lh = lane >> 5;
    const int kbeg = split * p.k_per_split;
    const int kend = min(d.K, kbeg + p.k_per_split);
    const int nslab = (kend - kbeg + QSLAB - 1) / QSLAB;      // the planes are zero beyond K up to the next multiple of 32

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    if (nslab > 0) {
        const char* A = reinterpret_cast<const char*>(d.A);
        const char* B = reinterpret_cast<const char*>(d.B);
        const size_t apb = d.a_plane_stride * 2, bpb = d.b_plane_stride * 2;
        const size_t kb0 = (size_t)kbeg * 2;                  // both operands K-contiguous: a slab advances 64 bytes along every row
        const QLaneSrc src = qlane_sources(d, m0, n0, wave, lane);
        const unsigned swz = (lr >> 2) & 3;
        const unsigned off0 = lr * 64 + ((lh ^ swz) << 4), off1 = lr * 64 + (((2 + lh) ^ swz) << 4);
        QFrags f0, f1;

#pragma unroll
        for (int u = 0; u < NSTAGE; ++u)
            if (u < nslab) qissue_slab(A, B, apb, bpb, kb0 + (size_t)u * QSLAB * 2, src, smem + u * QSTAGE, wave, (size_t)d.K * 2, (size_t)d.lda * 2, (size_t)d.ldb * 2);
        if (nslab >= 3 && NSTAGE == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * QPIECES) : "memory");
        else if (nslab >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QPIECES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        qread_frags(smem, off0, wm0, wn0, f0);
        int st = 0;                                          // t % NSTAGE
        for (int t = 0; t < nslab; ++t) {
            char* cur = smem + st * QSTAGE;
            const int st1 = st + 1 == NSTAGE ? 0 : st + 1;
            qread_frags(cur, off1, wm0, wn0, f1);
            qmfma(f0, acc);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < nslab) {
                // this wavefront's pieces of slab t+1 have landed (slab t+2 may stay in flight) and its reads of slab t are in registers;
                // behind the barrier that holds for every wavefront: stage `cur` is free, slab t+1 is readable
                if (NSTAGE == 3 && t + 2 < nslab) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(QPIECES) : "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (t + NSTAGE < nslab && GQ_KNOCK != 1) qissue_slab(A, B, apb, bpb, kb0 + (size_t)(t + NSTAGE) * QSLAB * 2, src, cur, wave, (size_t)d.K * 2, (size_t)d.lda * 2, (size_t)d.ldb * 2);
                qread_frags(smem + st1 * QSTAGE, off0, wm0, wn0, f0);
            }
            __builtin_amdgcn_sched_barrier(0);
            qmfma(f1, acc);
            __builtin_amdgcn_sched_barrier(0);
            st = st1;
        }
    }

    // undo the row scales: accumulator element e of block (i, j) is (m, n) = (wm0 + 32 i + lr, wn0 + 32 j + 8 (e / 4) + 4 lh + e % 4)
    {
        int ea[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) ea[i] = amax_shift(d.a_amax[min(m0 + wm0 + i * 32 + lr, d.M - 1)]);
        const bool b_vec = (reinterpret_cast<uintptr_t>(d.b_amax) & 15) == 0;
#pragma unroll
        for (int j = 0; j < 2; ++j)
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
                    for (int i = 0; i < 2; ++i) acc[i][j][4 * g + q] = __builtin_ldexpf(acc[i][j][4 * g + q], -(ea[i] + eb[q]));
            }
    }
    __syncthreads();                                         // the ring is dead: reuse as epilogue staging
    if (GQ_KNOCK == 3) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(acc[i][j]));
        return;
    }
    tile_epilogue_rows<QBM, QBN, 2, 2>(p, acc, reinterpret_cast<float*>(smem + wave * EPI_WAVE_BYTES), m0, n0, wm0, wn0, lane, split, tile_local,
                                       p.vec_io != 0);
}

template <int NSTAGE>
int launch_pairs(hipStream_t st, GemmParams& p) {
    constexpr size_t ring = (size_t)NSTAGE * QSTAGE, staging = (QNT / 64) * (size_t)EPI_WAVE_BYTES;
    constexpr size_t smem = ring > staging ? ring : staging;
    auto kern = gemm_pairs_kernel<NSTAGE>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return GRAPPA_ERR_LAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(p.ntiles_launch * p.nsplit), dim3(QNT), smem, st, p);
    return grappa_launch_status();
}

// fp32 X[R][C] -> pair format; one 32 x 32 tile per 256-thread workgroup, through LDS when transposing.  amax: bit patterns of the
// largest magnitude of every OUTPUT row (R values, or C values when transposing)
__global__ __launch_bounds__(256) void split_pairs_kernel(int R, int C, const float* __restrict__ x, int ldx, const unsigned* __restrict__ amax,
                                                          uint16_t* __restrict__ out, int ldo, size_t plane_stride, int transpose) {
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
            out[(size_t)r * ldo + c] = hb;
            out[plane_stride + (size_t)r * ldo + c] = lb;
        }
    }
    if (!transpose) return;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = c0 + tr + 8 * q, r = r0 + tc;                  // out[c][r]: consecutive threads -> consecutive r
        if (c < C && r < R) {
            out[(size_t)c * ldo + r] = tile[0][tc][tr + 8 * q];
            out[plane_stride + (size_t)c * ldo + r] = tile[1][tc][tr + 8 * q];
        }
    }
}

}  // namespace

extern "C" int grappa_split_pairs_f32(void* stream, int R, int C, const float* x, int ldx, const uint32_t* amax, uint16_t* pairs, int ldp,
                                      size_t plane_stride, int transpose) {
    if (R < 0 || C < 0) return GRAPPA_ERR_ARG;
    if (R == 0 || C == 0) return GRAPPA_OK;
    if (!x || !amax || !pairs || ldx < C || ldp < (transpose ? R : C)) return GRAPPA_ERR_ARG;
    hipLaunchKernelGGL(split_pairs_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), R, C, x, ldx,
                       amax, pairs, ldp, plane_stride, transpose);
    return grappa_launch_status();
}

// called by grappa_gemm_f32 (gemm_f32.hip) when both operands are in the pair format (precision F32_F16X3); tile 256 x 128
int grappa_launch_gemm_pairs(hipStream_t st, GemmParams& p) {
    static const int nstage = getenv("GRAPPA_PAIRS_STAGES") ? atoi(getenv("GRAPPA_PAIRS_STAGES")) : 3;      // tuning only
    return nstage == 2 ? launch_pairs<2>(st, p) : launch_pairs<3>(st, p);
}
