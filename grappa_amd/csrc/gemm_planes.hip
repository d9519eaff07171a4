// fp32-grade GEMM on the bf16 matrix cores from operands that are ALREADY split ("plane format").
//
// Plane format of an fp32 matrix X[R][C]: three bf16 matrices P0, P1, P2 of the same shape with
//     P0 = bf16(X), P1 = bf16(X - P0), P2 = bf16(X - P0 - P1)            (round to nearest even, residuals exact in fp32)
// so that X == P0 + P1 + P2 exactly (24 significant bits).  A tensor is split ONCE by the kernel that produces it (GEMM epilogue,
// LayerNorm, attention, dropout-backward; weights once per optimiser step) instead of by every workgroup of every GEMM that reads
// it: gemm_bf16x.hip spends ~16 % of its time re-splitting operands that do not change between launches.
//
// With planes in HBM the main loop has no vector arithmetic at all: operand tiles go global -> LDS by LDS-DMA
// (global_load_lds_dwordx4, 16 B per lane, no registers), fragments are single ds_read_b128 (K-contiguous operands) or pairs of
// transposing ds_read_b64_tr_b16 (k-major operands: the weight-gradient layout), and the wavefront issues the 6 partial
// products a0b0, a0b1, a1b0, a1b1, a0b2, a2b0 (smallest first) as v_mfma_f32_32x32x16_bf16 into fp32 accumulators.
//
// Tile 256 x 128, 512 threads = 8 wavefronts as 4 x 2, 64 x 64 per wavefront (transposed accumulators as in gemm_bf16x_impl.h, same
// float4 row epilogue).  K advances in slabs of 32: one stage = 3 planes x (256 + 128) rows x 64 B = 72 KB, two stages.
// LDS image of a stage (linear per plane, which LDS-DMA requires: destination = wave-uniform base + 16 * lane):
//   K-contiguous operand: [row][4 chunks of 16 B]; the chunk a lane fetches is XOR-ed with (row >> 2) & 3 on the SOURCE side and
//                         on the read side, so the 16 lanes of a ds_read_b128 group (rows r..r+3, r+12.., r+20..) hit 16 distinct
//                         16-byte slots of the 256-byte bank row: conflict free.
//   k-major operand:      [k][row chunks of 16 B] (512 B / 256 B per k); chunk index XOR-ed with (k & 3) << 2, so the four k-rows
//                         of a transposing read sit in four different 64-byte bank groups.
// Pipeline per slab t (stage t & 1): the fragments of the slab's second half are read while the first half is multiplied.  Once
// every wavefront holds both halves in REGISTERS (barrier X1, a quarter into the slab) the stage is free and the DMA of slab t+2
// starts into it; three quarters into the slab every wavefront waits for ITS pieces of slab t+1 (counted vmcnt: slab t+2 stays in
// flight), barrier X2, and the first fragments of slab t+1 are read under the remaining MFMAs.  A slab's DMA is in flight for
// ~1.5 slab times with both LDS stages owned by the DMA engine: the registers are the third buffer.
// the straight-line epilogue classes (gemm_common.h) cost this file's one-plane kernel 30 registers, i.e. its second workgroup per CU
// (115 -> 145 VGPRs: the bf16 storage configuration's step went 86 -> 91 ms with them, 99 ms with the registers but without the classes)
#define GRAPPA_NO_FAST_EPI
#define GRAPPA_EPI_RES_LN 0          // (the host refuses res_ln for the plane kernels)
#include <cstdlib>
#include "gemm_common.h"

using namespace grappa_gemm;

int grappa_launch_gemm_bf16_il(hipStream_t st, GemmParams& p);      // gemm_pairs_il.hip

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// experiments only (tools/gemm_planes_check.py --variants): GP_KNOCK 1 = no LDS-DMA after the first two slabs (compute + LDS
// reads alone), 2 = no MFMAs (DMA + LDS reads alone); GP_NOEPI = accumulators are not stored; GP_STAGGER = the workgroups of the
// first round start up to GP_STAGGER microseconds apart.  Never defined in the shipped library.
#ifndef GP_KNOCK
#define GP_KNOCK 0
#endif
#ifndef GP_NOEPI
#define GP_NOEPI 0
#endif
#ifndef GP_STAGGER
#define GP_STAGGER 0
#endif
// GP_PERSIST: the weight-plane kernel walks its tiles with at most 256 resident workgroups (a tile's stores drain under the next
// tile's main loop instead of holding the CU until they are acknowledged)
#ifndef GP_PERSIST
#define GP_PERSIST 0
#endif

namespace {

constexpr int PBM = 256, PBN = 128, PSLAB = 32, PNT = 512;
constexpr int A_PLANE = PBM * PSLAB * 2;      // bytes of one plane of the A tile in a stage (16 KB)
constexpr int B_PLANE = PBN * PSLAB * 2;      // 8 KB

enum PMode { PX1 = 1, PX3 = 3, PX6 = 6, PX9 = 9 };
template <int MODE> struct PPieces { static constexpr int NP = MODE == PX1 ? 1 : (MODE == PX3 ? 2 : 3); };

// 16 bytes of zeros in device memory: the LDS-DMA source of k-rows beyond K (k-major operands need no padding in HBM)
__device__ __attribute__((aligned(16))) unsigned grappa_zero16[4] = {0u, 0u, 0u, 0u};

__device__ inline void glds16(const char* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// per-lane source offsets (bytes, without the slab's uniform part) of the three LDS-DMA pieces a wavefront issues per plane:
// two 1 KB pieces of A (tile row blocks `wave` and `wave + 8`), one of B (row block `wave`)
struct LaneSrc { unsigned a0, a1, b; };

template <bool KMAJOR>
__device__ inline LaneSrc lane_sources(const grappa_gemm_desc& d, int m0, int n0, int wave, int lane) {
    LaneSrc s;
    if (!KMAJOR) {
        // piece = 16 rows x 64 B; lane -> (row = lane >> 2, physical chunk = lane & 3); logical chunk = physical ^ ((row >> 2) & 3)
        const int rin = lane >> 2, c = (lane & 3) ^ ((lane >> 4) & 3);
        const int ra0 = min(m0 + wave * 16 + rin, d.M - 1), ra1 = min(m0 + (wave + 8) * 16 + rin, d.M - 1);
        const int rb = min(n0 + wave * 16 + rin, d.N - 1);
        s.a0 = ((unsigned)ra0 * (unsigned)d.lda + 8u * c) * 2u;
        s.a1 = ((unsigned)ra1 * (unsigned)d.lda + 8u * c) * 2u;
        s.b = ((unsigned)rb * (unsigned)d.ldb + 8u * c) * 2u;
    } else {
        // A piece = 2 k-rows x 512 B: lane -> (k = 2 * piece + (lane >> 5), physical chunk = lane & 31)
        // B piece = 4 k-rows x 256 B: lane -> (k = 4 * piece + (lane >> 4), physical chunk = lane & 15)
        // logical chunk = physical ^ ((k & 3) << 2); columns are clamped inside the (16-byte padded) row
        const int ka0 = 2 * wave + (lane >> 5), ka1 = 2 * (wave + 8) + (lane >> 5), kb = 4 * wave + (lane >> 4);
        const int ja0 = (lane & 31) ^ ((ka0 & 3) << 2), ja1 = (lane & 31) ^ ((ka1 & 3) << 2), jb = (lane & 15) ^ ((kb & 3) << 2);
        const int ca0 = min(m0 + 8 * ja0, d.lda - 8), ca1 = min(m0 + 8 * ja1, d.lda - 8), cb = min(n0 + 8 * jb, d.ldb - 8);
        s.a0 = ((unsigned)ka0 * (unsigned)d.lda + (unsigned)ca0) * 2u;
        s.a1 = ((unsigned)ka1 * (unsigned)d.lda + (unsigned)ca1) * 2u;
        s.b = ((unsigned)kb * (unsigned)d.ldb + (unsigned)cb) * 2u;
    }
    return s;
}

// LDS-DMA of one slab into `stage`: 3 pieces per plane per wavefront
template <int NP>
__device__ inline void issue_slab(const char* __restrict__ A, const char* __restrict__ B, size_t a_plane_bytes, size_t b_plane_bytes, size_t a_uni,
                                  size_t b_uni, const LaneSrc& s, char* __restrict__ stage, int wave) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const char* ap = A + p * a_plane_bytes + a_uni;
        const char* bp = B + p * b_plane_bytes + b_uni;
        glds16(ap + s.a0, stage + p * A_PLANE + wave * 1024);
        glds16(ap + s.a1, stage + p * A_PLANE + (wave + 8) * 1024);
        glds16(bp + s.b, stage + NP * A_PLANE + p * B_PLANE + wave * 1024);
    }
}

// the same for the LAST slab of a k-major K range that is not a multiple of 32: k-rows >= krem are fetched from the zero page
template <int NP>
__device__ inline void issue_slab_ktail(const char* __restrict__ A, const char* __restrict__ B, size_t a_plane_bytes, size_t b_plane_bytes, size_t a_uni,
                                        size_t b_uni, const LaneSrc& s, char* __restrict__ stage, int wave, int lane, int krem) {
    const char* z = reinterpret_cast<const char*>(grappa_zero16);
    const bool oka0 = 2 * wave + (lane >> 5) < krem, oka1 = 2 * (wave + 8) + (lane >> 5) < krem, okb = 4 * wave + (lane >> 4) < krem;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const char* ap = A + p * a_plane_bytes + a_uni;
        const char* bp = B + p * b_plane_bytes + b_uni;
        glds16(oka0 ? ap + s.a0 : z, stage + p * A_PLANE + wave * 1024);
        glds16(oka1 ? ap + s.a1 : z, stage + p * A_PLANE + (wave + 8) * 1024);
        glds16(okb ? bp + s.b : z, stage + NP * A_PLANE + p * B_PLANE + wave * 1024);
    }
}

template <int NP> struct PFrags { bf16x8 a[2][NP], b[2][NP]; };

// per-lane LDS read offsets, fixed for the whole kernel
struct ReadOff {
    unsigned kk[2];          // K-contiguous: byte offset inside a 32-row band for k-half 0 / 1
    unsigned ta[2][2], tb[2][2];   // k-major: [sub tile i / j][read 0 / 1] byte offsets for k-half 0 (k-half 1 adds 16 k-rows)
};

template <bool KMAJOR>
__device__ inline ReadOff read_offsets(int wave, int lane) {
    ReadOff r;
    if (!KMAJOR) {
        const int lr = lane & 31, lh = lane >> 5, swz = (lr >> 2) & 3;
        r.kk[0] = lr * 64 + ((lh ^ swz) << 4);
        r.kk[1] = lr * 64 + (((2 + lh) ^ swz) << 4);
    } else {
        // a 16-lane group reads a block of 4 k-rows x 16 rows: lane 4q + p supplies &(k-row q, rows 4p .. 4p + 3) and receives
        // row (lane & 15) of the four k-rows.  MFMA operand lane (lr, lh): rows lr, k = 8 lh + 0..7 -> two reads (k-rows 8lh + 4rd + q)
        const int g16 = lane >> 4, blk = (g16 & 1) * 16, lh = g16 >> 1, q = (lane & 15) >> 2, pp = lane & 3;
        const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int rd = 0; rd < 2; ++rd) {
                const int k = 8 * lh + 4 * rd + q;
                const int ma = wm0 + i * 32 + blk + 4 * pp, nb = wn0 + i * 32 + blk + 4 * pp;
                r.ta[i][rd] = k * (PBM * 2) + ((((ma >> 3) ^ (q << 2))) << 4) + ((ma & 7) << 1);
                r.tb[i][rd] = k * (PBN * 2) + ((((nb >> 3) ^ (q << 2))) << 4) + ((nb & 7) << 1);
            }
    }
    return r;
}

__device__ inline bf16x8 tr_pair(const char* p0, const char* p1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p1);
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// fragments of k-half `kh` (16 k) of a staged slab
template <int NP, bool KMAJOR>
__device__ inline void read_pfrags(const char* __restrict__ stage, const ReadOff& ro, int kh, int wm0, int wn0, PFrags<NP>& f) {
    const char* a_s = stage;
    const char* b_s = stage + NP * A_PLANE;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        if (!KMAJOR) {
#pragma unroll
            for (int i = 0; i < 2; ++i) f.a[i][p] = *reinterpret_cast<const bf16x8*>(a_s + p * A_PLANE + (wm0 + i * 32) * 64 + ro.kk[kh]);
#pragma unroll
            for (int j = 0; j < 2; ++j) f.b[j][p] = *reinterpret_cast<const bf16x8*>(b_s + p * B_PLANE + (wn0 + j * 32) * 64 + ro.kk[kh]);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                f.a[i][p] = tr_pair(a_s + p * A_PLANE + kh * 16 * (PBM * 2) + ro.ta[i][0], a_s + p * A_PLANE + kh * 16 * (PBM * 2) + ro.ta[i][1]);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                f.b[j][p] = tr_pair(b_s + p * B_PLANE + kh * 16 * (PBN * 2) + ro.tb[j][0], b_s + p * B_PLANE + kh * 16 * (PBN * 2) + ro.tb[j][1]);
        }
    }
}

// MFMAs [LO, HI) of one k-half: products smallest first (pa + pb descending), the 2 x 2 accumulators innermost
template <int MODE, int LO, int HI>
__device__ inline void pmfma_range(const PFrags<PPieces<MODE>::NP>& f, f32x16 (&acc)[2][2]) {
    constexpr int NP = PPieces<MODE>::NP;
    int idx = 0;
#pragma unroll
    for (int s = 2 * (NP - 1); s >= 0; --s) {
#pragma unroll
        for (int pa = 0; pa < NP; ++pa) {
            const int pb = s - pa;
            if (pb < 0 || pb >= NP) continue;
            if (MODE == PX6 && s > 2) continue;
            if (MODE == PX3 && s > 1) continue;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    // B fragment first: the accumulator holds the transposed tile (4 consecutive n per lane, tile_epilogue_rows)
                    if (idx >= LO && idx < HI) {
                        if (GP_KNOCK == 2) asm volatile("" ::"v"(f.b[j][pb]), "v"(f.a[i][pa]));      // keep the fragment reads alive
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.b[j][pb], f.a[i][pa], acc[i][j], 0, 0, 0);
                    }
                    ++idx;
                }
        }
    }
}

// sum of the 8 bf16 of a fragment as fp32 (bias gradient = row sums of the k-major A operand)
__device__ inline float frag_sum(const bf16x8& v) {
    const uint4 u = __builtin_bit_cast(uint4, v);
    const unsigned w[4] = {u.x, u.y, u.z, u.w};
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) s += __uint_as_float(w[e] << 16) + __uint_as_float(w[e] & 0xffff0000u);
    return s;
}

template <int MODE, bool KMAJOR>
__global__ __launch_bounds__(PNT, 2) void gemm_planes_kernel(GemmParams p) {
    constexpr int NP = PPieces<MODE>::NP;
    constexpr int STAGE = NP * (A_PLANE + B_PLANE);
    constexpr int NM = MODE * 4;                  // MFMAs per k-half and wavefront
    extern __shared__ char smem[];
    const grappa_gemm_desc& d = p.d;
    const TileCoord tc = map_workgroup(p);
    const int split = tc.split, tile_local = tc.tile_local, tile_n = tc.tile_n;
    const int m0 = tc.tile_m * PBM, n0 = tile_n * PBN;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
    const int kbeg = split * p.k_per_split;
    const int kend = min(d.K, kbeg + p.k_per_split);
    const int nslab = (kend - kbeg + PSLAB - 1) / PSLAB;       // K ranges are zero-padded to whole slabs by the producer of the planes

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    if (GP_STAGGER > 0 && blockIdx.x < 256) {
        // ~GP_STAGGER us spread over the 32 workgroups that share an XCD (s_sleep 127 = 8128 cycles ~ 3.9 us at 2.1 GHz)
        const int steps = ((blockIdx.x >> 3) & 31) * GP_STAGGER / 4;
        for (int q = 0; q < steps; ++q) __builtin_amdgcn_s_sleep(4);
    }

    const bool do_cs = KMAJOR && d.a_colsum != nullptr && tile_n == 0 && (wave & 1) == 0;
    float cs[2] = {0.f, 0.f};
    if (nslab > 0) {
        const char* A = reinterpret_cast<const char*>(d.A);
        const char* B = reinterpret_cast<const char*>(d.B);
        const size_t apb = d.a_plane_stride * 2, bpb = d.b_plane_stride * 2;
        // uniform byte offset of slab s: K-contiguous rows advance by 64 B, k-major operands by 32 rows
        const size_t a_step = KMAJOR ? (size_t)PSLAB * d.lda * 2 : (size_t)PSLAB * 2;
        const size_t b_step = KMAJOR ? (size_t)PSLAB * d.ldb * 2 : (size_t)PSLAB * 2;
        const size_t a_base = (size_t)kbeg / PSLAB * a_step, b_base = (size_t)kbeg / PSLAB * b_step;
        const LaneSrc src = lane_sources<KMAJOR>(d, m0, n0, wave, lane);
        const ReadOff ro = read_offsets<KMAJOR>(wave, lane);
        PFrags<NP> f0, f1;

        // k-major operands carry no padding: the last slab of a ragged K range takes its missing k-rows from a page of zeros
        const int krem_last = KMAJOR ? (kend - kbeg) - (nslab - 1) * PSLAB : PSLAB;
#define GP_ISSUE(T_, STAGE_)                                                                                                                  \
    do {                                                                                                                                      \
        if (KMAJOR && (T_) == nslab - 1 && krem_last < PSLAB)                                                                                 \
            issue_slab_ktail<NP>(A, B, apb, bpb, a_base + (size_t)(T_) * a_step, b_base + (size_t)(T_) * b_step, src, STAGE_, wave, lane, krem_last); \
        else                                                                                                                                  \
            issue_slab<NP>(A, B, apb, bpb, a_base + (size_t)(T_) * a_step, b_base + (size_t)(T_) * b_step, src, STAGE_, wave);                \
    } while (0)
        GP_ISSUE(0, smem);
        if (nslab > 1) {
            GP_ISSUE(1, smem + STAGE);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NP) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        read_pfrags<NP, KMAJOR>(smem, ro, 0, wm0, wn0, f0);
        for (int t = 0; t < nslab; ++t) {
            char* cur = smem + (t & 1) * STAGE;
            char* nxt = smem + ((t + 1) & 1) * STAGE;
            read_pfrags<NP, KMAJOR>(cur, ro, 1, wm0, wn0, f1);
            pmfma_range<MODE, 0, NM / 2>(f0, acc);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 2 < nslab) {
                // X1: every wavefront holds both halves of slab t in registers -> stage `cur` is free: start the DMA of slab t+2
                // (in flight for ~1.5 slab times: until X2 of slab t+1)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (GP_KNOCK != 1) GP_ISSUE(t + 2, cur);
            }
            __builtin_amdgcn_sched_barrier(0);
            pmfma_range<MODE, NM / 2, NM>(f0, acc);
            if (do_cs) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) cs[i] += frag_sum(f0.a[i][pl]);
            }
            __builtin_amdgcn_sched_barrier(0);
            pmfma_range<MODE, 0, NM / 2>(f1, acc);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < nslab) {
                // X2: this wavefront's pieces of slab t+1 have landed (slab t+2 may stay in flight), then everyone's
                if (t + 2 < nslab && GP_KNOCK != 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NP) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                read_pfrags<NP, KMAJOR>(nxt, ro, 0, wm0, wn0, f0);
            }
            __builtin_amdgcn_sched_barrier(0);
            pmfma_range<MODE, NM / 2, NM>(f1, acc);
            if (do_cs) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) cs[i] += frag_sum(f1.a[i][pl]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

#undef GP_ISSUE
    if (do_cs) {
        // lanes (lr, 0) and (lr, 1) hold the two k-halves of row wm0 + i*32 + lr
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float sum = cs[i] + __shfl_xor(cs[i], 32, 64);
            const int m = m0 + wm0 + i * 32 + (lane & 31);
            if (lane < 32 && m < d.M) {
                if (p.nsplit > 1) p.cs_slab[(size_t)split * d.M + m] = sum;
                else d.a_colsum[m] += sum;
            }
        }
    }
    __syncthreads();                                         // the stages are dead: reuse as epilogue staging
    if (GP_NOEPI) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(acc[i][j]));
        return;
    }
    tile_epilogue_rows<PBM, PBN, 2, 2>(p, acc, reinterpret_cast<float*>(smem + wave * EPI_WAVE_BYTES), m0, n0, wm0, wn0, lane, split, tile_local,
                                       p.vec_io != 0);
}

// ------------------------------------------------------------------------------------------------------------------------------
// "Weight-plane" GEMM: A = fp32 activations [M][K] (K-contiguous, as every producer writes them), B = a weight matrix in the plane
// format (split once per optimiser step: grappa_weight_planes).  Forward (B = planes of W[N][K]) and dgrad (B = planes of the
// TRANSPOSED weight) products.  Both operands reach LDS by LDS-DMA, A as raw fp32 rows of 128 B (whole cache lines); the
// wavefronts are laid out 8 x 1 (each owns 32 rows x all 128 columns of the tile), so every A fragment is read, split into its
// three bf16 pieces and used by exactly ONE wavefront, in registers: 8 floats -> 44 vector instructions per 24 MFMAs, no LDS
// stores, no staging registers, nothing redundant.  (gemm_bf16x.hip splits both operands cooperatively BEFORE the LDS: 78 vector
// instructions + 9 LDS stores per thread for the same 24 MFMAs, and re-splits the weights in every workgroup of every launch.)
// A stage: [256 rows][8 chunks of 16 B], chunk XOR-ed with (row >> 1) & 7 (conflict-free ds_read_b128 by rows); B stage as above.
// Per slab of 32: 32 KB (A) + 24 KB (B) = 56 KB by DMA instead of the 72 KB of two plane operands.  What bounds all these kernels
// is the operand supply: the bytes a CU can keep in flight (its LDS) over the memory latency.  The activations are the long-latency
// operand (HBM / Infinity Cache), the weights sit in L2: so A gets a ring of three stages and is requested up to 2.5 slabs ahead,
// B two stages.
constexpr int WA_BYTES = PBM * PSLAB * 4;     // fp32 A tile of a stage (32 KB)

struct WLaneSrc { unsigned a[4], b; };

__device__ inline WLaneSrc wlane_sources(const grappa_gemm_desc& d, int m0, int n0, int wave, int lane) {
    WLaneSrc s;
    // A piece = 8 rows x 128 B: lane -> (row = lane >> 3, physical chunk = lane & 7); pieces wave, wave + 8, wave + 16, wave + 24
    const int c = (lane & 7) ^ ((((wave & 1) << 2) + (lane >> 4)) & 7);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = min(m0 + (wave + 8 * q) * 8 + (lane >> 3), d.M - 1);
        s.a[q] = ((unsigned)row * (unsigned)d.lda + 4u * c) * 4u;
    }
    const int cb = (lane & 3) ^ ((lane >> 4) & 3);
    const int rb = min(n0 + wave * 16 + (lane >> 2), d.N - 1);
    s.b = ((unsigned)rb * (unsigned)d.ldb + 8u * cb) * 2u;
    return s;
}

template <int NP>
__device__ inline void wissue_b(const char* __restrict__ B, size_t b_plane_bytes, size_t b_uni, const WLaneSrc& s, char* __restrict__ bstage, int wave) {
#pragma unroll
    for (int p = 0; p < NP; ++p) glds16(B + p * b_plane_bytes + b_uni + s.b, bstage + p * B_PLANE + wave * 1024);
}
__device__ inline void wissue_a(const char* __restrict__ A, size_t a_uni, const WLaneSrc& s, char* __restrict__ astage, int wave) {
#pragma unroll
    for (int q = 0; q < 4; ++q) glds16(A + a_uni + s.a[q], astage + (wave + 8 * q) * 1024);
}

template <int NP> struct WFrags { float4 araw[2]; bf16x8 b[4][NP]; };

template <int NP>
__device__ inline void wread_frags(const char* __restrict__ a_s, const char* __restrict__ b_s, unsigned a_off0, unsigned a_off1, unsigned b_off, WFrags<NP>& f) {
    f.araw[0] = *reinterpret_cast<const float4*>(a_s + a_off0);
    f.araw[1] = *reinterpret_cast<const float4*>(a_s + a_off1);
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int j = 0; j < 4; ++j) f.b[j][p] = *reinterpret_cast<const bf16x8*>(b_s + p * B_PLANE + j * 32 * 64 + b_off);
}

// 8 consecutive-k fp32 values -> NP bf16 fragments (round to nearest even, residuals exact in fp32)
template <int NP>
__device__ inline void split_frag(const float4 (&raw)[2], bf16x8 (&a)[NP]) {
    float r[8] = {raw[0].x, raw[0].y, raw[0].z, raw[0].w, raw[1].x, raw[1].y, raw[1].z, raw[1].w};
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        unsigned u[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            grappa_bf16x2 h;
            h[0] = (__bf16)r[2 * e];
            h[1] = (__bf16)r[2 * e + 1];
            u[e] = __builtin_bit_cast(unsigned, h);
            if (p + 1 < NP) {
                r[2 * e] -= __uint_as_float(u[e] << 16);
                r[2 * e + 1] -= __uint_as_float(u[e] & 0xffff0000u);
            }
        }
        a[p] = __builtin_bit_cast(bf16x8, make_uint4(u[0], u[1], u[2], u[3]));
    }
}

template <int MODE, int LO, int HI>
__device__ inline void wmfma_range(const bf16x8 (&a)[PPieces<MODE>::NP], const WFrags<PPieces<MODE>::NP>& f, f32x16 (&acc)[4]) {
    constexpr int NP = PPieces<MODE>::NP;
    int idx = 0;
#pragma unroll
    for (int s = 2 * (NP - 1); s >= 0; --s) {
#pragma unroll
        for (int pa = 0; pa < NP; ++pa) {
            const int pb = s - pa;
            if (pb < 0 || pb >= NP) continue;
            if (MODE == PX6 && s > 2) continue;
            if (MODE == PX3 && s > 1) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (idx >= LO && idx < HI) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.b[j][pb], a[pa], acc[j], 0, 0, 0);
                ++idx;
            }
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(PNT, 2) void gemm_wplanes_kernel(GemmParams p) {
    constexpr int NP = PPieces<MODE>::NP;
    constexpr int BSTAGE = NP * B_PLANE;
    constexpr int NM = MODE * 4;
    extern __shared__ char smem[];
    // LDS: a ring of THREE A stages (activations come from HBM / the Infinity Cache: long latency, so they are requested up to
    // 2.5 slabs ahead) and two B stages (the weights stay in the XCD's L2)
    char* const a_ring = smem;
    char* const b_ring = smem + 3 * WA_BYTES;
    const grappa_gemm_desc& d = p.d;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wm0 = wave * 32;
    const int total_wgs = p.ntiles_launch * p.nsplit;
  for (int wg = blockIdx.x; wg < total_wgs; wg += gridDim.x) {
    if (wg != (int)blockIdx.x) __syncthreads();              // the previous tile's epilogue staging is done with the LDS
    const TileCoord tc = map_logical(p, total_wgs, wg);
    const int split = tc.split, tile_local = tc.tile_local;
    const int m0 = tc.tile_m * PBM, n0 = tc.tile_n * PBN;
    const int kbeg = split * p.k_per_split;
    const int kend = min(d.K, kbeg + p.k_per_split);
    const int nslab = (kend - kbeg + PSLAB - 1) / PSLAB;

    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;

    if (nslab > 0) {
        const char* A = reinterpret_cast<const char*>(d.A);
        const char* B = reinterpret_cast<const char*>(d.B);
        const size_t bpb = d.b_plane_stride * 2;
        const size_t a_base = (size_t)kbeg * 4, b_base = (size_t)kbeg * 2;
        const WLaneSrc src = wlane_sources(d, m0, n0, wave, lane);
        const int lr = lane & 31, lh = lane >> 5;
        // A: row wm0 + lr, floats 16 kh + 8 lh .. + 7 = chunks 4 kh + 2 lh, + 1, XOR-ed with (row >> 1) & 7
        const unsigned arow = (unsigned)(wm0 + lr) * 128u, aswz = (lr >> 1) & 7;
        unsigned aoff[2][2];
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int e = 0; e < 2; ++e) aoff[kh][e] = arow + ((((4 * kh + 2 * lh + e) ^ aswz)) << 4);
        const unsigned bswz = (lr >> 2) & 3;
        const unsigned boff[2] = {(unsigned)lr * 64u + ((lh ^ bswz) << 4), (unsigned)lr * 64u + (((2 + lh) ^ bswz) << 4)};
        WFrags<NP> f0, f1;
        bf16x8 a0[NP], a1[NP];

        // issue order (the wait counts below rely on it): B0 A0 [B1 A1] [A2] | X1(t): B(t+2) A(t+3)
        wissue_b<NP>(B, bpb, b_base, src, b_ring, wave);
        wissue_a(A, a_base, src, a_ring, wave);
        if (nslab > 1) {
            wissue_b<NP>(B, bpb, b_base + PSLAB * 2, src, b_ring + BSTAGE, wave);
            wissue_a(A, a_base + PSLAB * 4, src, a_ring + WA_BYTES, wave);
        }
        if (nslab > 2) wissue_a(A, a_base + 2 * PSLAB * 4, src, a_ring + 2 * WA_BYTES, wave);
        if (nslab > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP + 8) : "memory");
        else if (nslab > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP + 4) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        wread_frags<NP>(a_ring, b_ring, aoff[0][0], aoff[0][1], boff[0], f0);
        int ar = 0;                                      // t % 3
        for (int t = 0; t < nslab; ++t) {
            char* a_cur = a_ring + ar * WA_BYTES;
            char* b_cur = b_ring + (t & 1) * BSTAGE;
            const int ar1 = ar == 2 ? 0 : ar + 1;
            wread_frags<NP>(a_cur, b_cur, aoff[1][0], aoff[1][1], boff[1], f1);
            split_frag<NP>(f0.araw, a0);
            wmfma_range<MODE, 0, NM / 2>(a0, f0, acc);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 2 < nslab) {
                // X1: both halves of slab t are in registers in every wavefront -> its stages are free: B of slab t+2, A of slab t+3
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                wissue_b<NP>(B, bpb, b_base + (size_t)(t + 2) * PSLAB * 2, src, b_cur, wave);
                if (t + 3 < nslab) wissue_a(A, a_base + (size_t)(t + 3) * PSLAB * 4, src, a_cur, wave);
            }
            __builtin_amdgcn_sched_barrier(0);
            wmfma_range<MODE, NM / 2, NM>(a0, f0, acc);
            split_frag<NP>(f1.araw, a1);
            wmfma_range<MODE, 0, NM / 2>(a1, f1, acc);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < nslab) {
                // X2: this wavefront's pieces of slab t+1 have landed; younger requests stay in flight: A(t+2), B(t+2), A(t+3)
                if (t + 3 < nslab) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP + 8) : "memory");
                else if (t + 2 < nslab) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP + 4) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                wread_frags<NP>(a_ring + ar1 * WA_BYTES, b_ring + ((t + 1) & 1) * BSTAGE, aoff[0][0], aoff[0][1], boff[0], f0);
            }
            __builtin_amdgcn_sched_barrier(0);
            wmfma_range<MODE, NM / 2, NM>(a1, f1, acc);
            __builtin_amdgcn_sched_barrier(0);
            ar = ar1;
        }
    }
    __syncthreads();                                         // the stages are dead: reuse as epilogue staging
    // the wavefront's 32 x 128 block as two 32 x 64 bands of the row epilogue
    float* wave_buf = reinterpret_cast<float*>(smem + wave * EPI_WAVE_BYTES);
    const bool vec_io = p.vec_io != 0;
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
        const f32x16 band[2] = {acc[2 * h], acc[2 * h + 1]};
        epilogue_band<PBM, PBN, 2>(p, band, wave_buf, m0, n0, m0 + wm0, n, lane, b4, split, tile_local, vec_io);
    }
  }
}

template <int MODE>
int launch_wplanes(hipStream_t st, GemmParams& p) {
    constexpr size_t stages = 3 * (size_t)WA_BYTES + 2 * PPieces<MODE>::NP * B_PLANE, staging = (PNT / 64) * (size_t)EPI_WAVE_BYTES;
    constexpr size_t smem = stages > staging ? stages : staging;
    auto kern = gemm_wplanes_kernel<MODE>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return GRAPPA_ERR_LAUNCH;
        attr_set = true;
    }
    const int total = p.ntiles_launch * p.nsplit;
    GRAPPA_LAUNCH(kern, dim3(GP_PERSIST && total > 256 ? 256 : total), dim3(PNT), smem, st, p);
    return grappa_launch_status();
}

template <int MODE, bool KMAJOR>
int launch_planes(hipStream_t st, GemmParams& p) {
    constexpr size_t stages = 2 * (size_t)PPieces<MODE>::NP * (A_PLANE + B_PLANE), staging = (PNT / 64) * (size_t)EPI_WAVE_BYTES;
    constexpr size_t smem = stages > staging ? stages : staging;
    auto kern = gemm_planes_kernel<MODE, KMAJOR>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return GRAPPA_ERR_LAUNCH;
        attr_set = true;
    }
    GRAPPA_LAUNCH(kern, dim3(p.ntiles_launch * p.nsplit), dim3(PNT), smem, st, p);
    return grappa_launch_status();
}

template <int MODE>
int launch_planes_layout(hipStream_t st, GemmParams& p) {
    return p.d.a_kcontig ? launch_planes<MODE, false>(st, p) : launch_planes<MODE, true>(st, p);
}

}  // namespace

namespace {
// fp32 X[R][C] -> bf16 planes; one 32 x 32 tile per 256-thread workgroup (4 elements per thread), through LDS when transposing
__global__ __launch_bounds__(256) void split_planes_kernel(int R, int C, const float* __restrict__ x, int ldx, uint16_t* __restrict__ out, int ldo,
                                                           size_t plane_stride, int transpose) {
    __shared__ uint16_t tile[3][32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tc = threadIdx.x & 31, tr = threadIdx.x >> 5;          // 8 rows per pass
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = r0 + tr + 8 * q, c = c0 + tc;
        float v = (r < R && c < C) ? x[(size_t)r * ldx + c] : 0.f;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            const __bf16 h = (__bf16)v;
            const uint16_t bits = __builtin_bit_cast(uint16_t, h);
            v -= (float)h;
            if (transpose) tile[pl][tr + 8 * q][tc] = bits;
            else if (r < R && c < C) out[pl * plane_stride + (size_t)r * ldo + c] = bits;
        }
    }
    if (!transpose) return;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = c0 + tr + 8 * q, r = r0 + tc;                  // out[c][r]: consecutive threads -> consecutive r
        if (c < C && r < R) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) out[pl * plane_stride + (size_t)c * ldo + r] = tile[pl][tc][tr + 8 * q];
        }
    }
}
}  // namespace

extern "C" int grappa_split_planes_f32(void* stream, int R, int C, const float* x, int ldx, uint16_t* planes, int ldp, size_t plane_stride,
                                       int transpose) {
    if (R < 0 || C < 0) return GRAPPA_ERR_ARG;
    if (R == 0 || C == 0) return GRAPPA_OK;
    if (!x || !planes || ldx < C || ldp < (transpose ? R : C)) return GRAPPA_ERR_ARG;
    GRAPPA_LAUNCH(split_planes_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), R, C, x, ldx,
                       planes, ldp, plane_stride, transpose);
    return grappa_launch_status();
}

// called by grappa_gemm_f32 (gemm_f32.hip) when B (only) or both operands are in the plane format; tile 256 x 128
int grappa_launch_gemm_planes(hipStream_t st, GemmParams& p, int precision) {
    if (!p.d.a_planes) {
        switch (precision) {
            case GRAPPA_GEMM_F32_BF16X9: return launch_wplanes<PX9>(st, p);
            case GRAPPA_GEMM_F32_MFMA:
            case GRAPPA_GEMM_F32_BF16X6: return launch_wplanes<PX6>(st, p);
            case GRAPPA_GEMM_BF16X3: return launch_wplanes<PX3>(st, p);
            case GRAPPA_GEMM_BF16: return launch_wplanes<PX1>(st, p);
            default: return GRAPPA_ERR_ARG;
        }
    }
    switch (precision) {
        case GRAPPA_GEMM_F32_BF16X9: return launch_planes_layout<PX9>(st, p);
        case GRAPPA_GEMM_F32_MFMA:                      // plane operands carry no fp32 copy: the fp32-grade product is the x6 one
        case GRAPPA_GEMM_F32_BF16X6: return launch_planes_layout<PX6>(st, p);
        case GRAPPA_GEMM_BF16X3: return launch_planes_layout<PX3>(st, p);
        case GRAPPA_GEMM_BF16: {
            // round 5: K-contiguous one-plane products take the pinned pipeline of gemm_pairs_il.hip
            const int kk = p.d.K - (p.nsplit - 1) * p.k_per_split;       // the shortest K range of the launch (the last): it has to hold the pipeline's four slabs too
            if (p.d.a_kcontig && p.d.b_kcontig && ((p.bm == 256 && (p.bn == 128 || p.bn == 256)) || (p.bm == 128 && p.bn == 128)) && (p.d.K & 63) == 0 && (p.k_per_split & 63) == 0 && kk >= 128 &&
                (size_t)p.d.M * p.d.lda * 2 < (1ull << 32) && (size_t)p.d.N * p.d.ldb * 2 < (1ull << 32))
                return grappa_launch_gemm_bf16_il(st, p);
            if (p.bn != 128 || p.bm != 256) return GRAPPA_ERR_ARG;              // (the plane kernel has the 256 x 128 tile only: the plan and this check agree by construction)
            return launch_planes_layout<PX1>(st, p);
        }
        default: return GRAPPA_ERR_ARG;
    }
}
