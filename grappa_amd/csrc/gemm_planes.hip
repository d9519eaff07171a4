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
// Tile 256 x 128, 512 threads = 8 wavefronts as 4 x 2, 64 x 64 per wavefront (transposed accumulators as in gemm_bf16x.hip, same
// float4 row epilogue).  K advances in slabs of 32: one stage = 3 planes x (256 + 128) rows x 64 B = 72 KB, two stages.
// LDS image of a stage (linear per plane, which LDS-DMA requires: destination = wave-uniform base + 16 * lane):
//   K-contiguous operand: [row][4 chunks of 16 B]; the chunk a lane fetches is XOR-ed with (row >> 2) & 3 on the SOURCE side and
//                         on the read side, so the 16 lanes of a ds_read_b128 group (rows r..r+3, r+12.., r+20..) hit 16 distinct
//                         16-byte slots of the 256-byte bank row: conflict free.
//   k-major operand:      [k][row chunks of 16 B] (512 B / 256 B per k); chunk index XOR-ed with (k & 3) << 2, so the four k-rows
//                         of a transposing read sit in four different 64-byte bank groups.
// Pipeline per slab t (stage t & 1): the fragments of the slab's second half are read while the first half is multiplied; in
// the middle of the second half every wavefront waits for ITS LDS-DMA of slab t+1 (vmcnt(0)), one barrier, then the DMA of slab
// t+2 is issued into the stage just drained and the first fragments of slab t+1 are read under the remaining MFMAs.
#include "gemm_common.h"

using namespace grappa_gemm;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int PBM = 256, PBN = 128, PSLAB = 32, PNT = 512;
constexpr int A_PLANE = PBM * PSLAB * 2;      // bytes of one plane of the A tile in a stage (16 KB)
constexpr int B_PLANE = PBN * PSLAB * 2;      // 8 KB

enum PMode { PX1 = 1, PX3 = 3, PX6 = 6, PX9 = 9 };
template <int MODE> struct PPieces { static constexpr int NP = MODE == PX1 ? 1 : (MODE == PX3 ? 2 : 3); };

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
                    if (idx >= LO && idx < HI) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.b[j][pb], f.a[i][pa], acc[i][j], 0, 0, 0);
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

        issue_slab<NP>(A, B, apb, bpb, a_base, b_base, src, smem, wave);
        if (nslab > 1) {
            issue_slab<NP>(A, B, apb, bpb, a_base + a_step, b_base + b_step, src, smem + STAGE, wave);
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
            pmfma_range<MODE, 0, NM>(f0, acc);
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
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wavefront's pieces of slab t+1 have landed
                __builtin_amdgcn_s_barrier();                              // ... everyone's; and every wavefront is done reading `cur`
                __builtin_amdgcn_sched_barrier(0);
                if (t + 2 < nslab) issue_slab<NP>(A, B, apb, bpb, a_base + (t + 2) * a_step, b_base + (t + 2) * b_step, src, cur, wave);
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
    tile_epilogue_rows<PBM, PBN, 2, 2>(p, acc, reinterpret_cast<float*>(smem + wave * EPI_WAVE_BYTES), m0, n0, wm0, wn0, lane, split, tile_local,
                                       p.vec_io != 0);
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
    hipLaunchKernelGGL(kern, dim3(p.ntiles_launch * p.nsplit), dim3(PNT), smem, st, p);
    return grappa_launch_status();
}

template <int MODE>
int launch_planes_layout(hipStream_t st, GemmParams& p) {
    return p.d.a_kcontig ? launch_planes<MODE, false>(st, p) : launch_planes<MODE, true>(st, p);
}

}  // namespace

// called by grappa_gemm_f32 (gemm_f32.hip) when both operands are in the plane format; tile 256 x 128
int grappa_launch_gemm_planes(hipStream_t st, GemmParams& p, int precision) {
    switch (precision) {
        case GRAPPA_GEMM_F32_BF16X9: return launch_planes_layout<PX9>(st, p);
        case GRAPPA_GEMM_F32_MFMA:                      // plane operands carry no fp32 copy: the fp32-grade product is the x6 one
        case GRAPPA_GEMM_F32_BF16X6: return launch_planes_layout<PX6>(st, p);
        case GRAPPA_GEMM_BF16X3: return launch_planes_layout<PX3>(st, p);
        case GRAPPA_GEMM_BF16: return launch_planes_layout<PX1>(st, p);
        default: return GRAPPA_ERR_ARG;
    }
}
