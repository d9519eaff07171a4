// fp32 GEMM emulated on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, 16x the rate of the native fp32 MFMA) by operand
// splitting: every fp32 operand is written as a sum of bf16 pieces, a = a0 + a1 + a2 with a0 = bf16(a), a1 = bf16(a - a0),
// a2 = bf16(a - a0 - a1) (each subtraction exact in fp32; three round-to-nearest pieces carry 24 significant bits), and the
// product is assembled from bf16 x bf16 partial products accumulated in fp32:
//   X9 : all 9 a_p*b_q -- every partial product is exact in fp32, so only the accumulation order differs from an fp32 FMA chain
//   X6 : drops a1*b2, a2*b1, a2*b2 (each <= 2^-24 |a||b|): ~2 ulp of fp32 per product
//   X3 : two pieces, a0*b0 + a0*b1 + a1*b0 (~2^-16 relative);  X1: plain bf16 operands (bf16 compute configs)
// Structure: 512-thread workgroup (8 wavefronts as 2 x 4, 64x32 accumulators each), 128x128x32 tile, one workgroup per CU.
// Global fp32 -> registers (prefetch distance 2) -> split to bf16 pieces -> LDS [piece][row][32 bf16 + 16 B pad] (two stages,
// 120 KB), MFMA operand fragments are single ds_read_b128 (8 consecutive k of one row).  Row-contiguous ("k-major") sources are
// transposed on the way in: a thread owns one row and reads its 4 k values with 4 row-coalesced dword loads.
// Epilogue, split-K slabs, XCD-aware order and the tail launch are shared with the native kernel (gemm_common.h).
#include "gemm_common.h"

using namespace grappa_gemm;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK = GEMM_BK;
constexpr int NT = 512;
constexpr int BM = 128, BN = 128;
constexpr int WAVES_N = 4;              // 2 x 4 wavefronts
constexpr int TM = 2, TN = 1;           // 32x32 accumulators per wavefront: 64 x 32
constexpr int ROWB = 80;                // LDS row: 32 bf16 (64 B) + 16 B pad -> conflict-free ds_read_b128 over 16 consecutive rows
constexpr int PIECE_B = BM * ROWB;      // one bf16 piece of one operand tile
constexpr int NQ = BM * BK / 4 / NT;    // quads (4 consecutive k of one row) per thread and operand = 2

enum Mode { X1 = 1, X3 = 3, X6 = 6, X9 = 9 };
template <int MODE> struct Pieces { static constexpr int NP = MODE == X1 ? 1 : (MODE == X3 ? 2 : 3); };

struct Quad { float x[4]; };

// element (row, kq..kq+3) owned by this thread for quad slot i
template <bool KCONT>
__device__ inline void quad_coords(int i, int& row, int& kq) {
    const int f = threadIdx.x + i * NT;
    if (KCONT) { row = f >> 3; kq = (f & 7) << 2; }
    else { row = f & (BM - 1); kq = (f >> 7) << 2; }
}

// Loads never wait for their data: out-of-range rows and k are only CLAMPED here (the addresses stay inside the operand);
// the k tail is zeroed when the tile is split and stored (store_quads<MASK>), one pipeline step later.
template <bool KCONT, bool VEC>
__device__ inline void load_quads(const float* __restrict__ src, int ld, int row0, int k0, int R, int Kend, Quad (&q)[NQ]) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        int row, kq;
        quad_coords<KCONT>(i, row, kq);
        const int gr = min(row0 + row, R - 1);          // out-of-range rows are clamped: their results are never stored
        const int gk = k0 + kq;
        if (KCONT && VEC) {
            const float4 v = *reinterpret_cast<const float4*>(src + (size_t)gr * ld + (gk < Kend ? gk : 0));   // K % 4 == 0 here
            q[i].x[0] = v.x; q[i].x[1] = v.y; q[i].x[2] = v.z; q[i].x[3] = v.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ck = gk + j < Kend ? gk + j : 0;
                q[i].x[j] = KCONT ? src[(size_t)gr * ld + ck] : src[(size_t)ck * ld + gr];
            }
        }
    }
}

__device__ inline unsigned f2u(float x) { return __float_as_uint(x); }
__device__ inline float u2f(unsigned x) { return __uint_as_float(x); }

// split 4 consecutive-k fp32 values into NP bf16 pieces (round to nearest even; the residual r - float(piece) is exact in
// fp32) and store each piece's 4 values as one 8-byte LDS write.  krem = valid k of this tile counted from its first column
// (MASK: the last tile of a K range, whose tail is zero-filled here).
template <int NP, bool KCONT, bool MASK>
__device__ inline void store_quads(char* __restrict__ opbase, const Quad (&q)[NQ], int krem) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        int row, kq;
        quad_coords<KCONT>(i, row, kq);
        float r[4] = {q[i].x[0], q[i].x[1], q[i].x[2], q[i].x[3]};
        if (MASK) {
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = kq + j < krem ? r[j] : 0.f;
        }
        char* dst = opbase + row * ROWB + kq * 2;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            bf16x2 h01, h23;
            h01[0] = (__bf16)r[0]; h01[1] = (__bf16)r[1];
            h23[0] = (__bf16)r[2]; h23[1] = (__bf16)r[3];
            const unsigned u01 = __builtin_bit_cast(unsigned, h01), u23 = __builtin_bit_cast(unsigned, h23);
            *reinterpret_cast<uint2*>(dst + p * PIECE_B) = make_uint2(u01, u23);
            if (p + 1 < NP) {                       // float(bf16) is the 16 bits moved to the top half of the word
                r[0] -= u2f(u01 << 16); r[1] -= u2f(u01 & 0xffff0000u);
                r[2] -= u2f(u23 << 16); r[3] -= u2f(u23 & 0xffff0000u);
            }
        }
    }
}

template <int NP>
__device__ inline bf16x8 read_frag(const char* __restrict__ opbase, int piece, int row, int ks, int lh) {
    return *reinterpret_cast<const bf16x8*>(opbase + piece * PIECE_B + row * ROWB + (ks * 16 + 8 * lh) * 2);
}

template <int NP> struct Frags { bf16x8 a[TM][NP], b[TN][NP]; };

// MFMA operand fragments of one 16-deep k-step of a staged tile (18 ds_read_b128 for three pieces)
template <int NP>
__device__ inline void read_frags(const char* __restrict__ a_s, const char* __restrict__ b_s, int wm0, int wn0, int lr, int lh, int ks,
                                  Frags<NP>& f) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int i = 0; i < TM; ++i) f.a[i][p] = read_frag<NP>(a_s, p, wm0 + i * 32 + lr, ks, lh);
#pragma unroll
        for (int j = 0; j < TN; ++j) f.b[j][p] = read_frag<NP>(b_s, p, wn0 + j * 32 + lr, ks, lh);
    }
}

// all partial products of the mode for one k-step, smallest terms first; consecutive MFMAs go to different accumulators
template <int MODE>
__device__ inline void mfma_kstep(const Frags<Pieces<MODE>::NP>& f, f32x16 (&acc)[TM][TN]) {
    constexpr int NP = Pieces<MODE>::NP;
#pragma unroll
    for (int s = 2 * (NP - 1); s >= 0; --s) {                  // s = pa + pb, descending: smallest partial products first
#pragma unroll
        for (int pa = 0; pa < NP; ++pa) {
            const int pb = s - pa;
            if (pb < 0 || pb >= NP) continue;
            if (MODE == X6 && s > 2) continue;                  // X6 keeps pa + pb <= 2
            if (MODE == X3 && s > 1) continue;                  // X3 keeps pa + pb <= 1
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][pa], f.b[j][pb], acc[i][j], 0, 0, 0);
        }
    }
}

template <int MODE, bool AK, bool BKC, bool VEC>
__device__ inline void load_pair(const grappa_gemm_desc& d, int m0, int n0, int k0, int kend, Quad (&qa)[NQ], Quad (&qb)[NQ]) {
    load_quads<AK, VEC>(d.A, d.lda, m0, k0, d.M, kend, qa);
    load_quads<BKC, VEC>(d.B, d.ldb, n0, k0, d.N, kend, qb);
}

__device__ inline float quad_sum(const Quad (&q)[NQ]) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NQ; ++i) s += (q[i].x[0] + q[i].x[1]) + (q[i].x[2] + q[i].x[3]);
    return s;
}

template <bool KCONT>
__device__ inline float quad_sum_masked(const Quad (&q)[NQ], int krem) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        int row, kq;
        quad_coords<KCONT>(i, row, kq);
#pragma unroll
        for (int j = 0; j < 4; ++j) s += kq + j < krem ? q[i].x[j] : 0.f;
    }
    return s;
}

// One K-step of 32.  On entry f0 holds the fragments of the first 16 k of stage `cur`; on exit those of stage cur^1.
//   global loads of tile kt+2 -> L | fragment reads of the second 16 k | MFMAs on f0, interleaved with the bf16 split + LDS store
//   of tile kt+1 (registers S) | barrier | fragment reads of the next stage | MFMAs on f1 (cover that read latency)
// so the matrix pipe has work queued on both sides of the barrier.  TAIL = one of the last steps of the K range: loads / stores
// happen only while tiles remain and the stored tile is masked to the valid k; main-loop steps do both unconditionally.
struct KRange {
    int kbeg, kend, nk, rot;
    // first k of the tile processed at step t.  The K loop of a workgroup starts `rot` tiles into its range and wraps: workgroups
    // that run side by side then read different 128-byte columns of the operands at any moment, instead of all hitting the L2
    // channels that one column's addresses (rows 2^n bytes apart) map to.  rot == 0 when the range has a partial last tile.
    __device__ inline int k_of(int t) const {
        int x = t + rot;
        x = x >= nk ? x - nk : x;
        return kbeg + x * BK;
    }
};

template <int MODE, bool AK, bool BKC, bool VEC, bool TAIL>
__device__ inline void pipeline_step(const grappa_gemm_desc& d, char* __restrict__ smem, f32x16 (&acc)[TM][TN], int m0, int n0, const KRange& kr,
                                     int kt, int wm0, int wn0, int lr, int lh, Quad (&la)[NQ], Quad (&lb)[NQ],
                                     const Quad (&sa)[NQ], const Quad (&sb)[NQ], Frags<Pieces<MODE>::NP>& f0, float& cs, bool do_cs) {
    constexpr int NP = Pieces<MODE>::NP;
    constexpr int OPB = NP * PIECE_B, STAGE = 2 * OPB;
    const int cur = kt & 1;
    const bool do_store = !TAIL || kt + 1 < kr.nk;                // tile kt+1 exists
    const bool do_load = !TAIL || kt + 2 < kr.nk;                 // tile kt+2 exists
    if (do_load) load_pair<MODE, AK, BKC, VEC>(d, m0, n0, kr.k_of(kt + 2), kr.kend, la, lb);
    __builtin_amdgcn_sched_barrier(0);
    const char* a_s = smem + cur * STAGE;
    char* nxt = smem + (cur ^ 1) * STAGE;
    Frags<NP> f1;
    read_frags<NP>(a_s, a_s + OPB, wm0, wn0, lr, lh, 1, f1);
    mfma_kstep<MODE>(f0, acc);
    if (do_store) {
        const int krem = TAIL ? kr.kend - kr.k_of(kt + 1) : BK;
        store_quads<NP, AK, TAIL>(nxt, sa, krem);
        store_quads<NP, BKC, TAIL>(nxt + OPB, sb, krem);
        if (!AK && do_cs) cs += TAIL ? quad_sum_masked<AK>(sa, krem) : quad_sum(sa);
    }
    __syncthreads();
    if (do_store) read_frags<NP>(nxt, nxt + OPB, wm0, wn0, lr, lh, 0, f0);
    mfma_kstep<MODE>(f1, acc);
}

template <int MODE, bool AK, bool BKC, bool VEC>
__global__ __launch_bounds__(NT) void gemm_bf16x_kernel(GemmParams p) {
    constexpr int NP = Pieces<MODE>::NP;
    constexpr int OPB = NP * PIECE_B;
    extern __shared__ char smem[];
    const grappa_gemm_desc& d = p.d;
    const TileCoord tc = map_workgroup(p);
    const int split = tc.split, tile_local = tc.tile_local, tile_n = tc.tile_n;
    const int m0 = tc.tile_m * BM, n0 = tile_n * BN;
    const int kbeg = split * p.k_per_split;
    const int kend = min(d.K, kbeg + p.k_per_split);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm0 = (wave / WAVES_N) * 64, wn0 = (wave % WAVES_N) * 32;
    const int lr = lane & 31, lh = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    KRange kr;
    kr.kbeg = kbeg;
    kr.kend = kend;
    kr.nk = (kend - kbeg + BK - 1) / BK;
    kr.rot = 0;     // measured: a per-workgroup rotation gains 10-15 % on K-contiguous bf16 operands but loses 25 % on the wgrad layout
    const int nk = kr.nk;
    const bool do_cs = !AK && d.a_colsum != nullptr && tile_n == 0;
    float cs = 0.f;                    // sum over k of A(row, k) for the row this thread stages (row = tid & 127 for both quads)
    if (nk > 0) {
        Quad a0[NQ], b0[NQ], a1[NQ], b1[NQ];
        Frags<NP> f0;
        load_pair<MODE, AK, BKC, VEC>(d, m0, n0, kr.k_of(0), kend, a0, b0);
        load_pair<MODE, AK, BKC, VEC>(d, m0, n0, nk > 1 ? kr.k_of(1) : kbeg, kend, a1, b1);
        store_quads<NP, AK, true>(smem, a0, kend - kr.k_of(0));
        store_quads<NP, BKC, true>(smem + OPB, b0, kend - kr.k_of(0));
        if (!AK && do_cs) cs += quad_sum_masked<AK>(a0, kend - kr.k_of(0));
        __syncthreads();
        read_frags<NP>(smem, smem + OPB, wm0, wn0, lr, lh, 0, f0);
        int kt = 0;
#define GRAPPA_STEP(TAIL, LA, LB, SA, SB) \
    pipeline_step<MODE, AK, BKC, VEC, TAIL>(d, smem, acc, m0, n0, kr, kt, wm0, wn0, lr, lh, LA, LB, SA, SB, f0, cs, do_cs)
        // main loop: two steps per trip (the register sets swap roles), never touching the last tile of the range
        for (; kt + 3 < nk; kt += 2) {
            GRAPPA_STEP(false, a0, b0, a1, b1);
            ++kt;
            GRAPPA_STEP(false, a1, b1, a0, b0);
            --kt;
        }
        for (; kt < nk; kt += 2) {                    // the last (up to 3) steps
            GRAPPA_STEP(true, a0, b0, a1, b1);
            if (kt + 1 < nk) {
                ++kt;
                GRAPPA_STEP(true, a1, b1, a0, b0);
                --kt;
            }
        }
#undef GRAPPA_STEP
    }

    if (!AK && do_cs) {
        __syncthreads();                                     // every wavefront is past its last fragment read
        float* red = reinterpret_cast<float*>(smem);
        red[threadIdx.x] = cs;
        __syncthreads();
        if (threadIdx.x < BM) {
            const float s = (red[threadIdx.x] + red[threadIdx.x + BM]) + (red[threadIdx.x + 2 * BM] + red[threadIdx.x + 3 * BM]);
            const int m = m0 + threadIdx.x;
            if (m < d.M) {
                if (p.nsplit > 1) p.cs_slab[(size_t)split * d.M + m] = s;
                else d.a_colsum[m] += s;
            }
        }
    }
    tile_epilogue<BM, BN, TM, TN>(p, acc, m0, n0, wm0, wn0, lr, lh, split, tile_local);
}

template <int MODE, bool AK, bool BKC, bool VEC>
int launch_mode(hipStream_t st, GemmParams& p) {
    constexpr size_t smem = 2 * 2 * (size_t)Pieces<MODE>::NP * PIECE_B;
    auto kern = gemm_bf16x_kernel<MODE, AK, BKC, VEC>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return GRAPPA_ERR_LAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(p.ntiles_launch * p.nsplit), dim3(NT), smem, st, p);
    return grappa_launch_status();
}

template <int MODE>
int launch_layout(hipStream_t st, GemmParams& p, bool vec) {
    const grappa_gemm_desc& d = p.d;
    if (d.a_kcontig && d.b_kcontig) return vec ? launch_mode<MODE, true, true, true>(st, p) : launch_mode<MODE, true, true, false>(st, p);
    if (d.a_kcontig) return vec ? launch_mode<MODE, true, false, true>(st, p) : launch_mode<MODE, true, false, false>(st, p);
    return launch_mode<MODE, false, false, false>(st, p);     // row-contiguous operands never use the float4 path
}

}  // namespace

// called by grappa_gemm_f32 (gemm_f32.hip) for precision != GRAPPA_GEMM_F32_MFMA; p.bm == p.bn == 128
int grappa_launch_gemm_bf16x(hipStream_t st, GemmParams& p, int precision, bool vec_kcontig) {
    switch (precision) {
        case GRAPPA_GEMM_F32_BF16X9: return launch_layout<X9>(st, p, vec_kcontig);
        case GRAPPA_GEMM_F32_BF16X6: return launch_layout<X6>(st, p, vec_kcontig);
        case GRAPPA_GEMM_BF16X3: return launch_layout<X3>(st, p, vec_kcontig);
        case GRAPPA_GEMM_BF16: return launch_layout<X1>(st, p, vec_kcontig);
        default: return GRAPPA_ERR_ARG;
    }
}
