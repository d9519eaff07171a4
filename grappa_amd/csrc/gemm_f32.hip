// fp32 GEMM with fused epilogue on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact f32,
// bitwise a k-ordered fmaf chain; peak 157.3 TFLOP/s on MI355X).
//
// Structure: 256-thread workgroup (4 wavefronts), BMxBNx32 tile, each wavefront owns a (BM/WAVES_M)x(BN/WAVES_N)
// sub-tile as 32x32 MFMA accumulators.  Software pipeline per K-step of 32 (register prefetch distance 2, two LDS stages):
//   [issue the 16-byte global loads of tile kt+2] [MFMAs on the first half of tile kt] [LDS stores of tile kt+1, loaded one
//   step earlier -> no exposed wait] [MFMAs on the second half] [one barrier]
// pinned with scheduling barriers (hipcc otherwise sinks the loads below the MFMA block and exposes the memory latency).
// Operands sit in LDS so that the 64 lanes of one MFMA operand read hit 64 different banks:
//   K-contiguous source  (X[M,K], W[N,K], dY[M,N']): S[row][32+1]  (pad 1 -> bank = (row+k) % 32)
//   row-contiguous source (W[N',K'] for dgrad, dY/X for wgrad): S[k][BROW]
// Ragged M/N edges cost nothing in the main loop: out-of-range rows are CLAMPED to a valid row (their results are never
// stored), out-of-range K is zero-selected; only shapes whose leading dimension forbids 16-byte loads (K = 85, ld = 511)
// take the scalar-load kernel.  Workgroup order is XCD-aware: consecutive logical ids (= one XCD's L2) are the column tiles
// of one row panel and, for split-K, the tiles of one K-slice.  Split-K (weight gradients: K = #tokens) writes fp32 slabs
// that a second kernel sums in a fixed order (bitwise reproducible) before applying the epilogue.
#include <cstdlib>
#include "gemm_common.h"

using namespace grappa_gemm;

int grappa_launch_gemm_bf16x(hipStream_t st, GemmParams& p, int precision, bool vec_kcontig);   // gemm_bf16x.hip
int grappa_launch_gemm_planes(hipStream_t st, GemmParams& p, int precision);                      // gemm_planes.hip
int grappa_launch_gemm_pairs(hipStream_t st, GemmParams& p);                                      // gemm_pairs.hip
int grappa_launch_amax_combine(hipStream_t st, int M, int nseg, const unsigned* part, unsigned* out);   // amax.hip
int grappa_launch_gemm_bf16x_group4(hipStream_t st, const GemmGroup4& g, int precision, bool b_kcontig);   // gemm_bf16x.hip
int grappa_launch_gemm_pairs_group4(hipStream_t st, const GemmGroup4& g);                                  // gemm_pairs.hip
int grappa_launch_amax_combine4(hipStream_t st, int count, const int* M, const int* nseg, const unsigned* const* part, unsigned* const* out);   // amax.hip
int grappa_launch_gemm_bf16x_grouped(hipStream_t st, const GemmParams* d_ps, const int* d_wg_begin, int nprob, int total_wgs, int precision, bool vec, int psrc);   // gemm_bf16x.hip

namespace {

constexpr int BK = GEMM_BK;
constexpr int NTHREADS = 256;


template <int BROW, bool KCONT>
struct Tile {
    static constexpr int SIZE = KCONT ? BROW * (BK + 1) : BK * BROW;
    static constexpr int NV = BROW * BK / 4 / NTHREADS;   // float4 per thread
};

// 16-byte loads, branch free.  R = number of valid rows, Kend = end of this workgroup's K range.
template <int BROW, bool KCONT>
__device__ inline void load_tile_vec(const float* __restrict__ src, int ld, int row0, int k0, int R, int Kend,
                                     float4 (&v)[Tile<BROW, KCONT>::NV]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < Tile<BROW, KCONT>::NV; ++i) {
        const int f = tid + i * NTHREADS;
        if (KCONT) {
            const int row = min(row0 + (f >> 3), R - 1);
            const int gk = k0 + ((f & 7) << 2);
            const bool ok = gk < Kend;                                // K % 4 == 0 on this path: a float4 is all in or all out
            const float4 x = *reinterpret_cast<const float4*>(src + (size_t)row * ld + (ok ? gk : 0));
            v[i] = ok ? x : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            constexpr int QR = BROW / 4;
            const int gk = k0 + f / QR;
            const bool ok = gk < Kend;
            const int row = min(row0 + ((f % QR) << 2), ((R + 3) & ~3) - 4);   // stays inside the (16-byte padded) row
            const float4 x = *reinterpret_cast<const float4*>(src + (size_t)(ok ? gk : 0) * ld + row);
            v[i] = ok ? x : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

// scalar loads from clamped addresses + select (shapes whose leading dimension / K forbid 16-byte loads)
template <int BROW, bool KCONT>
__device__ inline void load_tile_scalar(const float* __restrict__ src, int ld, int row0, int k0, int R, int Kend,
                                        float4 (&v)[Tile<BROW, KCONT>::NV]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < Tile<BROW, KCONT>::NV; ++i) {
        const int f = tid + i * NTHREADS;
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int gr, gk;
            if (KCONT) {
                gr = row0 + (f >> 3);
                gk = k0 + ((f & 7) << 2) + j;
            } else {
                constexpr int QR = BROW / 4;
                gk = k0 + f / QR;
                gr = row0 + ((f % QR) << 2) + j;
            }
            const bool ok = gr < R && gk < Kend;
            const int cr = ok ? gr : 0, ck = ok ? gk : 0;           // element (0,0) always exists
            const float x = KCONT ? src[(size_t)cr * ld + ck] : src[(size_t)ck * ld + cr];
            e[j] = ok ? x : 0.0f;
        }
        v[i] = make_float4(e[0], e[1], e[2], e[3]);
    }
}

template <int BROW, bool KCONT>
__device__ inline void store_tile(float* __restrict__ S, const float4 (&v)[Tile<BROW, KCONT>::NV]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < Tile<BROW, KCONT>::NV; ++i) {
        const int f = tid + i * NTHREADS;
        if (KCONT) {
            const int row = f >> 3, kq = (f & 7) << 2;
            float* q = S + row * (BK + 1) + kq;
            q[0] = v[i].x; q[1] = v[i].y; q[2] = v[i].z; q[3] = v[i].w;
        } else {
            constexpr int QR = BROW / 4;
            const int k = f / QR, rq = (f % QR) << 2;
            *reinterpret_cast<float4*>(S + k * BROW + rq) = v[i];
        }
    }
}

template <int BROW, bool KCONT>
__device__ inline float read_operand(const float* __restrict__ S, int row, int k) {
    return KCONT ? S[row * (BK + 1) + k] : S[k * BROW + row];
}

template <int BM, int BN, int TM, int TN, bool AK, bool BKC, int K0, int K1>
__device__ inline void compute_part(const float* __restrict__ a_s, const float* __restrict__ b_s, f32x16 (&acc)[TM][TN], int wm0, int wn0,
                                    int lr, int lh) {
#pragma unroll
    for (int kk = K0; kk < K1; kk += 2) {
        float a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = read_operand<BM, AK>(a_s, wm0 + i * 32 + lr, kk + lh);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = read_operand<BN, BKC>(b_s, wn0 + j * 32 + lr, kk + lh);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
}

template <int BM, int BN, bool AK, bool BKC, bool VEC>
__device__ inline void load_pair(const grappa_gemm_desc& d, int m0, int n0, int k0, int kend, float4 (&ra)[Tile<BM, AK>::NV],
                                 float4 (&rb)[Tile<BN, BKC>::NV]) {
    if (VEC) {
        load_tile_vec<BM, AK>(d.A, d.lda, m0, k0, d.M, kend, ra);
        load_tile_vec<BN, BKC>(d.B, d.ldb, n0, k0, d.N, kend, rb);
    } else {
        load_tile_scalar<BM, AK>(d.A, d.lda, m0, k0, d.M, kend, ra);
        load_tile_scalar<BN, BKC>(d.B, d.ldb, n0, k0, d.N, kend, rb);
    }
}

// one K-step of the pipeline; LA/LB: register set receiving tile kt+2, SA/SB: register set holding tile kt+1
template <int NV>
__device__ inline void colsum_accumulate(float4 (&cs)[NV], const float4 (&v)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        cs[i].x += v[i].x; cs[i].y += v[i].y; cs[i].z += v[i].z; cs[i].w += v[i].w;
    }
}

template <int BM, int BN, int TM, int TN, bool AK, bool BKC, bool VEC, bool DO_LOAD, bool DO_STORE>
__device__ inline void pipeline_step(const grappa_gemm_desc& d, float* __restrict__ smem, f32x16 (&acc)[TM][TN], int m0, int n0, int kend,
                                     int k_load, int cur, int wm0, int wn0, int lr, int lh, float4 (&la)[Tile<BM, AK>::NV],
                                     float4 (&lb)[Tile<BN, BKC>::NV], const float4 (&sa)[Tile<BM, AK>::NV],
                                     const float4 (&sb)[Tile<BN, BKC>::NV], float4 (&cs)[Tile<BM, AK>::NV], bool do_cs) {
    constexpr int ASZ = Tile<BM, AK>::SIZE, BSZ = Tile<BN, BKC>::SIZE;
    if (DO_LOAD) load_pair<BM, BN, AK, BKC, VEC>(d, m0, n0, k_load, kend, la, lb);
    __builtin_amdgcn_sched_barrier(0);
    const float* a_s = smem + cur * (ASZ + BSZ);
    compute_part<BM, BN, TM, TN, AK, BKC, 0, BK / 2>(a_s, a_s + ASZ, acc, wm0, wn0, lr, lh);
    __builtin_amdgcn_sched_barrier(0);
    if (DO_STORE) {
        float* nxt = smem + (cur ^ 1) * (ASZ + BSZ);
        store_tile<BM, AK>(nxt, sa);
        store_tile<BN, BKC>(nxt + ASZ, sb);
        if (!AK && do_cs) colsum_accumulate<Tile<BM, AK>::NV>(cs, sa);     // bias gradient rides on the wgrad's A tiles
    }
    __builtin_amdgcn_sched_barrier(0);
    compute_part<BM, BN, TM, TN, AK, BKC, BK / 2, BK>(a_s, a_s + ASZ, acc, wm0, wn0, lr, lh);
    __syncthreads();
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool AK, bool BKC, bool VEC>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_f32_kernel(GemmParams p) {
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int ASZ = Tile<BM, AK>::SIZE;
    extern __shared__ float smem[];   // stage s: A at s*(ASZ+BSZ), B behind it

    const grappa_gemm_desc& d = p.d;
    const TileCoord tc = map_workgroup(p);
    const int split = tc.split, tile_local = tc.tile_local, tile_m = tc.tile_m, tile_n = tc.tile_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = split * p.k_per_split;
    const int kend = min(d.K, kbeg + p.k_per_split);

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
    const int lr = lane & 31, lh = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int nk = (kend - kbeg + BK - 1) / BK;
    // column sums of the row-contiguous A operand (sum over K of A(m,k)): computed by the tile_n == 0 workgroups from the
    // A tiles they stage anyway (the bias gradient db = colsum(dz) of a weight-gradient GEMM)
    const bool do_cs = !AK && d.a_colsum != nullptr && tile_n == 0;
    float4 cs[Tile<BM, AK>::NV];
#pragma unroll
    for (int i = 0; i < Tile<BM, AK>::NV; ++i) cs[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nk > 0) {
        float4 a0[Tile<BM, AK>::NV], b0[Tile<BN, BKC>::NV], a1[Tile<BM, AK>::NV], b1[Tile<BN, BKC>::NV];
        load_pair<BM, BN, AK, BKC, VEC>(d, m0, n0, kbeg, kend, a0, b0);                   // tile 0 -> set 0 -> LDS stage 0
        load_pair<BM, BN, AK, BKC, VEC>(d, m0, n0, kbeg + BK, kend, a1, b1);              // tile 1 -> set 1 (zeros past the end)
        store_tile<BM, AK>(smem, a0);
        store_tile<BN, BKC>(smem + ASZ, b0);
        if (!AK && do_cs) colsum_accumulate<Tile<BM, AK>::NV>(cs, a0);
        __syncthreads();
        int kt = 0;
#define GRAPPA_STEP(DL, DS, LA, LB, SA, SB) \
    pipeline_step<BM, BN, TM, TN, AK, BKC, VEC, DL, DS>(d, smem, acc, m0, n0, kend, kbeg + (kt + 2) * BK, kt & 1, wm0, wn0, lr, lh, LA, LB, SA, SB, cs, do_cs)
        // two K-steps per trip so that the register sets are compile-time names:
        //   even kt: load tile kt+2 into set 0 (tile kt already lives in LDS), store tile kt+1 from set 1; odd kt: the other way round.
        // Loads past the end of the K range are clamped + zero-selected, so every step but the last can prefetch unconditionally.
        for (; kt + 2 < nk; kt += 2) {
            GRAPPA_STEP(true, true, a0, b0, a1, b1);
            ++kt;
            GRAPPA_STEP(true, true, a1, b1, a0, b0);
            --kt;
        }
        if (kt + 1 < nk) {                                    // two steps left (kt even)
            GRAPPA_STEP(false, true, a0, b0, a1, b1);
            ++kt;
            GRAPPA_STEP(false, false, a1, b1, a0, b0);
        } else if (kt < nk) {                                 // one step left
            GRAPPA_STEP(false, false, a0, b0, a1, b1);
        }
#undef GRAPPA_STEP
    }

    if (!AK && do_cs) {
        // every thread holds partial sums for the 4 rows m = 4*(tid % QR) + {0..3}; threads tid, tid+QR, ... share them
        constexpr int QR = BM / 4;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < Tile<BM, AK>::NV; ++i) { t.x += cs[i].x; t.y += cs[i].y; t.z += cs[i].z; t.w += cs[i].w; }
        float4* red = reinterpret_cast<float4*>(smem);          // the pipeline's last barrier has retired all LDS reads
        red[threadIdx.x] = t;
        __syncthreads();
        if (threadIdx.x < QR) {
            float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int g = 0; g < NTHREADS / QR; ++g) {
                const float4 v = red[threadIdx.x + g * QR];
                s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
            }
            const float vals[4] = {s4.x, s4.y, s4.z, s4.w};
            for (int j = 0; j < 4; ++j) {
                const int m = m0 + 4 * threadIdx.x + j;
                if (m < d.M) {
                    if (p.nsplit > 1) p.cs_slab[(size_t)split * d.M + m] = vals[j];
                    else d.a_colsum[m] += vals[j];
                }
            }
        }
    }

    tile_epilogue<BM, BN, TM, TN>(p, acc, m0, n0, wm0, wn0, lr, lh, split, tile_local);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool AK, bool BKC, bool VEC>
int launch_cfg(hipStream_t st, GemmParams& p) {
    constexpr size_t smem = 2 * (size_t)(Tile<BM, AK>::SIZE + Tile<BN, BKC>::SIZE) * sizeof(float);
    auto kern = gemm_f32_kernel<BM, BN, WAVES_M, WAVES_N, AK, BKC, VEC>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return GRAPPA_ERR_LAUNCH;
        attr_set = true;
    }
    dim3 grid(p.ntiles_launch * p.nsplit);
    GRAPPA_LAUNCH(kern, grid, dim3(NTHREADS), smem, st, p);
    return grappa_launch_status();
}

struct Plan {
    int cfg;
    int nsplit, k_per_split;            // main launch (all tiles, or tiles [0, main_tiles) when a tail exists)
    int main_tiles;                     // == total tiles when there is no tail
    int tail_nsplit, tail_k_per_split;  // tail launch: tiles [main_tiles, tiles) with their K range split (0 = no tail)
};

// cfg 0: 128x128, 1: 64x64, 2: 128x32, 3: 32x128, 4: 128x64
// cfg 5, 6: the 128x128 and 256x128 tiles of the bf16-split kernel (gemm_bf16x_impl.h), one 512-thread workgroup per CU
constexpr int NCFG = 9;                 // cfg 7: the 256 x 256 tile of the pair-format kernel (GRAPPA_PAIRS_TILE=256); cfg 8: its 128 x 128 tile (both operands pairs)
constexpr int CFG_BM[NCFG] = {128, 64, 128, 32, 128, 128, 256, 256, 128};
constexpr int CFG_BN[NCFG] = {128, 64, 32, 128, 64, 128, 128, 256, 128};
constexpr int CFG_CONC[NCFG] = {2, 4, 4, 4, 3, 1, 1, 1, 2};   // co-resident workgroups per CU (LDS 67.6 / 33.8 / 42 / 42 / 50.7 / 72 / 108 KB, VGPR budget)

// Per-call plan options (grappa_gemm_desc, ABI 10; nothing here is process-wide state: the environment only provides the defaults, read once).
//   cfg / nsplit: force the tile configuration / split-K factor (tuning, tests); tail: -1 the model's choice, 0 never, 1 forced where possible;
//   tails_on: tail launches allowed at all.  A caller that keeps several streams busy turns them off: a product's partial last round then
//   overlaps with another stream's kernels, and the tail's two extra launches and slab round trip cost more than they return (C2 step, writer
//   heads on four streams: 36.4 -> 36.0 ms; on one stream 37.4 -> 37.5; profiles/r3_plan_tails_ab.txt)
struct PlanOpts {
    int cfg = -1, nsplit = 0, tail = -1;
    bool tails_on = true;
};
constexpr bool TAILS_DEFAULT = true;      // grappa_gemm_desc.plan_tail == 0 (callers that care say 1 or 2)
PlanOpts plan_opts_of(const grappa_gemm_desc& d) {
    PlanOpts o;
    o.cfg = d.plan_cfg > 0 ? d.plan_cfg - 1 : -1;
    o.nsplit = d.plan_nsplit > 0 ? d.plan_nsplit : 0;
    o.tails_on = d.plan_tail == 0 ? TAILS_DEFAULT : d.plan_tail != 2;
    o.tail = d.plan_tail == 3 ? 1 : -1;
    return o;
}
// split-K summed by a launch of gemm_splitk_reduce_kernel behind the product (default) or inside the product's own launch by the last
// workgroup of each tile (grappa_gemm_desc.splitk_reduce = 2).  Same bits; the second is the slower one on this
// chip (profiles/r2_splitk_in_kernel_rejected.txt): one workgroup sums nsplit x 128 KB behind an L2-invalidating acquire while the rest of
// the chip idles at the end of the launch, the reduction kernel spreads the same reads over 256 CUs
bool splitk_in_kernel(const grappa_gemm_desc& d) {
    return d.splitk_reduce == 2;
}
bool use_bf16x(int M, int N, int precision) { return precision != GRAPPA_GEMM_F32_MFMA && M > 32 && N > 32; }

// Tile choice by a small cost model in CU-cycles.  A workgroup of tile c over k_per_split columns of K costs
// BM*BN*(k + K0_c) / RATE_c; a CU runs CONC_c of them at once (each then CONC_c times slower), the chip drains the grid in
// ceil(workgroups / (256*CONC_c)) such rounds.  Split-K adds the slab round trip and two launches.  With an unsplit K the
// tiles beyond the last full 256 can run as a second, split-K "tail" launch (rem * ts workgroups, each 1/ts long) instead of
// costing a whole extra round.  This avoids the "one extra workgroup = one extra round" cliffs of a fixed tile.
// CUs a product plans for: all 256, also with the writer heads on four streams (planning for a share of the chip measured 36.1 -> 37.1 ms per
// C2 step at 128, 37.2 at 64: profiles/r3_plan_cus_rejected.txt)
constexpr long plan_cus() { return 256; }

struct CostModel {
    // MACs per cycle per CU sustained in the main loop, and the per-tile prologue + epilogue expressed in columns of K
    const double rate[NCFG] = {128.0, 90.0, 70.0, 70.0, 119.0, 205.0, 307.0, 450.0, 230.0};
    const double k0[NCFG] = {96.0, 96.0, 96.0, 96.0, 96.0, 128.0, 160.0, 200.0, 160.0};
    double grid(int c, long wgs, int kps) const {
        const int conc = CFG_CONC[c];
        const long ncu = plan_cus();
        double t = (double)CFG_BM[c] * CFG_BN[c] * (kps + k0[c]) / rate[c];
        // a 128 x 128 workgroup of the fp16-split kernel on an otherwise idle chip: 5.0 us + 0.334 us per 16 columns of K measured
        // (profiles/r4_small_gemm_probe.txt) = 50 (k + 240) cycles; the loaded chip's 80 (k + 128) as the grid fills it
        if (c == 5 && wgs < ncu) {
            const double lone = 50.0 * (kps + 240.0);
            t = lone + (t - lone) * ((double)wgs / (double)ncu);
        }
        // the pair kernel's 128 x 128 tile alone on a CU: its 12 MFMAs per wavefront and slab (0.16 us) + DMA turn-round, ~0.2 us per slab
        if (c == 8 && wgs < ncu) {
            const double lone = 30.0 * (kps + 400.0);
            t = lone + (t - lone) * ((double)wgs / (double)ncu);
        }
        if (wgs <= ncu * conc) {
            const double per_cu = (double)((wgs + ncu - 1) / ncu);
            return per_cu * t * (per_cu < 2 && conc > 1 ? 1.3 : 1.0);   // a lone workgroup of a multi-resident tile cannot hide its barrier bubbles
        }
        return (double)((wgs + ncu * conc - 1) / (ncu * conc)) * conc * t;
    }
    static double splitk(int nsplit, double elems) { return 14000.0 + nsplit * elems / 200.0; }   // two launches + slab write / reduce (the reduction launch alone: 5.4 us)

    // The pinned-pipeline pair kernels (gemm_pairs_il.hip), round 5: a launch is a staircase of rounds of one workgroup per CU (the 128 x 128 tile's two
    // co-resident workgroups share the CU's MFMA time: rounds of half the length), fitted to profiles/r5_pairs_tile_sweep.txt (15 row counts x 6 layer
    // shapes x 3 tiles; us at the chip's ~2.1 GHz -> the model's cycles).  256 x 256: a round that fills the chip runs 2x longer than a lone workgroup
    // (the operand intake is shared), so its last, partly filled round is priced by its fill.
    double pairs_grid(int c, long wgs, int kps) const {
        const double ncu = (double)plan_cus();
        const long rounds = (long)((wgs + ncu - 1) / ncu);
        double us;
        if (c == 6) {
            us = rounds * 0.034 * (kps + 132.0) * ((rounds >= 3 && kps > 1024) ? 1.17 : 1.0);
        } else if (c == 8) {
            us = rounds * 0.019 * (kps + 150.0) * ((rounds >= 5 && kps > 1024) ? 1.2 : 1.0);
        } else {
            const double lone = 14.0 + 0.019 * kps, full = 16.0 + 0.069 * kps;
            const double fill = (double)(wgs - (rounds - 1) * (long)ncu) / ncu;
            us = (rounds - 1) * full + lone + fill * (full - lone);
        }
        return us * 2100.0;
    }
};

constexpr bool pairs_tile_choice() { return true; }      // the round-5 planner: a tile per product from the fitted staircase model
constexpr int pairs_cfg() { return 6; }                   // the 256 x 128 tile where the model has no say (fp32 A + weight pairs)

Plan make_plan(int M, int N, int K, const PlanOpts& opt, bool vec = true, bool bf16x = false, bool planes = false, bool pairs = false, bool pairs_small = false,
               int planes_tile = 0) {      // 256 / 128: the bf16 pinned-pipeline kernel's other tiles (GRAPPA_BF16_TILE)
    Plan best;
    best.cfg = 1;
    best.nsplit = 1;
    best.k_per_split = (K + BK - 1) / BK * BK;
    best.main_tiles = 0;
    best.tail_nsplit = 0;
    best.tail_k_per_split = 0;
    double best_cost = 1e300;
    const CostModel cm;
    // the shortest K range of a split: one slab of BK (rounds 1-3: eight): the model may cut K = 512 sixteen ways instead of
    // two (small products -- one molecule, a batch of 32 -- are a few workgroups that each take in their whole K range at a CU's ~30 GB/s)
    constexpr int min_ksteps = 1;
    // (the bf16 plane kernels keep the limit of eight: their configuration's parity sits at its 2e-2 tolerance and moves with the K cuts)
    const int mk = (planes && !pairs) ? 8 : min_ksteps;
    int max_split = K >= 2 * mk * BK ? K / (mk * BK) : 1;
    if (max_split > 64) max_split = 64;
    int max_tail_split = K >= 8 * BK ? K / (4 * BK) : 1;      // the tail may be cut finer than a full split-K GEMM
    if (max_tail_split > 64) max_tail_split = 64;
    // a forced number of K ranges (plan_nsplit, tuning) that this K cannot be cut into means the finest cut that exists, not "no plan"
    int want_split = 0;
    if (opt.nsplit > 0)
        for (int ns = 1; ns <= max_split; ns = ns < 4 ? ns + 1 : ns + (ns + 3) / 4)
            if (ns <= opt.nsplit) want_split = ns;
    for (int c = 0; c < NCFG; ++c) {
        const bool forced_pairs = pairs && opt.cfg >= 6 && opt.cfg <= 8;       // tuning: the pair kernels' tile (6: 256 x 128, 7: 256 x 256, 8: 128 x 128) can be forced too
        if (opt.cfg >= 0 && c != opt.cfg && (!planes || forced_pairs)) continue;
        if (bf16x != (c >= 5)) continue;
        // the plane-format kernels have one tile shape each; the pair kernel (both operands pairs) also a 128 x 128 one
        // (pairs_small: both operands pairs and no forced tile -- round 5: all three tiles compete under the fitted staircase model)
        if (planes && !forced_pairs && c != (pairs ? pairs_cfg() : (planes_tile == 256 ? 7 : (planes_tile == 128 ? 8 : 6))) && !(pairs_small && (c == 8 || (c == 7 && pairs_tile_choice())))) continue;
        if (!planes && (c == 7 || c == 8)) continue;
        if (!vec && (c == 0 || c == 4)) continue;        // the scalar-load kernel is only built for the small tiles
        if (c == 2 && N > 32) continue;
        if (c == 3 && M > 32) continue;
        if ((c == 0 || c == 4 || c == 1) && (N <= 32 || M <= 32)) continue;
        const long tiles = (long)((M + CFG_BM[c] - 1) / CFG_BM[c]) * ((N + CFG_BN[c] - 1) / CFG_BN[c]);
        const double te = (double)CFG_BM[c] * CFG_BN[c];
        for (int ns = 1; ns <= max_split; ns = ns < 4 ? ns + 1 : ns + (ns + 3) / 4) {
            if (want_split > 0 && ns != want_split) continue;
            int kps = (K + ns - 1) / ns;
            // (the pinned-pipeline kernels walk pairs of slabs -- 2 x 16 columns in the pair format, 2 x 32 in the bf16 one: split ranges in whole pairs)
            const bool bf16_il_tile = (planes_tile == 256 && c == 7) || (planes_tile == 128 && c == 8);      // no plane-kernel fallback for these tiles: plan what the pipeline takes
            const int kround = bf16_il_tile ? 64 : (pairs ? 32 : BK);
            kps = (kps + kround - 1) / kround * kround;
            if (bf16_il_tile && (kps < 128 || K - ((K + kps - 1) / kps - 1) * kps < 128)) continue;      // every K range holds the pipeline's four slabs of 32
            const int nsplit = (K + kps - 1) / kps;
            const bool il_model = pairs && pairs_tile_choice() && c >= 6;
            double cost = il_model ? cm.pairs_grid(c, tiles * nsplit, kps) : cm.grid(c, tiles * nsplit, kps);
            if (nsplit > 1) cost += CostModel::splitk(nsplit, (double)M * N);
            Plan cand = best;
            cand.cfg = c;
            cand.nsplit = nsplit;
            cand.k_per_split = kps;
            cand.main_tiles = (int)tiles;
            cand.tail_nsplit = 0;
            cand.tail_k_per_split = 0;
            const long ncu = plan_cus();
            const long rem = tiles % ncu;
            // (pair kernels, measured: a tail launch pays only for a deep K and a handful of leftover tiles -- at K = 512 it lost 4-15% on every row count)
            const bool tail_ok = !il_model || opt.tail == 1 || (K >= 1024 && rem <= ncu / 8);
            if (nsplit == 1 && tiles > ncu && rem > 0 && rem <= ncu * 5 / 8 && opt.tail != 0 && opt.tails_on && tail_ok) {
                int ts = (int)(ncu / rem);
                if (ts > max_tail_split) ts = max_tail_split;
                int tkps = (K + ts - 1) / ts;
                tkps = (tkps + kround - 1) / kround * kround;
                const int tns = (K + tkps - 1) / tkps;
                if (tns >= 2) {
                    const double with_tail = il_model ? cm.pairs_grid(c, tiles - rem, kps) + cm.pairs_grid(c, rem * tns, tkps) + CostModel::splitk(tns, rem * te)
                                                      : cm.grid(c, tiles - rem, kps) + cm.grid(c, rem * tns, tkps) + CostModel::splitk(tns, rem * te);
                    if (with_tail < cost || opt.tail == 1) {
                        cost = with_tail;
                        cand.main_tiles = (int)(tiles - rem);
                        cand.tail_nsplit = tns;
                        cand.tail_k_per_split = tkps;
                    }
                }
            }
            if (cost < best_cost) {
                best_cost = cost;
                best = cand;
            }
        }
    }
    return best;
}

size_t plan_workspace_floats(const Plan& pl, int M, int N) {
    const long tiles = (long)((M + CFG_BM[pl.cfg] - 1) / CFG_BM[pl.cfg]) * ((N + CFG_BN[pl.cfg] - 1) / CFG_BN[pl.cfg]);
    const size_t te = (size_t)CFG_BM[pl.cfg] * CFG_BN[pl.cfg];
    size_t need = 0;
    // slabs, column-sum partials, one ticket (int) per tile of the launch
    if (pl.nsplit > 1) need = (size_t)pl.nsplit * tiles * te + (size_t)pl.nsplit * M + tiles;
    if (pl.tail_nsplit > 1) {
        const size_t t = (size_t)pl.tail_nsplit * (tiles - pl.main_tiles) * te + (size_t)pl.tail_nsplit * M + (tiles - pl.main_tiles);
        if (t > need) need = t;
    }
    return need;
}

template <bool AK, bool BKC, bool VEC>
int dispatch_cfg(hipStream_t st, GemmParams& p, int cfg) {
    switch (cfg) {
        case 0: return launch_cfg<128, 128, 2, 2, AK, BKC, VEC>(st, p);
        case 1: return launch_cfg<64, 64, 2, 2, AK, BKC, VEC>(st, p);
        case 2: return launch_cfg<128, 32, 4, 1, AK, BKC, VEC>(st, p);
        case 3: return launch_cfg<32, 128, 1, 4, AK, BKC, VEC>(st, p);
        default: return launch_cfg<128, 64, 2, 2, AK, BKC, VEC>(st, p);
    }
}

template <bool AK, bool BKC>
int dispatch(hipStream_t st, GemmParams& p, int cfg, bool vec) {
    if (vec) return dispatch_cfg<AK, BKC, true>(st, p, cfg);
    // scalar-load kernel: odd shapes only (K = 85, ld = 511, ...); the big tiles are not instantiated for it
    return dispatch_cfg<AK, BKC, false>(st, p, cfg);
}


// the straight-line row epilogues (gemm_common.h epilogue_band_fast): one output tensor, whole float4s, no pre-activation addend
int choose_epi_class(const grappa_gemm_desc& d, bool vec_io, bool bf16x) {
    if (!(bf16x && (d.N & 3) == 0 && !d.C2 && !d.C1p && !d.pre && !d.accumulate)) return 0;
    const bool f32_only = vec_io && d.C && !d.Cp && !d.resp && !d.auxp;
    const bool bf16_only = d.Cp && !d.C && !d.res && !d.aux && d.cp_nplanes == 1 && (!d.resp || d.resp_nplanes == 1) &&
                           (!d.auxp || d.auxp_nplanes == 1);
    if (!(f32_only || bf16_only)) return 0;
    const bool has_aux = d.aux || d.auxp, has_res = d.res || d.resp;
    int cls;
    if (has_aux) cls = (!d.bias && d.act == GRAPPA_ACT_NONE && d.drop_p == 0.0f) ? 4 : 0;
    else if (d.act == GRAPPA_ACT_ELU) cls = (d.drop_p == 0.0f && !has_res) ? 2 : 0;
    else cls = (d.drop_p > 0.0f || has_res) ? 3 : 1;
    if (cls == 3 && d.res_ln_mean) cls = f32_only ? 5 : 0;             // the residual recomputed from the rows before their LayerNorm
    return cls == 0 ? 0 : cls + (bf16_only ? 8 : 0);
}
}  // namespace

// per-segment row maxima of OUT (d.out_amax): M x ceil(N / 32) words at most (32 columns = the narrowest wavefront share of a tile row),
// placed behind the split-K slabs
static size_t amax_part_bytes(int M, int N) { return (size_t)M * ((N + 31) / 32) * sizeof(unsigned); }

static size_t workspace_floats_all(int M, int N, int K, const PlanOpts& o) {
    size_t a = plan_workspace_floats(make_plan(M, N, K, o, true), M, N);
    const size_t b = plan_workspace_floats(make_plan(M, N, K, o, false), M, N);
    if (b > a) a = b;
    if (use_bf16x(M, N, GRAPPA_GEMM_F32_BF16X9)) {
        const size_t c = plan_workspace_floats(make_plan(M, N, K, o, true, true), M, N);
        if (c > a) a = c;
        const size_t e = plan_workspace_floats(make_plan(M, N, K, o, true, true, true), M, N);
        if (e > a) a = e;
        const size_t f = plan_workspace_floats(make_plan(M, N, K, o, true, true, true, true), M, N);
        if (f > a) a = f;
        const size_t g = plan_workspace_floats(make_plan(M, N, K, o, true, true, true, true, true), M, N);
        if (g > a) a = g;
        const size_t h = plan_workspace_floats(make_plan(M, N, K, o, true, true, true, false, false, 256), M, N);
        if (h > a) a = h;
        const size_t h2 = plan_workspace_floats(make_plan(M, N, K, o, true, true, true, false, false, 128), M, N);
        if (h2 > a) a = h2;
    }
    return a;
}

// one ticket (int) per tile beside the slabs when the reduction runs inside the product's launch: plan_workspace_floats counts them
extern "C" size_t grappa_gemm_f32_workspace_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    PlanOpts on, off;
    on.tails_on = true;
    off.tails_on = false;
    const size_t a = workspace_floats_all(M, N, K, on), b = workspace_floats_all(M, N, K, off);
    return (a > b ? a : b) * sizeof(float) + amax_part_bytes(M, N);
}

extern "C" size_t grappa_gemm_f32_workspace_bytes_desc(const grappa_gemm_desc* d) {
    if (!d || d->M <= 0 || d->N <= 0 || d->K <= 0) return 0;
    const size_t a = workspace_floats_all(d->M, d->N, d->K, plan_opts_of(*d)) * sizeof(float) + amax_part_bytes(d->M, d->N);
    const size_t b = grappa_gemm_f32_workspace_bytes(d->M, d->N, d->K);
    return a > b ? a : b;
}

static int plan_report(int M, int N, int K, int precision, const PlanOpts& o, int* tile_m, int* tile_n, int* nsplit, int* tail_tiles, int* tail_nsplit) {
    if (M <= 0 || N <= 0 || K <= 0 || !tile_m || !tile_n || !nsplit || !tail_tiles || !tail_nsplit) return GRAPPA_ERR_ARG;
    if (precision < GRAPPA_GEMM_F32_MFMA || precision > GRAPPA_GEMM_F32_F16X3) return GRAPPA_ERR_ARG;
    Plan pl = make_plan(M, N, K, o, true, use_bf16x(M, N, precision));
    if (pl.main_tiles <= 0) return GRAPPA_ERR_ARG;          // a forced tile this product has no kernel for
    *tile_m = CFG_BM[pl.cfg];
    *tile_n = CFG_BN[pl.cfg];
    *nsplit = pl.nsplit;
    const long tiles = (long)((M + CFG_BM[pl.cfg] - 1) / CFG_BM[pl.cfg]) * ((N + CFG_BN[pl.cfg] - 1) / CFG_BN[pl.cfg]);
    *tail_tiles = pl.tail_nsplit > 1 ? (int)(tiles - pl.main_tiles) : 0;
    *tail_nsplit = pl.tail_nsplit;
    return GRAPPA_OK;
}

extern "C" int grappa_gemm_f32_plan(int M, int N, int K, int precision, int* tile_m, int* tile_n, int* nsplit, int* tail_tiles, int* tail_nsplit) {
    PlanOpts o;
    o.tails_on = TAILS_DEFAULT;
    return plan_report(M, N, K, precision, o, tile_m, tile_n, nsplit, tail_tiles, tail_nsplit);
}

// the same for a descriptor's shape, precision and plan options (fp32 operands)
extern "C" int grappa_gemm_f32_plan_desc(const grappa_gemm_desc* d, int* tile_m, int* tile_n, int* nsplit, int* tail_tiles, int* tail_nsplit) {
    if (!d) return GRAPPA_ERR_ARG;
    return plan_report(d->M, d->N, d->K, d->precision, plan_opts_of(*d), tile_m, tile_n, nsplit, tail_tiles, tail_nsplit);
}

// ------------------------------------------------------------------------------------------------ grouped weight gradients
namespace {
constexpr int GROUP_MAX = GRAPPA_GEMM_GROUP_MAX;
struct GroupUpload {              // kernel-argument carrier: descriptors reach device memory without a host-side copy to wait for
    GemmParams p[8];
};
struct GroupIndex {
    int wg_begin[GROUP_MAX + 1], blk_begin[GROUP_MAX + 1];
};
__global__ void group_upload_kernel(GroupUpload u, int n, GemmParams* dst) {
    const int words = (int)(sizeof(GemmParams) / 4);
    for (int i = threadIdx.x; i < n * words; i += blockDim.x)
        reinterpret_cast<unsigned*>(dst)[i] = reinterpret_cast<const unsigned*>(u.p)[i];
}
__global__ void group_index_kernel(GroupIndex ix, int n, int* dst_wg, int* dst_blk, int* tickets, int ntickets) {
    for (int i = threadIdx.x; i <= n; i += blockDim.x) {
        dst_wg[i] = ix.wg_begin[i];
        dst_blk[i] = ix.blk_begin[i];
    }
    for (int i = threadIdx.x; i < ntickets; i += blockDim.x) tickets[i] = 0;      // split-K tickets of the launch that follows
}

struct GroupPlan {
    int kps;
    int nsplit[GROUP_MAX], tiles[GROUP_MAX];
    int total_wgs;
    size_t slab_floats[GROUP_MAX], cs_floats[GROUP_MAX];
    size_t header_bytes, ticket_offset, total_bytes;      // tickets: one int per tile of every split problem, behind all slabs
    int total_tickets;
};

// pair-format operand of the wgrad layout (ABI 8): [K tokens][rows] pairs, every token row under its own scale (rowmax[k]); F32_F16X3,
// one whole-tensor scale per operand (amax_bcast), 16-byte aligned rows, rows % 32 == 0 (the pair rows carry no partial granule)
bool pair_operand_ok(const void* q, int ld, int rows, const uint32_t* rowmax, int precision, int bcast_bit, int amax_bcast) {
    return precision == GRAPPA_GEMM_F32_F16X3 && rowmax != nullptr && (amax_bcast & bcast_bit) != 0 && (reinterpret_cast<uintptr_t>(q) & 15) == 0 &&
           (ld & 7) == 0 && (rows & 31) == 0 && ld >= 2 * rows;
}

bool group_desc_ok(const grappa_gemm_desc& d, int precision) {
    if (d.a_kcontig || d.b_kcontig) return false;                                      // the wgrad layout
    if (d.a_planes && !pair_operand_ok(d.A, d.lda, d.M, d.a_rowmax, d.precision, 1, d.amax_bcast)) return false;
    if (d.b_planes && !pair_operand_ok(d.B, d.ldb, d.N, d.b_rowmax, d.precision, 2, d.amax_bcast)) return false;
    if (!d.A || !d.B || !d.C || d.Cp || d.C1p || d.resp || d.auxp || d.out_amax) return false;
    if (d.M <= 32 || d.N <= 32 || d.K <= 0) return false;
    if (d.precision != precision || d.precision == GRAPPA_GEMM_F32_MFMA) return false;
    if (d.precision == GRAPPA_GEMM_F32_F16X3 && (!d.a_amax || !d.b_amax)) return false;
    if (d.drop_p < 0.0f || d.drop_p >= 1.0f) return false;
    return true;
}

// one K chunk for every problem of the group: the chunk that minimises rounds x tile time + the slab round trips (same cost units
// as CostModel: CU-cycles)
GroupPlan plan_group(const grappa_gemm_desc* descs, int n) {
    GroupPlan g;
    double work = 0;
    int kmax = 0;
    for (int i = 0; i < n; ++i) {
        g.tiles[i] = ((descs[i].M + 255) / 256) * ((descs[i].N + 127) / 128);
        work += (double)g.tiles[i] * descs[i].K;
        kmax = descs[i].K > kmax ? descs[i].K : kmax;
    }
    double best = 1e300;
    g.kps = (kmax + 31) / 32 * 32;
    for (int R = 1; R <= 12; ++R) {
        int kps = (int)(work / (256.0 * R));
        kps = (kps + 31) / 32 * 32;
        if (kps < 1024) kps = 1024;
        if (kps > (kmax + 31) / 32 * 32) kps = (kmax + 31) / 32 * 32;
        long wgs = 0;
        double slab = 0;
        for (int i = 0; i < n; ++i) {
            const int ns = (descs[i].K + kps - 1) / kps;
            wgs += (long)g.tiles[i] * ns;
            if (ns > 1) slab += (double)ns * descs[i].M * descs[i].N / 200.0;
        }
        const double cost = (double)((wgs + 255) / 256) * 32768.0 * (kps + 160.0) / 307.0 + slab;
        if (cost < best) {
            best = cost;
            g.kps = kps;
        }
    }
    g.total_wgs = 0;
    size_t off = 0;
    for (int i = 0; i < n; ++i) {
        g.nsplit[i] = (descs[i].K + g.kps - 1) / g.kps;
        g.total_wgs += g.tiles[i] * g.nsplit[i];
        g.slab_floats[i] = g.nsplit[i] > 1 ? (size_t)g.nsplit[i] * g.tiles[i] * 256 * 128 : 0;
        g.cs_floats[i] = g.nsplit[i] > 1 ? (size_t)g.nsplit[i] * descs[i].M : 0;
        off += g.slab_floats[i] + g.cs_floats[i];
    }
    g.header_bytes = ((size_t)n * sizeof(GemmParams) + 2 * (GROUP_MAX + 1) * sizeof(int) + 255) / 256 * 256;
    g.ticket_offset = g.header_bytes + off * sizeof(float);
    g.total_tickets = 0;
    for (int i = 0; i < n; ++i) g.total_tickets += g.nsplit[i] > 1 ? g.tiles[i] : 0;
    g.total_bytes = g.ticket_offset + (size_t)g.total_tickets * sizeof(int);
    return g;
}
}  // namespace

extern "C" size_t grappa_gemm_f32_grouped_workspace_bytes(const grappa_gemm_desc* descs, int n) {
    if (!descs || n <= 0 || n > GROUP_MAX) return 0;
    return plan_group(descs, n).total_bytes;
}

extern "C" int grappa_gemm_f32_grouped(void* stream, const grappa_gemm_desc* descs, int n, void* ws, size_t ws_bytes) {
    if (!descs || n <= 0 || n > GROUP_MAX) return GRAPPA_ERR_ARG;
    const int precision = descs[0].precision;
    // operand formats: all fp32 (psrc 0), one combination for the whole group (1 .. 3: the kernel specialised for it), or mixed (4: the
    // kernel that reads every product's own flags -- one launch for a backward pass's products whatever their producers wrote)
    int psrc = (descs[0].a_planes ? 1 : 0) | (descs[0].b_planes ? 2 : 0);
    for (int i = 0; i < n; ++i) {
        if (!group_desc_ok(descs[i], precision)) return GRAPPA_ERR_ARG;
        if (((descs[i].a_planes ? 1 : 0) | (descs[i].b_planes ? 2 : 0)) != psrc) psrc = 4;
    }
    const GroupPlan g = plan_group(descs, n);
    if (!ws || ws_bytes < g.total_bytes) return GRAPPA_ERR_WORKSPACE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    GemmParams* d_ps = reinterpret_cast<GemmParams*>(ws);
    int* d_wg = reinterpret_cast<int*>(reinterpret_cast<char*>(ws) + (size_t)n * sizeof(GemmParams));
    int* d_blk = d_wg + (GROUP_MAX + 1);
    float* slab = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + g.header_bytes);
    GroupIndex ix;
    GroupUpload up;
    ix.wg_begin[0] = ix.blk_begin[0] = 0;
    const bool in_kernel = splitk_in_kernel(descs[0]);        // (one launch: the group follows its first product's option)
    int* tickets = reinterpret_cast<int*>(reinterpret_cast<char*>(ws) + g.ticket_offset);
    int* next_ticket = tickets;
    auto al16 = [](const void* q, int ld) { return q == nullptr || ((reinterpret_cast<uintptr_t>(q) & 15) == 0 && (ld & 3) == 0); };
    int total_blocks = 0;
    // 16-byte loads along the rows of both operands of every product: aligned bases, leading dimensions % 4 == 0 covering round_up(rows, 4)
    bool vec = true;
    for (int i = 0; i < n; ++i) {
        const grappa_gemm_desc& d = descs[i];
        vec = vec && (reinterpret_cast<uintptr_t>(d.A) & 15) == 0 && (d.lda & 3) == 0 && ((d.M + 3) & ~3) <= d.lda &&
              (reinterpret_cast<uintptr_t>(d.B) & 15) == 0 && (d.ldb & 3) == 0 && ((d.N + 3) & ~3) <= d.ldb;
    }
    if (psrc != 0 && !vec) return GRAPPA_ERR_ARG;                                      // (fp32 partner of a pair operand must allow 16-byte row loads)
    for (int i0 = 0; i0 < n; i0 += 8) {
        const int cnt = n - i0 < 8 ? n - i0 : 8;
        for (int j = 0; j < cnt; ++j) {
            const int i = i0 + j;
            const grappa_gemm_desc& d = descs[i];
            GemmParams& p = up.p[j];
            p.d = d;
            p.k_per_split = g.kps;
            p.nsplit = g.nsplit[i];
            p.slab = p.nsplit > 1 ? slab : nullptr;
            p.cs_slab = p.nsplit > 1 ? slab + g.slab_floats[i] : nullptr;
            slab += g.slab_floats[i] + g.cs_floats[i];
            p.tickets = nullptr;
            if (p.nsplit > 1 && in_kernel) {
                p.tickets = next_ticket;
                next_ticket += g.tiles[i];
            }
            p.drop_scale = d.drop_p > 0.0f ? 1.0f / (1.0f - d.drop_p) : 1.0f;
            p.drop_salt = d.drop_salt;
            p.bm = 256;
            p.bn = 128;
            p.tiles_m = (d.M + 255) / 256;
            p.tiles_n = (d.N + 127) / 128;
            p.tile_begin = 0;
            p.ntiles_launch = g.tiles[i];
            p.vec_io = al16(d.C, d.ldc) && al16(d.C2, d.ldc2) && al16(d.pre, d.ldpre) && al16(d.res, d.ldres) && al16(d.aux, d.ldaux);
            p.amax_part = nullptr;
            p.amax_seg = 32;
            p.epi_class = 0;
            ix.wg_begin[i + 1] = ix.wg_begin[i] + g.tiles[i] * g.nsplit[i];
            int blocks = 0;
            if (p.nsplit > 1 && !in_kernel) {
                blocks = (int)(((size_t)g.tiles[i] * 256 * 128 / 4 + REDUCE_THREADS - 1) / REDUCE_THREADS);
                if (blocks > 1024) blocks = 1024;
            }
            ix.blk_begin[i + 1] = ix.blk_begin[i] + blocks;
            total_blocks = ix.blk_begin[i + 1];
        }
        GRAPPA_LAUNCH(group_upload_kernel, dim3(1), dim3(256), 0, st, up, cnt, d_ps + i0);
    }
    GRAPPA_LAUNCH(group_index_kernel, dim3(1), dim3(256), 0, st, ix, n, d_wg, d_blk, tickets, in_kernel ? g.total_tickets : 0);
    if (grappa_launch_status() != GRAPPA_OK) return GRAPPA_ERR_LAUNCH;
    int rc = grappa_launch_gemm_bf16x_grouped(st, d_ps, d_wg, n, g.total_wgs, precision, vec, psrc);
    if (rc != GRAPPA_OK) return rc;
    if (total_blocks > 0) {
        GRAPPA_LAUNCH(gemm_splitk_reduce_grouped_kernel, dim3(total_blocks), dim3(REDUCE_THREADS), 0, st, d_ps, d_blk, n);
        rc = grappa_launch_status();
    }
    return rc;
}

extern "C" int grappa_gemm_f32(void* stream, const grappa_gemm_desc* d, void* ws, size_t ws_bytes) {
    if (!d || !d->A || !d->B || (!d->C && !d->Cp && !d->C1p)) return GRAPPA_ERR_ARG;
    if (!d->C && (d->C2 || d->accumulate)) return GRAPPA_ERR_ARG;
    if (d->C1p && (d->C || d->C2) ) return GRAPPA_ERR_ARG;        // C1p replaces the (C, C2) pair: the final value then goes to Cp
    if (d->C1p && !d->Cp) return GRAPPA_ERR_ARG;
    for (int np : {d->cp_nplanes, d->resp_nplanes, d->auxp_nplanes})
        if (np != 0 && np != 1 && np != 3) return GRAPPA_ERR_ARG;
    if (d->a_planes && !d->b_planes) return GRAPPA_ERR_ARG;
    if ((d->a_planes || d->b_planes) && !d->a_kcontig && d->precision == GRAPPA_GEMM_F32_F16X3) return GRAPPA_ERR_ARG;   // ABI 8 pair operands of a weight gradient: grappa_gemm_f32_grouped only
    const bool planes = d->b_planes != 0;
    // pair format (fp16 hi / lo planes + the row maxima that define the power-of-two scale of every row): both operands, K-contiguous
    const bool pairs = planes && d->precision == GRAPPA_GEMM_F32_F16X3;
    if (pairs && (!d->a_kcontig || !d->b_kcontig || !d->a_amax || !d->b_amax || d->amax_bcast || d->a_colsum)) return GRAPPA_ERR_ARG;
    // "weight pairs": fp32 A (whole slabs of 16 in K, 16-byte aligned rows) with the weight in pairs
    if (pairs && !d->a_planes && ((d->K & 15) != 0 || (reinterpret_cast<uintptr_t>(d->A) & 15) != 0 || (d->lda & 3) != 0 || d->lda < d->K)) return GRAPPA_ERR_ARG;
    if (planes) {
        auto ok = [](const void* q, int ld, int cols) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0 && (ld & 7) == 0 && ld >= cols; };
        const int kpad = (d->K + 31) / 32 * 32;
        const int krow = pairs ? 2 * kpad : kpad;          // fp16 elements of a K-contiguous row: the pair format holds hi AND lo
        if (d->M <= 32 || d->N <= 32 || d->a_kcontig != d->b_kcontig) return GRAPPA_ERR_ARG;
        if (d->a_planes) {
            if (d->a_kcontig ? !(ok(d->A, d->lda, krow) && ok(d->B, d->ldb, krow)) : !(ok(d->A, d->lda, d->M) && ok(d->B, d->ldb, d->N))) return GRAPPA_ERR_ARG;
        } else if (pairs) {
            if (!ok(d->B, d->ldb, krow)) return GRAPPA_ERR_ARG;
        } else {
            // fp32 A [M][K] + weight planes B [N][K]: whole slabs of 32 in K (fp32 rows are not zero padded), 16-byte aligned rows
            if (!d->a_kcontig || (d->K & 31) != 0 || (reinterpret_cast<uintptr_t>(d->A) & 15) != 0 || (d->lda & 3) != 0 || d->lda < d->K) return GRAPPA_ERR_ARG;
            if (!ok(d->B, d->ldb, kpad)) return GRAPPA_ERR_ARG;
        }
        const size_t arows = d->a_kcontig ? (size_t)d->M : (size_t)d->K, brows = d->b_kcontig ? (size_t)d->N : (size_t)d->K;
        if (arows * d->lda * (d->a_planes ? 2 : 4) >= (1ull << 32) || brows * d->ldb * 2 >= (1ull << 32)) return GRAPPA_ERR_ARG;
    }
    auto al8 = [](const void* q, int ld) { return q == nullptr || ((reinterpret_cast<uintptr_t>(q) & 7) == 0 && (ld & 3) == 0); };
    if (!al8(d->Cp, d->ldcp) || !al8(d->resp, d->ldresp) || !al8(d->auxp, d->ldauxp) || !al8(d->C1p, d->ldc1p)) return GRAPPA_ERR_ARG;
    if (d->M < 0 || d->N < 0 || d->K < 0) return GRAPPA_ERR_ARG;
    if (d->M == 0 || d->N == 0) return GRAPPA_OK;
    if (d->K == 0) return GRAPPA_ERR_ARG;
    if (d->a_kcontig == 0 && d->b_kcontig == 1) return GRAPPA_ERR_ARG;   // layout never needed by the path
    if (d->drop_p < 0.0f || d->drop_p >= 1.0f) return GRAPPA_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    GemmParams p;
    p.d = *d;
    // 16-byte loads: aligned base, leading dimension % 4 == 0, and for K-contiguous operands K % 4 == 0
    const bool vecA = (reinterpret_cast<uintptr_t>(d->A) & 15) == 0 && (d->lda & 3) == 0 && (!d->a_kcontig || (d->K & 3) == 0);
    const bool vecB = (reinterpret_cast<uintptr_t>(d->B) & 15) == 0 && (d->ldb & 3) == 0 && (!d->b_kcontig || (d->K & 3) == 0);
    // row-contiguous operands read float4 along rows: the padded row (ld) must cover round_up(rows, 4)
    const bool padA = d->a_kcontig || ((d->M + 3) & ~3) <= d->lda;
    const bool padB = d->b_kcontig || ((d->N + 3) & ~3) <= d->ldb;
    const bool vec = vecA && vecB && padA && padB;
    p.drop_scale = d->drop_p > 0.0f ? 1.0f / (1.0f - d->drop_p) : 1.0f;
    p.drop_salt = d->drop_salt;
    auto al16 = [](const void* q, int ld) { return q == nullptr || ((reinterpret_cast<uintptr_t>(q) & 15) == 0 && (ld & 3) == 0); };
    p.vec_io = al16(d->C, d->ldc) && al16(d->C2, d->ldc2) && al16(d->pre, d->ldpre) && al16(d->res, d->ldres) && al16(d->aux, d->ldaux);
    if (d->precision < GRAPPA_GEMM_F32_MFMA || d->precision > GRAPPA_GEMM_F32_F16X3) return GRAPPA_ERR_ARG;
    const bool bf16x = planes || use_bf16x(d->M, d->N, d->precision);
    // fp16 pieces need the row maxima of both operands (fp32 operands only: the plane format is a bf16 split)
    if (d->precision == GRAPPA_GEMM_F32_F16X3 && bf16x && (!d->a_amax || !d->b_amax)) return GRAPPA_ERR_ARG;
    // the native fp32 kernel (precision F32_MFMA, or M / N <= 32) keeps its register-lean fp32-only epilogue walk
    if (!bf16x && (d->Cp || d->C1p || d->resp || d->auxp || !d->C)) return GRAPPA_ERR_ARG;
    if (d->res_ln_mean) {
        // residual = LayerNorm(res): fp32 rows, all four arrays, the split kernels' shared epilogue only (not the native fp32 kernel's walk)
        if (d->aux || d->auxp) return GRAPPA_ERR_ARG;                   // (the ELU' class adds its residual raw: refused, not computed wrongly)
        if (!bf16x || !d->res || !d->res_ln_rstd || !d->res_ln_gamma || !d->res_ln_beta || (d->N & 3) ||
            ((reinterpret_cast<uintptr_t>(d->res_ln_gamma) | reinterpret_cast<uintptr_t>(d->res_ln_beta)) & 15) != 0 || (planes && !pairs))
            return GRAPPA_ERR_ARG;
    }
    constexpr bool small_tile = true;
    const PlanOpts popt = plan_opts_of(*d);
    Plan pl = make_plan(d->M, d->N, d->K, popt, vec || planes, bf16x, planes, pairs, pairs && d->a_planes && small_tile && popt.cfg < 0, 0);
    if (pl.main_tiles <= 0) return GRAPPA_ERR_ARG;                       // a forced tile (plan_cfg) that no kernel of this product's operand formats has: refused, not skipped
    if (pairs && !d->a_planes && pl.cfg != 6) return GRAPPA_ERR_ARG;      // (fp32 A + weight pairs: the 256 x 128 tile only)
    if (d->a_colsum && d->a_kcontig) return GRAPPA_ERR_ARG;          // column sums ride on the row-contiguous (wgrad) A operand only
    if (d->plan_cfg < 0 || d->plan_cfg > NCFG || d->plan_nsplit < 0 || d->plan_tail < 0 || d->plan_tail > 3 || d->splitk_reduce < 0 || d->splitk_reduce > 2) return GRAPPA_ERR_ARG;
    const size_t need = plan_workspace_floats(pl, d->M, d->N) * sizeof(float);
    if (need > 0 && (!ws || ws_bytes < need)) return GRAPPA_ERR_WORKSPACE;
    p.bm = CFG_BM[pl.cfg];
    p.bn = CFG_BN[pl.cfg];
    p.tiles_m = (d->M + p.bm - 1) / p.bm;
    p.tiles_n = (d->N + p.bn - 1) / p.bn;
    const int tiles = p.tiles_m * p.tiles_n;
    const size_t te = (size_t)p.bm * p.bn;

    auto launch = [&](int tile_begin, int ntiles, int nsplit, int kps) -> int {
        p.tile_begin = tile_begin;
        p.ntiles_launch = ntiles;
        p.nsplit = nsplit;
        p.k_per_split = kps;
        p.slab = nullptr;
        p.cs_slab = nullptr;
        p.tickets = nullptr;
        if (nsplit > 1) {
            p.slab = reinterpret_cast<float*>(ws);
            p.cs_slab = p.slab + (size_t)nsplit * ntiles * te;
            if (bf16x && !planes && splitk_in_kernel(*d)) {                // the last workgroup of a tile reduces it (gemm_common.h splitk_finish_tile)
                p.tickets = reinterpret_cast<int*>(p.cs_slab + (size_t)nsplit * d->M);
                if (hipMemsetAsync(p.tickets, 0, (size_t)ntiles * sizeof(int), st) != hipSuccess) return GRAPPA_ERR_LAUNCH;
            }
        }
        int rc;
        if (pairs) rc = grappa_launch_gemm_pairs(st, p);
        else if (planes) rc = grappa_launch_gemm_planes(st, p, d->precision);
        else if (bf16x) rc = grappa_launch_gemm_bf16x(st, p, d->precision, vec);
        else if (d->a_kcontig && d->b_kcontig) rc = dispatch<true, true>(st, p, pl.cfg, vec);
        else if (d->a_kcontig) rc = dispatch<true, false>(st, p, pl.cfg, vec);
        else rc = dispatch<false, false>(st, p, pl.cfg, vec);
        if (rc != GRAPPA_OK) return rc;
        if (nsplit > 1 && !p.tickets) rc = launch_splitk_reduce(st, p);
        return rc;
    };
    int rc = GRAPPA_OK;
    // row maxima of OUT: the row-epilogue kernels and the split-K reduction leave per-segment maxima behind the slabs (plain stores),
    // one small launch combines them; the native fp32 kernel (tiny or non-default products) leaves them to one pass over its output
    const bool amax_fused = (d->out_amax || d->out_amax_parts) && bf16x;
    if (d->out_amax_parts && (!bf16x || d->out_amax)) return GRAPPA_ERR_ARG;          // partials: the split kernels' epilogue only, instead of out_amax
    // a consumer of partials: fp32 A of the split kernels, or (round 5) of the pinned-pipeline weight-pairs kernel (the round-3 weight-pairs loop has no combine)
    // (the SAME predicate as the dispatch, grappa_wpairs_il_takes in gemm_common.h -- the round-3 fall-back loop reads one maximum per
    //  row, so partial maxima are refused wherever a launch of this plan would fall back to it, e.g. a short last split-K range)
    const bool wpairs_il_ok = pairs && !d->a_planes && grappa_wpairs_il_takes(d->K, pl.nsplit, pl.k_per_split) &&
                              (pl.tail_nsplit <= 1 || pl.main_tiles >= tiles || grappa_wpairs_il_takes(d->K, pl.tail_nsplit, pl.tail_k_per_split));
    if (d->a_amax_nseg > 1 && (!d->a_kcontig || (planes && !wpairs_il_ok) || !bf16x || (d->amax_bcast & 1))) return GRAPPA_ERR_ARG;
    if (d->out_amax && !amax_fused && !d->C) return GRAPPA_ERR_ARG;
    p.amax_part = nullptr;
    p.amax_seg = 32;                                        // segments of 32 columns whatever the tile (ABI 8: the consumer may combine them)
    // the straight-line row epilogues (gemm_common.h epilogue_band_fast): one fp32 output, whole float4s, no pre-activation addend
    p.epi_class = choose_epi_class(*d, p.vec_io != 0, bf16x);
    if (amax_fused && d->out_amax_parts) {
        p.amax_part = d->out_amax_parts;         // the caller's array: the consumer combines
    } else if (amax_fused) {
        if (!ws || ws_bytes < need + amax_part_bytes(d->M, d->N)) return GRAPPA_ERR_WORKSPACE;
        p.amax_part = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws) + need);
    }
    if (pl.main_tiles > 0) rc = launch(0, pl.main_tiles, pl.nsplit, pl.k_per_split);
    if (rc == GRAPPA_OK && pl.tail_nsplit > 1 && pl.main_tiles < tiles)
        rc = launch(pl.main_tiles, tiles - pl.main_tiles, pl.tail_nsplit, pl.tail_k_per_split);
    if (rc == GRAPPA_OK && amax_fused && !d->out_amax_parts) rc = grappa_launch_amax_combine(st, d->M, (d->N + p.amax_seg - 1) / p.amax_seg, p.amax_part, d->out_amax);
    if (rc == GRAPPA_OK && d->out_amax && !amax_fused) {
        const float* o = d->C2 ? d->C2 : d->C;
        rc = grappa_amax_f32(stream, d->M, d->N, o, d->C2 ? d->ldc2 : d->ldc, d->out_amax, nullptr, nullptr, 0);
    }
    return rc;
}


// ------------------------------------------------------------------------------------------------ grouped forward / input-gradient products
// Up to four independent products of ONE layout and operand format in one launch: the same product of the four writer heads.  Each
// keeps its own epilogue (class, bias, residual, dropout seed, row maxima); no split-K (these products have K <= 2048), tile 256 x 128.
namespace {
bool group4_desc_ok(const grappa_gemm_desc& d, const grappa_gemm_desc& first) {
    if (!d.A || !d.B || !d.C || d.Cp || d.C1p || d.resp || d.auxp || d.a_colsum) return false;
    if (d.M <= 32 || d.N <= 32 || d.K <= 0 || !d.a_kcontig) return false;
    if (d.precision != first.precision || d.precision == GRAPPA_GEMM_F32_MFMA || d.b_kcontig != first.b_kcontig) return false;
    if ((d.a_planes != 0) != (first.a_planes != 0) || (d.b_planes != 0) != (first.b_planes != 0)) return false;
    if (d.drop_p < 0.0f || d.drop_p >= 1.0f) return false;
    const bool pairs = d.a_planes && d.b_planes;
    if ((d.a_planes || d.b_planes) && !(pairs && d.precision == GRAPPA_GEMM_F32_F16X3 && d.b_kcontig && !d.amax_bcast)) return false;
    if (d.precision == GRAPPA_GEMM_F32_F16X3 && (!d.a_amax || !d.b_amax)) return false;
    if (pairs) {
        const int kpad = (d.K + 31) / 32 * 32;
        auto ok = [](const void* q, int ld, int cols) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0 && (ld & 7) == 0 && ld >= cols; };
        if (!ok(d.A, d.lda, 2 * kpad) || !ok(d.B, d.ldb, 2 * kpad)) return false;
        if ((size_t)d.M * d.lda * 2 >= (1ull << 32) || (size_t)d.N * d.ldb * 2 >= (1ull << 32)) return false;
    } else {
        // 16-byte loads: K-contiguous operands along k (K % 4 == 0), the row-contiguous weight of the input gradient along its rows
        const bool vecA = (reinterpret_cast<uintptr_t>(d.A) & 15) == 0 && (d.lda & 3) == 0 && (d.K & 3) == 0;
        const bool vecB = (reinterpret_cast<uintptr_t>(d.B) & 15) == 0 && (d.ldb & 3) == 0 && (!d.b_kcontig || (d.K & 3) == 0);
        const bool padB = d.b_kcontig || ((d.N + 3) & ~3) <= d.ldb;
        if (!(vecA && vecB && padB)) return false;
    }
    if (d.out_amax_parts && d.out_amax) return false;
    if (d.a_amax_nseg > 1 && (d.a_planes || (d.amax_bcast & 1))) return false;
    if (d.res_ln_mean && (!d.res || !d.res_ln_rstd || !d.res_ln_gamma || !d.res_ln_beta || (d.N & 3) || d.aux ||
                          ((reinterpret_cast<uintptr_t>(d.res_ln_gamma) | reinterpret_cast<uintptr_t>(d.res_ln_beta)) & 15) != 0))
        return false;
    return true;
}
}  // namespace

extern "C" size_t grappa_gemm_f32_group_workspace_bytes(const grappa_gemm_desc* descs, int n) {
    if (!descs || n <= 0 || n > GEMM_GROUP4_MAX) return 0;
    size_t need = 0;
    for (int i = 0; i < n; ++i)
        if (descs[i].out_amax && !descs[i].out_amax_parts) need += (amax_part_bytes(descs[i].M, descs[i].N) + 255) / 256 * 256;
    return need;
}

extern "C" int grappa_gemm_f32_group(void* stream, const grappa_gemm_desc* descs, int n, void* ws, size_t ws_bytes) {
    if (!descs || n <= 0 || n > GEMM_GROUP4_MAX) return GRAPPA_ERR_ARG;
    for (int i = 0; i < n; ++i)
        if (!group4_desc_ok(descs[i], descs[0])) return GRAPPA_ERR_ARG;
    if (ws_bytes < grappa_gemm_f32_group_workspace_bytes(descs, n) || (ws_bytes > 0 && !ws)) return GRAPPA_ERR_WORKSPACE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    auto al16 = [](const void* q, int ld) { return q == nullptr || ((reinterpret_cast<uintptr_t>(q) & 15) == 0 && (ld & 3) == 0); };
    GemmGroup4 g;
    g.count = n;
    g.wg_begin[0] = 0;
    char* part = reinterpret_cast<char*>(ws);
    for (int i = 0; i < n; ++i) {
        const grappa_gemm_desc& d = descs[i];
        GemmParams& p = g.p[i];
        p.d = d;
        p.bm = 256;
        p.bn = 128;
        p.tiles_m = (d.M + 255) / 256;
        p.tiles_n = (d.N + 127) / 128;
        p.tile_begin = 0;
        p.ntiles_launch = p.tiles_m * p.tiles_n;
        p.nsplit = 1;
        p.k_per_split = (d.K + BK - 1) / BK * BK;
        p.slab = nullptr;
        p.cs_slab = nullptr;
        p.tickets = nullptr;
        p.drop_scale = d.drop_p > 0.0f ? 1.0f / (1.0f - d.drop_p) : 1.0f;
        p.drop_salt = d.drop_salt;
        p.vec_io = al16(d.C, d.ldc) && al16(d.C2, d.ldc2) && al16(d.pre, d.ldpre) && al16(d.res, d.ldres) && al16(d.aux, d.ldaux);
        p.epi_class = choose_epi_class(d, p.vec_io != 0, true);
        p.amax_seg = 32;
        p.amax_part = nullptr;
        if (d.out_amax_parts) {
            p.amax_part = d.out_amax_parts;
        } else if (d.out_amax) {
            p.amax_part = reinterpret_cast<unsigned*>(part);
            part += (amax_part_bytes(d.M, d.N) + 255) / 256 * 256;
        }
        g.wg_begin[i + 1] = g.wg_begin[i] + p.ntiles_launch;
    }
    for (int i = n; i < GEMM_GROUP4_MAX; ++i) g.wg_begin[i + 1] = g.wg_begin[n];
    int rc = (descs[0].a_planes && descs[0].b_planes) ? grappa_launch_gemm_pairs_group4(st, g)
                                                      : grappa_launch_gemm_bf16x_group4(st, g, descs[0].precision, descs[0].b_kcontig != 0);
    if (rc == GRAPPA_OK) {                       // the row maxima of every product that asked for them: one combine launch
        int cnt = 0, Ms[4], segs[4];
        const unsigned* parts[4];
        unsigned* outs[4];
        for (int i = 0; i < n; ++i)
            if (descs[i].out_amax && !descs[i].out_amax_parts) {
                Ms[cnt] = descs[i].M;
                segs[cnt] = (descs[i].N + 31) / 32;
                parts[cnt] = g.p[i].amax_part;
                outs[cnt] = descs[i].out_amax;
                ++cnt;
            }
        if (cnt) rc = grappa_launch_amax_combine4(st, cnt, Ms, segs, parts, outs);
    }
    return rc;
}
