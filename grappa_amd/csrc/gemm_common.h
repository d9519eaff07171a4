// Shared pieces of the GEMM kernels (native fp32 MFMA: gemm_f32.hip; bf16-split emulation / bf16: gemm_bf16x.hip):
// launch parameters, the XCD-aware workgroup -> (split, tile) map, the fused epilogue and the split-K slab reduction.
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace grappa_gemm {

constexpr int GEMM_BK = 32;

struct GemmParams {
    grappa_gemm_desc d;
    int k_per_split;       // multiple of BK
    int nsplit;
    float* slab;           // [nsplit, ntiles_launch, BM*BN] tile-local partial sums when nsplit > 1
    float* cs_slab;        // [nsplit, M] column-sum partials when nsplit > 1 and d.a_colsum
    float drop_scale;
    int tiles_m, tiles_n;
    int tile_begin;        // this launch handles tiles [tile_begin, tile_begin + ntiles_launch) of the tiles_m x tiles_n grid
    int ntiles_launch;
    int bm, bn;            // tile shape (for the reduce kernel)
};

__device__ inline void epilogue_store(const GemmParams& p, int m, int n, float v) {
    const grappa_gemm_desc& d = p.d;
    if (d.pre) v += d.pre[(size_t)m * d.ldpre + n];
    if (d.bias) v += d.bias[n];
    if (d.act == GRAPPA_ACT_ELU) v = grappa_elu(v);
    if (d.aux) v *= grappa_elu_grad_from_out(d.aux[(size_t)m * d.ldaux + n]);
    float* out = d.C;
    int ldo = d.ldc;
    if (d.C2) {
        d.C[(size_t)m * d.ldc + n] = v;
        out = d.C2;
        ldo = d.ldc2;
    }
    if (d.drop_p > 0.0f) v = grappa_keep(d.drop_seed, (uint64_t)m * (uint64_t)d.N + (uint64_t)n, d.drop_p) ? v * p.drop_scale : 0.0f;
    if (d.res) v += d.res[(size_t)m * d.ldres + n];
    float* o = out + (size_t)m * ldo + n;
    if (d.accumulate) v += *o;
    *o = v;
}


// XCD-aware bijective remap of the linear workgroup id (blocks b and b+8 share an XCD): consecutive LOGICAL ids run on one XCD.
// Logical order = (split, tile_m, tile_n) with tile_n fastest, so the workgroups that share an XCD's L2 are the column tiles
// of one row panel and, for split-K, the tiles of one K-slice (they re-read the same operand rows).
struct TileCoord {
    int split, tile_local, tile_m, tile_n;
};
__device__ inline TileCoord map_workgroup(const GemmParams& p) {
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    TileCoord t;
    t.split = wgid / p.ntiles_launch;
    t.tile_local = wgid - t.split * p.ntiles_launch;
    const int tile = p.tile_begin + t.tile_local;
    t.tile_m = tile / p.tiles_n;
    t.tile_n = tile % p.tiles_n;
    return t;
}

// Epilogue of one wavefront's TM x TN grid of 32x32 accumulators (C/D layout: col = lane & 31,
// row = (e & 3) + 8*(e >> 2) + 4*(lane >> 5)).  Row-major walk: the 64-bit row offsets of every epilogue tensor are formed once
// per row, the per-column terms once per column.  With split-K the raw sums go to the tile-local slab instead.
template <int BM, int BN, int TM, int TN>
__device__ inline void tile_epilogue(const GemmParams& p, const f32x16 (&acc)[TM][TN], int m0, int n0, int wm0, int wn0, int lr, int lh,
                                     int split, int tile_local) {
    const grappa_gemm_desc& d = p.d;
    int ncol[TN];
    float bcol[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        ncol[j] = n0 + wn0 + j * 32 + lr;
        bcol[j] = (d.bias && ncol[j] < d.N) ? d.bias[ncol[j]] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = m0 + wm0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            if (m >= d.M) continue;
            if (p.nsplit > 1) {
                float* srow = p.slab + ((size_t)split * p.ntiles_launch + tile_local) * (BM * BN) + (size_t)(m - m0) * BN - n0;
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    if (ncol[j] < d.N) srow[ncol[j]] = acc[i][j][e];
                continue;
            }
            const float* pre_r = d.pre ? d.pre + (size_t)m * d.ldpre : nullptr;
            const float* aux_r = d.aux ? d.aux + (size_t)m * d.ldaux : nullptr;
            const float* res_r = d.res ? d.res + (size_t)m * d.ldres : nullptr;
            float* c_r = d.C + (size_t)m * d.ldc;
            float* c2_r = d.C2 ? d.C2 + (size_t)m * d.ldc2 : nullptr;
            const uint64_t drow = (uint64_t)m * (uint64_t)d.N;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = ncol[j];
                if (n >= d.N) continue;
                float v = acc[i][j][e];
                if (pre_r) v += pre_r[n];
                v += bcol[j];
                if (d.act == GRAPPA_ACT_ELU) v = grappa_elu(v);
                if (aux_r) v *= grappa_elu_grad_from_out(aux_r[n]);
                float* o = c_r + n;
                if (c2_r) {
                    *o = v;
                    o = c2_r + n;
                }
                if (d.drop_p > 0.0f) v = grappa_keep(d.drop_seed, drow + (uint64_t)n, d.drop_p) ? v * p.drop_scale : 0.0f;
                if (res_r) v += res_r[n];
                if (d.accumulate) v += *o;
                *o = v;
            }
        }
}

constexpr int REDUCE_THREADS = 256;

static __global__ __launch_bounds__(REDUCE_THREADS) void gemm_splitk_reduce_kernel(GemmParams p) {
    const int tile_elems = p.bm * p.bn;
    const size_t total = (size_t)p.ntiles_launch * tile_elems;
    const size_t split_stride = total;
    if (p.d.a_colsum && blockIdx.x == 0) {
        // column-sum partials were written by the tile_n == 0 workgroups of this launch
        for (int tl = 0; tl < p.ntiles_launch; ++tl) {
            const int tile = p.tile_begin + tl;
            if (tile % p.tiles_n != 0) continue;
            const int m0 = (tile / p.tiles_n) * p.bm;
            for (int mi = threadIdx.x; mi < p.bm; mi += REDUCE_THREADS) {
                const int m = m0 + mi;
                if (m >= p.d.M) continue;
                float v = 0.0f;
                for (int s = 0; s < p.nsplit; ++s) v += p.cs_slab[(size_t)s * p.d.M + m];
                p.d.a_colsum[m] += v;
            }
        }
    }
    for (size_t i = (size_t)blockIdx.x * REDUCE_THREADS + threadIdx.x; i < total; i += (size_t)gridDim.x * REDUCE_THREADS) {
        const int tl = (int)(i / tile_elems), rem = (int)(i - (size_t)tl * tile_elems);
        const int tile = p.tile_begin + tl;
        const int m = (tile / p.tiles_n) * p.bm + rem / p.bn, n = (tile % p.tiles_n) * p.bn + rem % p.bn;
        if (m >= p.d.M || n >= p.d.N) continue;
        float v = 0.0f;
        for (int s = 0; s < p.nsplit; ++s) v += p.slab[(size_t)s * split_stride + i];
        epilogue_store(p, m, n, v);
    }
}


inline int launch_splitk_reduce(hipStream_t st, const GemmParams& p) {
    const size_t total = (size_t)p.ntiles_launch * p.bm * p.bn;
    int blocks = (int)((total + REDUCE_THREADS - 1) / REDUCE_THREADS);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(blocks), dim3(REDUCE_THREADS), 0, st, p);
    return grappa_launch_status();
}

}  // namespace grappa_gemm
