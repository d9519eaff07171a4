// Shared pieces of the GEMM kernels (native fp32 MFMA: gemm_f32.hip; bf16-split emulation / bf16: gemm_bf16x.hip):
// launch parameters, the XCD-aware workgroup -> (split, tile) map, the fused epilogue and the split-K slab reduction.
#pragma once
#include "common.h"

// residual = LayerNorm(res) recomputed by the epilogue (grappa_gemm_desc.res_ln_*): compiled out of the translation units whose kernels
// never get it (gemm_planes.hip: its one-plane kernel lives on <= 128 registers for two workgroups per CU, and the branch costs 10)
#ifndef GRAPPA_EPI_RES_LN
#define GRAPPA_EPI_RES_LN 1
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef GB_KNOCK
#define GB_KNOCK 0            // timing experiments (gemm_bf16x_impl.h): 7 = the row epilogue issues no global stores
#endif
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
    int vec_io;            // every epilogue tensor is 16-byte aligned with a leading dimension % 4 == 0 (float4 epilogue)
    // d.out_amax: row maxima of OUT are first written per column segment (amax_seg = 32 columns wide), amax_part[seg * M + m] -- plain stores, no atomics (device-scope atomics cost more than the whole pass they replace) --
    // and combined by amax_combine_kernel after the product's launches
    unsigned* amax_part;
    int amax_seg;
    // straight-line row epilogue for the common cases (host-chosen, epilogue_band_fast): 0 = the general walk; 1 = bias;
    // 2 = bias + ELU; 3 = bias + dropout + residual; 4 = ELU' from the saved output (aux), optional residual; + 8: the same with
    // bf16 output / residual / aux (the bf16 storage configuration)
    int epi_class;
    // split-K finished inside the product's own launch: tickets[tile_local] counts the workgroups of a tile that have written their
    // slab; the last one to arrive sums the slabs in the fixed order s = 0, 1, ... (the same bits whichever workgroup that is) and
    // runs the epilogue.  nullptr: the host launches gemm_splitk_reduce_kernel instead (native fp32 and plane kernels)
    int* tickets;
    const uint64_t* drop_salt;      // device word mixed into d.drop_seed (grappa_set_dropout_salt), or nullptr
};
// several independent forward / input-gradient products in ONE launch (grappa_gemm_f32_group: the same product of the four writer heads):
// descriptors and the prefix of workgroups per product travel as the kernel's argument
constexpr int GEMM_GROUP4_MAX = 4;
struct GemmGroup4 {
    GemmParams p[GEMM_GROUP4_MAX];
    int wg_begin[GEMM_GROUP4_MAX + 1];
    int count;
};
__device__ inline int group4_find(const GemmGroup4& g, int wg) {
    int i = 0;
    while (i + 1 < g.count && wg >= g.wg_begin[i + 1]) ++i;
    return i;
}
// fp32 A + weight pairs: does EVERY K range of a launch (nsplit ranges of k_per_split columns, the last one shorter) hold the four 16-column slabs
// of the pinned pipeline (gemm_wpairs_il.hip)?  One predicate for the dispatch (gemm_pairs.hip) and for the host's check of per-segment operand
// maxima (gemm_f32.hip, a_amax_nseg): only that kernel combines them
inline bool grappa_wpairs_il_takes(int K, int nsplit, int k_per_split) {
    const int last = K - (nsplit - 1) * k_per_split;
    return (K & 31) == 0 && (k_per_split & 31) == 0 && last >= 64;
}
__device__ inline uint64_t drop_seed_of(const GemmParams& p) { return grappa_salted(p.d.drop_seed, p.drop_salt); }

// ---- plane format (include/grappa_hip.h): X = P0 + P1 + P2, three bf16 planes
__device__ inline float bf16_bits_to_f32(unsigned h) { return __uint_as_float(h << 16); }
__device__ inline float planes_load1(const uint16_t* __restrict__ p, size_t stride, size_t idx, int np = 3) {
    if (np == 1) return bf16_bits_to_f32(p[idx]);
    return (bf16_bits_to_f32(p[idx]) + bf16_bits_to_f32(p[idx + stride])) + bf16_bits_to_f32(p[idx + 2 * stride]);   // exact: 8 + 8 + 8 bits
}
// four consecutive elements (8-byte aligned)
__device__ inline void planes_load4(const uint16_t* __restrict__ p, size_t stride, size_t idx, float (&v)[4], int np = 3) {
    if (np == 1) {
        const uint2 a = *reinterpret_cast<const uint2*>(p + idx);
        v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
        v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
        return;
    }
    const uint2 a = *reinterpret_cast<const uint2*>(p + idx), b = *reinterpret_cast<const uint2*>(p + idx + stride),
                c = *reinterpret_cast<const uint2*>(p + idx + 2 * stride);
    v[0] = (__uint_as_float(a.x << 16) + __uint_as_float(b.x << 16)) + __uint_as_float(c.x << 16);
    v[1] = (__uint_as_float(a.x & 0xffff0000u) + __uint_as_float(b.x & 0xffff0000u)) + __uint_as_float(c.x & 0xffff0000u);
    v[2] = (__uint_as_float(a.y << 16) + __uint_as_float(b.y << 16)) + __uint_as_float(c.y << 16);
    v[3] = (__uint_as_float(a.y & 0xffff0000u) + __uint_as_float(b.y & 0xffff0000u)) + __uint_as_float(c.y & 0xffff0000u);
}
typedef __bf16 grappa_bf16x2 __attribute__((ext_vector_type(2)));
__device__ inline void planes_store4(uint16_t* __restrict__ p, size_t stride, size_t idx, const float (&v)[4], int np = 3) {
    float r[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        if (pl >= np) break;
        grappa_bf16x2 h01, h23;
        h01[0] = (__bf16)r[0]; h01[1] = (__bf16)r[1];
        h23[0] = (__bf16)r[2]; h23[1] = (__bf16)r[3];
        const unsigned u01 = __builtin_bit_cast(unsigned, h01), u23 = __builtin_bit_cast(unsigned, h23);
        *reinterpret_cast<uint2*>(p + idx + pl * stride) = make_uint2(u01, u23);
        if (pl < 2) {
            r[0] -= __uint_as_float(u01 << 16); r[1] -= __uint_as_float(u01 & 0xffff0000u);
            r[2] -= __uint_as_float(u23 << 16); r[3] -= __uint_as_float(u23 & 0xffff0000u);
        }
    }
}
__device__ inline void planes_store1(uint16_t* __restrict__ p, size_t stride, size_t idx, float v, int np = 3) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        if (pl >= np) break;
        const __bf16 h = (__bf16)v;
        p[idx + pl * stride] = __builtin_bit_cast(uint16_t, h);
        v -= (float)h;
    }
}

// returns the value OUT(m, n) received
__device__ inline float epilogue_store(const GemmParams& p, int m, int n, float v) {
    const grappa_gemm_desc& d = p.d;
    if (d.pre) v += d.pre[(size_t)m * d.ldpre + n];
    if (d.bias) v += d.bias[n];
    if (d.act == GRAPPA_ACT_ELU) v = grappa_elu(v);
    if (d.aux) v *= grappa_elu_grad_from_out(d.aux[(size_t)m * d.ldaux + n]);
    else if (d.auxp) v *= grappa_elu_grad_from_out(planes_load1(d.auxp, d.auxp_plane_stride, (size_t)m * d.ldauxp + n, d.auxp_nplanes ? d.auxp_nplanes : 3));
    float* out = d.C;
    int ldo = d.ldc;
    if (d.C2) {
        d.C[(size_t)m * d.ldc + n] = v;
        out = d.C2;
        ldo = d.ldc2;
    }
    if (d.C1p) planes_store1(d.C1p, 0, (size_t)m * d.ldc1p + n, v, 1);
    if (d.drop_p > 0.0f) v = grappa_keep(drop_seed_of(p), (uint64_t)m * (uint64_t)d.N + (uint64_t)n, d.drop_p) ? v * p.drop_scale : 0.0f;
    if (GRAPPA_EPI_RES_LN && d.res && d.res_ln_mean) v += grappa_ln_apply(d.res[(size_t)m * d.ldres + n], d.res_ln_mean[m], d.res_ln_rstd[m], d.res_ln_gamma[n], d.res_ln_beta[n]);
    else if (d.res) v += d.res[(size_t)m * d.ldres + n];
    else if (d.resp) v += planes_load1(d.resp, d.resp_plane_stride, (size_t)m * d.ldresp + n, d.resp_nplanes ? d.resp_nplanes : 3);
    if (out) {
        float* o = out + (size_t)m * ldo + n;
        if (d.accumulate) v += *o;
        *o = v;
    }
    if (d.Cp) planes_store1(d.Cp, d.cp_plane_stride, (size_t)m * d.ldcp + n, v, d.cp_nplanes ? d.cp_nplanes : 3);
    return v;
}
__device__ inline unsigned mag_bits(float v) { return __float_as_uint(v) & 0x7fffffffu; }
// max over aligned groups of G = 16 or 8 lanes, valid in every lane, on the DPP data path (a __shfl_xor is an LDS crossbar round trip:
// four dependent ones per row made the row maxima cost more than the pass they replace)
template <int G>
__device__ inline unsigned group_umax(unsigned v) {
#define GRAPPA_DPP_MAX(CTRL) v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true))
    if (G == 16) {
        GRAPPA_DPP_MAX(0x128);      // row_ror:8
        GRAPPA_DPP_MAX(0x124);      // row_ror:4
        GRAPPA_DPP_MAX(0x122);      // row_ror:2
        GRAPPA_DPP_MAX(0x121);      // row_ror:1
    } else {
        GRAPPA_DPP_MAX(0x141);      // row_half_mirror: lane i <-> 7 - i
        GRAPPA_DPP_MAX(0xB1);       // quad_perm [1, 0, 3, 2]
        GRAPPA_DPP_MAX(0x4E);       // quad_perm [2, 3, 0, 1]
    }
#undef GRAPPA_DPP_MAX
    return v;
}


// XCD-aware bijective remap of the linear workgroup id (blocks b and b+8 share an XCD): consecutive LOGICAL ids run on one XCD.
// Logical order = (split, tile_m, tile_n) with tile_n fastest, so the workgroups that share an XCD's L2 are the column tiles
// of one row panel and, for split-K, the tiles of one K-slice (they re-read the same operand rows).
struct TileCoord {
    int split, tile_local, tile_m, tile_n;
};
__device__ inline TileCoord map_logical(const GemmParams& p, int nwg, int orig) {
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
__device__ inline TileCoord map_workgroup(const GemmParams& p) { return map_logical(p, gridDim.x, blockIdx.x); }

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
                if (d.drop_p > 0.0f) v = grappa_keep(drop_seed_of(p), drow + (uint64_t)n, d.drop_p) ? v * p.drop_scale : 0.0f;
                if (res_r) v += res_r[n];
                if (d.accumulate) v += *o;
                *o = v;
            }
        }
}

// Row-contiguous epilogue for kernels that compute the TRANSPOSED accumulator (MFMA called as B-fragment x A-fragment, so that
// element e of lane (lr, lh) is C(m = lr, n = (e & 3) + 8*(e >> 2) + 4*lh): four consecutive n per register quad).
// Per 32-row band the wavefront stages its 32 x (32*TN) block in a private LDS region with 16-byte writes, reads it back one
// full row segment per 16 lanes and walks C / pre / aux / res / C2 with float4 accesses: 4x fewer memory instructions than the
// element-wise walk and every store instruction covers whole 128-byte lines.  `vec_io` (host-checked: all epilogue tensors
// 16-byte aligned with leading dimensions % 4 == 0) selects the float4 path; otherwise elements go through epilogue_store.
constexpr int EPI_LD = 68;                          // floats per staged row (64 + 4 pad: conflict-free 16-byte writes)
constexpr int EPI_WAVE_BYTES = 32 * EPI_LD * 4;     // private staging region per wavefront

// The common cases of the row epilogue as straight-line code.  The general walk below decides ~30 uniform branches per trip (which
// tensors exist, their element types, ...): at K = 512 that walk was a quarter of the kernel (knock-outs: profiles/r2z_gemm_f16x3_*).
// Here the case is a template parameter, every LDS read and every load of res / aux is issued before the first use, and a lane works
// on whole float4s only (host: N % 4 == 0, all tensors 16-byte aligned, fp32, one output tensor, no pre-activation addend).
#ifndef GB_NT_STORE
#define GB_NT_STORE 2        // 1 = nontemporal stores of the fp32 result, 2 = + nontemporal loads of residual / saved activation (each read once): the
                             // 4 MB L2 of an XCD keeps the operand panels instead (C2 step -0.3 .. -0.5 ms, tools/nt_store_ab.sh; 0 = plain)
#endif
//   CLS 1: v + bias            2: elu(v + bias)            3: drop(v + bias) + res (drop_p may be 0, res may be NULL)
//   CLS 4: v * elu'(aux) (+ res)
//   CLS 5: drop(v + bias) + LayerNorm(res) (fp32 only: grappa_gemm_desc.res_ln_*, the residual recomputed from the rows before normalisation)
template <int TN, int CLS, typename T, int HB, int LD = EPI_LD>
__device__ __forceinline__ void epilogue_band_fast(const GemmParams& p, const f32x16 (&acc_i)[TN], float* __restrict__ wave_buf, int mband, int n, int lane,
                                                   const float4& b4) {
    const grappa_gemm_desc& d = p.d;
    constexpr bool BF = sizeof(T) == 2;                      // the bf16 storage configuration: one-plane (= plain bf16) output / res / aux
    const int lr = lane & 31, lh = lane >> 5;
    constexpr int ROWS_PER_IT = TN == 2 ? 4 : 8;
    constexpr int NIT = 32 / ROWS_PER_IT;
    const int rrow = TN == 2 ? (lane >> 4) : (lane >> 3), rc4 = TN == 2 ? ((lane & 15) << 2) : ((lane & 7) << 2);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(wave_buf + lr * LD + j * 32 + 8 * g + 4 * lh) =
                make_float4(acc_i[j][4 * g], acc_i[j][4 * g + 1], acc_i[j][4 * g + 2], acc_i[j][4 * g + 3]);
    const bool col_ok = n < d.N;
    const int mfirst = mband + rrow;
    T* out = BF ? reinterpret_cast<T*>(d.Cp) : reinterpret_cast<T*>(d.C);
    const int ldo = BF ? d.ldcp : d.ldc;
    const T* resq = BF ? reinterpret_cast<const T*>(d.resp) : reinterpret_cast<const T*>(d.res);
    const int ldr = BF ? d.ldresp : d.ldres;
    const T* side = (CLS == 3 || CLS == 5) ? resq : (CLS == 4 ? (BF ? reinterpret_cast<const T*>(d.auxp) : reinterpret_cast<const T*>(d.aux)) : nullptr);
    const int ldside = (CLS == 3 || CLS == 5) ? ldr : (BF ? d.ldauxp : d.ldaux);
    float4 lng = make_float4(0.f, 0.f, 0.f, 0.f), lnb = lng;           // CLS 5: gamma, beta of this lane's four columns
    if (CLS == 5 && col_ok) {
        lng = *reinterpret_cast<const float4*>(d.res_ln_gamma + n);
        lnb = *reinterpret_cast<const float4*>(d.res_ln_beta + n);
    }
    // HB trips per batch: the LDS reads and the loads of a batch are in flight together (4: 48 registers; the one-plane kernel, which
    // lives on 128 registers for two workgroups per CU, takes 2)
#pragma unroll
    for (int h = 0; h < NIT / HB; ++h) {
        float4 v[HB], t[HB], r4[HB];
#pragma unroll
        for (int k = 0; k < HB; ++k) {
            const int it = h * HB + k, m = mfirst + it * ROWS_PER_IT;
            v[k] = *reinterpret_cast<const float4*>(wave_buf + (it * ROWS_PER_IT + rrow) * LD + rc4);
            t[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            r4[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (GB_NT_STORE >= 2 && !BF) {
                typedef float nt_f32x4 __attribute__((ext_vector_type(4)));
                if ((CLS == 3 || CLS == 4 || CLS == 5) && side && col_ok && m < d.M) {
                    const nt_f32x4 w = __builtin_nontemporal_load(reinterpret_cast<const nt_f32x4*>(reinterpret_cast<const float*>(side) + (size_t)m * ldside + n));
                    t[k] = make_float4(w[0], w[1], w[2], w[3]);
                }
                if (CLS == 4 && resq && col_ok && m < d.M) {
                    const nt_f32x4 w = __builtin_nontemporal_load(reinterpret_cast<const nt_f32x4*>(reinterpret_cast<const float*>(resq) + (size_t)m * ldr + n));
                    r4[k] = make_float4(w[0], w[1], w[2], w[3]);
                }
            } else {
            if ((CLS == 3 || CLS == 4 || CLS == 5) && side && col_ok && m < d.M) t[k] = ld4(side + (size_t)m * ldside + n, 0);
            if (CLS == 4 && resq && col_ok && m < d.M) r4[k] = ld4(resq + (size_t)m * ldr + n, 0);
            }
        }
#pragma unroll
        for (int k = 0; k < HB; ++k) {
            const int it = h * HB + k, m = mfirst + it * ROWS_PER_IT;
            float x[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
            if (CLS != 4) { x[0] += b4.x; x[1] += b4.y; x[2] += b4.z; x[3] += b4.w; }
            if (CLS == 2) {
#pragma unroll
                for (int q = 0; q < 4; ++q) x[q] = grappa_elu(x[q]);
            }
            if (CLS == 5 && col_ok && m < d.M) {                       // the residual is LayerNorm of the rows just loaded
                const float mu = d.res_ln_mean[m], rs = d.res_ln_rstd[m];
                t[k] = make_float4(grappa_ln_apply(t[k].x, mu, rs, lng.x, lnb.x), grappa_ln_apply(t[k].y, mu, rs, lng.y, lnb.y),
                                   grappa_ln_apply(t[k].z, mu, rs, lng.z, lnb.z), grappa_ln_apply(t[k].w, mu, rs, lng.w, lnb.w));
            }
            if (CLS == 3 || CLS == 5) {
                if (d.drop_p > 0.0f) {
                    const uint64_t idx = (uint64_t)m * (uint64_t)d.N + (uint64_t)n;
#pragma unroll
                    for (int q = 0; q < 4; ++q) x[q] = grappa_keep(drop_seed_of(p), idx + q, d.drop_p) ? x[q] * p.drop_scale : 0.0f;
                }
                x[0] += t[k].x; x[1] += t[k].y; x[2] += t[k].z; x[3] += t[k].w;
            }
            if (CLS == 4) {
                x[0] = x[0] * grappa_elu_grad_from_out(t[k].x) + r4[k].x; x[1] = x[1] * grappa_elu_grad_from_out(t[k].y) + r4[k].y;
                x[2] = x[2] * grappa_elu_grad_from_out(t[k].z) + r4[k].z; x[3] = x[3] * grappa_elu_grad_from_out(t[k].w) + r4[k].w;
            }
            const bool ok = col_ok && m < d.M;
            if (ok) {
                if (GB_NT_STORE && !BF) {
                    typedef float nt_f32x4 __attribute__((ext_vector_type(4)));
                    nt_f32x4 w = {x[0], x[1], x[2], x[3]};
                    __builtin_nontemporal_store(w, reinterpret_cast<nt_f32x4*>(reinterpret_cast<float*>(out) + (size_t)m * ldo + n));
                } else {
                    st4(out + (size_t)m * ldo + n, 0, make_float4(x[0], x[1], x[2], x[3]));
                }
            }
            if (!BF && p.amax_part) {
                unsigned am = ok ? max(max(mag_bits(x[0]), mag_bits(x[1])), max(mag_bits(x[2]), mag_bits(x[3]))) : 0u;
                am = group_umax<8>(am);                              // segments of 32 columns (8 lanes x 4), whatever the tile: the consumer
                const int nseg0 = n - ((lane & 7) << 2);             // of the partials need not know which tile wrote them
                if ((lane & 7) == 0 && m < d.M && nseg0 < d.N) p.amax_part[(size_t)(nseg0 / p.amax_seg) * d.M + m] = am;
            }
        }
    }
}

// one 32-row band (accumulator row i of the wavefront's TM x TN grid)
template <int BM, int BN, int TN, int LD = EPI_LD>
__device__ __forceinline__ void epilogue_band(const GemmParams& p, const f32x16 (&acc_i)[TN], float* __restrict__ wave_buf, int m0, int n0, int mband,
                                     int n, int lane, const float4& b4, int split, int tile_local, bool vec_io) {
    const grappa_gemm_desc& d = p.d;
    const int lr = lane & 31, lh = lane >> 5;
    constexpr int ROWS_PER_IT = TN == 2 ? 4 : 8;                // 16 (TN == 2) or 8 lanes read one staged row
    constexpr int NIT = 32 / ROWS_PER_IT;
    const int rrow = TN == 2 ? (lane >> 4) : (lane >> 3), rc4 = TN == 2 ? ((lane & 15) << 2) : ((lane & 7) << 2);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(wave_buf + lr * LD + j * 32 + 8 * g + 4 * lh) =
                make_float4(acc_i[j][4 * g], acc_i[j][4 * g + 1], acc_i[j][4 * g + 2], acc_i[j][4 * g + 3]);
    for (int it = 0; it < NIT; ++it) {
        const int row = it * ROWS_PER_IT + rrow;
        const float4 v4 = *reinterpret_cast<const float4*>(wave_buf + row * LD + rc4);
        const int m = mband + row;
        unsigned am = 0u;                                    // largest |OUT(m, n .. n+3)| of this lane (d.out_amax)
        if (m < d.M && n < d.N) {
            if (p.nsplit > 1) {                              // raw partial sums, tile-local [BM][BN] slab (always aligned)
                float* srow = p.slab + ((size_t)split * p.ntiles_launch + tile_local) * (BM * BN) + (size_t)(m - m0) * BN + (n - n0);
                *reinterpret_cast<float4*>(srow) = v4;
            } else if (!vec_io || n + 3 >= d.N) {
                const float ve[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (n + q < d.N) am = max(am, mag_bits(epilogue_store(p, m, n + q, ve[q])));
            } else {
                float v[4] = {v4.x, v4.y, v4.z, v4.w};
                if (d.pre) {
                    const float4 t = *reinterpret_cast<const float4*>(d.pre + (size_t)m * d.ldpre + n);
                    v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
                }
                v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
                if (d.act == GRAPPA_ACT_ELU) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = grappa_elu(v[q]);
                }
                if (d.aux) {
                    const float4 t = *reinterpret_cast<const float4*>(d.aux + (size_t)m * d.ldaux + n);
                    v[0] *= grappa_elu_grad_from_out(t.x); v[1] *= grappa_elu_grad_from_out(t.y);
                    v[2] *= grappa_elu_grad_from_out(t.z); v[3] *= grappa_elu_grad_from_out(t.w);
                } else if (d.auxp) {
                    float t[4];
                    planes_load4(d.auxp, d.auxp_plane_stride, (size_t)m * d.ldauxp + n, t, d.auxp_nplanes ? d.auxp_nplanes : 3);
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] *= grappa_elu_grad_from_out(t[q]);
                }
                float* o = d.C ? d.C + (size_t)m * d.ldc + n : nullptr;
                if (d.C2) {
                    *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
                    o = d.C2 + (size_t)m * d.ldc2 + n;
                }
                if (d.C1p) planes_store4(d.C1p, 0, (size_t)m * d.ldc1p + n, v, 1);
                if (d.drop_p > 0.0f) {
                    const uint64_t idx = (uint64_t)m * (uint64_t)d.N + (uint64_t)n;
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = grappa_keep(drop_seed_of(p), idx + q, d.drop_p) ? v[q] * p.drop_scale : 0.0f;
                }
                if (GRAPPA_EPI_RES_LN && d.res && d.res_ln_mean) {
                    const float4 t = *reinterpret_cast<const float4*>(d.res + (size_t)m * d.ldres + n);
                    const float4 g = *reinterpret_cast<const float4*>(d.res_ln_gamma + n), be = *reinterpret_cast<const float4*>(d.res_ln_beta + n);
                    const float mu = d.res_ln_mean[m], rs = d.res_ln_rstd[m];
                    v[0] += grappa_ln_apply(t.x, mu, rs, g.x, be.x); v[1] += grappa_ln_apply(t.y, mu, rs, g.y, be.y);
                    v[2] += grappa_ln_apply(t.z, mu, rs, g.z, be.z); v[3] += grappa_ln_apply(t.w, mu, rs, g.w, be.w);
                } else if (d.res) {
                    const float4 t = *reinterpret_cast<const float4*>(d.res + (size_t)m * d.ldres + n);
                    v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
                } else if (d.resp) {
                    float t[4];
                    planes_load4(d.resp, d.resp_plane_stride, (size_t)m * d.ldresp + n, t, d.resp_nplanes ? d.resp_nplanes : 3);
                    v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
                }
                if (o && !(GB_KNOCK == 7 && v[0] != 123.456f)) {
                    if (d.accumulate) {
                        const float4 t = *reinterpret_cast<const float4*>(o);
                        v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
                    }
                    *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
                }
                if (d.Cp) planes_store4(d.Cp, d.cp_plane_stride, (size_t)m * d.ldcp + n, v, d.cp_nplanes ? d.cp_nplanes : 3);
                am = max(max(mag_bits(v[0]), mag_bits(v[1])), max(mag_bits(v[2]), mag_bits(v[3])));
            }
        }
        if (p.amax_part && p.nsplit == 1) {
            // the 8 lanes that hold 32 columns of a row combine; segment-major layout: the 4 / 8 rows of a trip are consecutive words,
            // the 32 rows of the band one 128-byte line
            am = group_umax<8>(am);
            const int nseg0 = n - ((lane & 7) << 2);         // first column of this lane group's segment
            if ((lane & 7) == 0 && m < d.M && nseg0 < d.N) p.amax_part[(size_t)(nseg0 / p.amax_seg) * d.M + m] = am;
        }
    }
}

template <int BM, int BN, int TM, int TN, int EPI_BATCH = 4>
__device__ __forceinline__ void tile_epilogue_rows(const GemmParams& p, const f32x16 (&acc)[TM][TN], float* __restrict__ wave_buf, int m0, int n0,
                                          int wm0, int wn0, int lane, int split, int tile_local, bool vec_io) {
    static_assert(TN * 32 <= 64 && TM == 2, "staging row holds 64 floats; two bands per wavefront");
    const grappa_gemm_desc& d = p.d;
    const int n = n0 + wn0 + (TN == 2 ? ((lane & 15) << 2) : ((lane & 7) << 2));
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d.bias && p.nsplit == 1) {
        b4.x = n < d.N ? d.bias[n] : 0.f;
        b4.y = n + 1 < d.N ? d.bias[n + 1] : 0.f;
        b4.z = n + 2 < d.N ? d.bias[n + 2] : 0.f;
        b4.w = n + 3 < d.N ? d.bias[n + 3] : 0.f;
    }
#ifndef GRAPPA_NO_FAST_EPI
    if (p.epi_class != 0 && p.nsplit == 1) {
        const int mb = m0 + wm0;
#define GRAPPA_FAST(CLS, T) epilogue_band_fast<TN, CLS, T, EPI_BATCH>(p, acc[0], wave_buf, mb, n, lane, b4); epilogue_band_fast<TN, CLS, T, EPI_BATCH>(p, acc[1], wave_buf, mb + 32, n, lane, b4); break
        switch (p.epi_class) {
            case 1: GRAPPA_FAST(1, float);
            case 2: GRAPPA_FAST(2, float);
            case 3: GRAPPA_FAST(3, float);
            case 4: GRAPPA_FAST(4, float);
#if GRAPPA_EPI_RES_LN
            case 5: GRAPPA_FAST(5, float);
#endif
            case 9: GRAPPA_FAST(1, grappa_bf16_t);
            case 10: GRAPPA_FAST(2, grappa_bf16_t);
            case 11: GRAPPA_FAST(3, grappa_bf16_t);
            default: GRAPPA_FAST(4, grappa_bf16_t);
        }
#undef GRAPPA_FAST
        return;
    }
#endif
    epilogue_band<BM, BN, TN>(p, acc[0], wave_buf, m0, n0, m0 + wm0, n, lane, b4, split, tile_local, vec_io);
    epilogue_band<BM, BN, TN>(p, acc[1], wave_buf, m0, n0, m0 + wm0 + 32, n, lane, b4, split, tile_local, vec_io);
}

constexpr int REDUCE_THREADS = 256;

__device__ inline void splitk_reduce_body(const GemmParams& p, int nblocks, int block);

static __global__ __launch_bounds__(REDUCE_THREADS) void gemm_splitk_reduce_kernel(GemmParams p) { splitk_reduce_body(p, gridDim.x, blockIdx.x); }

// grouped: blocks [blk_begin[g], blk_begin[g+1]) reduce problem g (problems with nsplit == 1 own no blocks)
static __global__ __launch_bounds__(REDUCE_THREADS) void gemm_splitk_reduce_grouped_kernel(const GemmParams* __restrict__ ps,
                                                                                          const int* __restrict__ blk_begin, int nprob) {
    const int b = blockIdx.x;
    int g = 0;
    while (g + 1 < nprob && b >= blk_begin[g + 1]) ++g;
    splitk_reduce_body(ps[g], blk_begin[g + 1] - blk_begin[g], b - blk_begin[g]);
}

// sums of one quad (four consecutive columns of a tile row, flat index i into the launch's [ntiles_launch][bm * bn] slab) over the
// splits in the fixed order s = 0, 1, ..., epilogue, row-segment maximum
__device__ __forceinline__ void splitk_reduce_quad(const GemmParams& p, size_t split_stride, int tile_elems, size_t i) {
    const int tl = (int)(i / tile_elems), rem = (int)(i - (size_t)tl * tile_elems);
    const int tile = p.tile_begin + tl;
    const int m = (tile / p.tiles_n) * p.bm + rem / p.bn, n = (tile % p.tiles_n) * p.bn + rem % p.bn;
    unsigned am = 0u;
    if (m < p.d.M && n < p.d.N) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
        for (int s = 0; s < p.nsplit; ++s) {
            const float4 t = *reinterpret_cast<const float4*>(p.slab + (size_t)s * split_stride + i);
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        const float ve[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (n + q < p.d.N) am = max(am, mag_bits(epilogue_store(p, m, n + q, ve[q])));
    }
    if (p.amax_part) {
        // amax_seg / 4 adjacent lanes (8 or 16, aligned: bn and the block size are multiples of 64 columns) share one row segment
        for (int o = p.amax_seg >> 3; o > 0; o >>= 1) am = max(am, (unsigned)__shfl_xor((int)am, o, 64));
        if ((threadIdx.x & ((p.amax_seg >> 2) - 1)) == 0 && m < p.d.M && n < p.d.N) p.amax_part[(size_t)(n / p.amax_seg) * p.d.M + m] = am;
    }
}

// column-sum partials of the rows of launch tile tl (written by the tile_n == 0 workgroups)
__device__ __forceinline__ void splitk_reduce_colsum(const GemmParams& p, int tl, int nthreads) {
    const int tile = p.tile_begin + tl;
    if (tile % p.tiles_n != 0) return;
    const int m0 = (tile / p.tiles_n) * p.bm;
    for (int mi = threadIdx.x; mi < p.bm; mi += nthreads) {
        const int m = m0 + mi;
        if (m >= p.d.M) continue;
        float v = 0.0f;
        for (int s = 0; s < p.nsplit; ++s) v += p.cs_slab[(size_t)s * p.d.M + m];
        p.d.a_colsum[m] += v;
    }
}

__device__ inline void splitk_reduce_body(const GemmParams& p, int nblocks, int block) {
    const int tile_elems = p.bm * p.bn;
    const size_t total = (size_t)p.ntiles_launch * tile_elems;
    if (p.d.a_colsum && block == 0)
        for (int tl = 0; tl < p.ntiles_launch; ++tl) splitk_reduce_colsum(p, tl, REDUCE_THREADS);
    // four consecutive columns of one tile row per thread (tiles are 16-byte aligned and bn % 4 == 0): 16-byte slab loads, several
    // in flight (the sum keeps its fixed order s = 0, 1, ...: reproducible)
    for (size_t i = ((size_t)block * REDUCE_THREADS + threadIdx.x) * 4; i < total; i += (size_t)nblocks * REDUCE_THREADS * 4)
        splitk_reduce_quad(p, total, tile_elems, i);
}

// Split-K finished by the product's own launch (p.tickets): called by every workgroup after its slab (and column-sum partial) is
// written.  Release (fence) -> ticket -> the last arrival acquires (fence) and reduces the tile.  No workgroup waits for another:
// the grid drains whatever the order of arrival.  s_last: one word of LDS no wavefront uses any more.
template <int NT>
#ifndef SK_EXP
#define SK_EXP 0          // timing experiments (tools/splitk_ab.sh): 1 = no fences (results not guaranteed), 2 = fences + ticket only (no sum: wrong results)
#endif
__device__ __forceinline__ void splitk_finish_tile(const GemmParams& p, int tile_local, volatile int* s_last) {
    if (SK_EXP != 1) __threadfence();
    __syncthreads();                                         // every wavefront's slab stores are fenced (and the LDS word is free)
    if (threadIdx.x == 0) *s_last = atomicAdd(p.tickets + tile_local, 1) == p.nsplit - 1;
    __syncthreads();
    if (!*s_last || SK_EXP == 2) return;
    if (SK_EXP != 1) __threadfence();
    const int tile_elems = p.bm * p.bn;
    const size_t total = (size_t)p.ntiles_launch * tile_elems;
    if (p.d.a_colsum) splitk_reduce_colsum(p, tile_local, NT);
    const size_t base = (size_t)tile_local * tile_elems;
    for (int i = threadIdx.x * 4; i < tile_elems; i += NT * 4) splitk_reduce_quad(p, total, tile_elems, base + i);
}


inline int launch_splitk_reduce(hipStream_t st, const GemmParams& p) {
    const size_t total = (size_t)p.ntiles_launch * p.bm * p.bn;
    int blocks = (int)((total / 4 + REDUCE_THREADS - 1) / REDUCE_THREADS);
    if (blocks > 4096) blocks = 4096;
    GRAPPA_LAUNCH(gemm_splitk_reduce_kernel, dim3(blocks), dim3(REDUCE_THREADS), 0, st, p);
    return grappa_launch_status();
}

}  // namespace grappa_gemm
