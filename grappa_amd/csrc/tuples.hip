// n-body (tuple) stage kernels.  Token tables are (s*T, F) row-major with row = pos*T + t.
//  * tuple_gather fwd/bwd : index gather of projected atom rows (+ positional-encoding column); the backward
//    sums through the inverse incidence (atom -> token rows) -- deterministic, no atomics;
//  * seqattn fwd/bwd      : multi-head self-attention over the s <= 4 tokens of a tuple, one wavefront per
//    tuple, everything in registers (scores are s*s per head);
//  * perm_concat fwd/bwd  : the symmetriser's permuted concatenation;
//  * param_out fwd/bwd    : output maps ToPositive / ToRange / gated torsion / hard cutoff.
#include "common.h"

namespace {

struct Perms {
    int p[6][4];
};

// ------------------------------------------------------------------------------------------------ gather
template <typename TA, typename TX>
__global__ __launch_bounds__(256) void tuple_gather_fwd_kernel(int T, int s, int W, const TA* __restrict__ a, int lda,
                                                               const int* __restrict__ idx, const float* __restrict__ pe,
                                                               TX* __restrict__ x, int ldx) {
    const int lane = threadIdx.x & 63;
    const int nrows = s * T, nvec = W >> 2;
    const int wave0 = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
    for (int r = wave0; r < nrows; r += nw) {
        const int pos = r / T, t = r - pos * T;
        const int atom = idx[(size_t)t * s + pos];
        const TA* src = a + (size_t)atom * lda;
        TX* dst = x + (size_t)r * ldx;
        const float pev = pe ? pe[pos] : 0.f;
        for (int c = lane; c < nvec; c += 64) {
            float4 v = ld4(src, c);
            if (pe && c == nvec - 1) v.w = pev;
            st4(dst, c, v);
        }
    }
}

template <typename TX, typename TA>
__global__ __launch_bounds__(256) void tuple_gather_bwd_kernel(int N, int W, const int* __restrict__ inv_ptr, const int* __restrict__ inv_rows,
                                                               const TX* __restrict__ dx, int lddx, TA* __restrict__ da, int ldda,
                                                               int has_pe, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int nvec = W >> 2;
    const int n = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (n >= N) return;
    const int r0 = inv_ptr[n], r1 = inv_ptr[n + 1];
    TA* dst = da + (size_t)n * ldda;
    for (int c = lane; c < nvec; c += 64) {
        float4 acc = accumulate ? ld4(dst, c) : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = r0; j < r1; ++j) {
            float4 v = ld4(dx + (size_t)inv_rows[j] * lddx, c);
            if (has_pe && c == nvec - 1) v.w = 0.f;
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        st4(dst, c, acc);
    }
}

// ------------------------------------------------------------------------------------------------ attention
__device__ inline float dot4(const float4& a, const float4& b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }

// magnitudes as fp32 bit patterns with the sign cleared (amax.hip): the rows' largest |value| for a following F32_F16X3 product
__device__ inline unsigned mag4(const float4& v) {
    return max(max(__float_as_uint(v.x) & 0x7fffffffu, __float_as_uint(v.y) & 0x7fffffffu),
               max(__float_as_uint(v.z) & 0x7fffffffu, __float_as_uint(v.w) & 0x7fffffffu));
}
__device__ inline unsigned wave_umax(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o, 64));
    return v;
}

// row_amax (optional): largest |out| of each of the tuple's S token rows
template <int S, typename TE>
__device__ __forceinline__ void seqattn_fwd_body(int T, int F, int dh, const TE* __restrict__ qkv, TE* __restrict__ out,
                                                 unsigned* __restrict__ row_amax, int vblock) {
    const int lane = threadIdx.x & 63;
    const int t = (vblock * blockDim.x + threadIdx.x) >> 6;
    if (t >= T) return;
    const int nvec = F >> 2, lph = dh >> 2;
    const float scale = 1.0f / sqrtf((float)dh);
    unsigned am[S];
#pragma unroll
    for (int i = 0; i < S; ++i) am[i] = 0u;
    for (int c0 = 0; c0 < nvec; c0 += 64) {
        const int c = c0 + lane;
        const bool ok = c < nvec;
        float4 q[S], k[S], v[S];
#pragma unroll
        for (int i = 0; i < S; ++i) {
            const TE* row = qkv + ((size_t)i * T + t) * 3 * F;
            q[i] = ok ? ld4(row, c) : make_float4(0.f, 0.f, 0.f, 0.f);
            k[i] = ok ? ld4(row, nvec + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            v[i] = ok ? ld4(row, 2 * nvec + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < S; ++i) {
            float sc[S], mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < S; ++j) {
                sc[j] = group_sum(dot4(q[i], k[j]), lph) * scale;
                mx = fmaxf(mx, sc[j]);
            }
            float den = 0.f;
#pragma unroll
            for (int j = 0; j < S; ++j) {
                sc[j] = expf(sc[j] - mx);
                den += sc[j];
            }
            const float inv = 1.0f / den;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < S; ++j) {
                const float p = sc[j] * inv;
                o.x += p * v[j].x; o.y += p * v[j].y; o.z += p * v[j].z; o.w += p * v[j].w;
            }
            if (ok) {
                st4(out + ((size_t)i * T + t) * F, c, o);
                am[i] = max(am[i], mag4(o));
            }
        }
    }
    if (row_amax) {
#pragma unroll
        for (int i = 0; i < S; ++i) {
            const unsigned m = wave_umax(am[i]);
            if (lane == 0) row_amax[(size_t)i * T + t] = m;
        }
    }
}

template <int S, typename TE>
__global__ __launch_bounds__(256) void seqattn_fwd_kernel(int T, int F, int dh, const TE* __restrict__ qkv, TE* __restrict__ out,
                                                          unsigned* __restrict__ row_amax) {
    seqattn_fwd_body<S, TE>(T, F, dh, qkv, out, row_amax, blockIdx.x);
}

// the attention of several heads (different s, T) in ONE launch: C ABI 8 grappa_seqattn_fwd_batched_f32 / _bwd_
struct SeqAttnBatch {
    grappa_seqattn_item it[GRAPPA_ROW_BATCH_MAX];
    int blk_begin[GRAPPA_ROW_BATCH_MAX + 1];
    int count;
};
__global__ __launch_bounds__(256) void seqattn_fwd_batched_kernel(SeqAttnBatch b) {
    int i = 0;
    while (i + 1 < b.count && (int)blockIdx.x >= b.blk_begin[i + 1]) ++i;
    const grappa_seqattn_item& t = b.it[i];
    const int vb = (int)blockIdx.x - b.blk_begin[i], F = t.nheads * t.dh;
    switch (t.s) {
        case 1: seqattn_fwd_body<1, float>(t.T, F, t.dh, t.qkv, t.out, t.amax, vb); break;
        case 2: seqattn_fwd_body<2, float>(t.T, F, t.dh, t.qkv, t.out, t.amax, vb); break;
        case 3: seqattn_fwd_body<3, float>(t.T, F, t.dh, t.qkv, t.out, t.amax, vb); break;
        default: seqattn_fwd_body<4, float>(t.T, F, t.dh, t.qkv, t.out, t.amax, vb); break;
    }
}

// The attention output straight in the PAIR format (common.h st_pairs4): in inference nothing but the out-projection product reads it.
// The S x F outputs of a tuple stay in registers until their rows' maxima are known (F <= 512: two trips of 64 lanes x 4 columns).
template <int S>
__global__ __launch_bounds__(256) void seqattn_fwd_pairs_kernel(int T, int F, int dh, const float* __restrict__ qkv, uint16_t* __restrict__ pairs, int ldp,
                                                                unsigned* __restrict__ row_amax) {
    const int lane = threadIdx.x & 63;
    const int t = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (t >= T) return;
    const int nvec = F >> 2, lph = dh >> 2;
    const float scale = 1.0f / sqrtf((float)dh);
    unsigned am[S];
    float4 keep[2][S];
#pragma unroll
    for (int i = 0; i < S; ++i) am[i] = 0u;
#pragma unroll
    for (int trip = 0; trip < 2; ++trip) {
        const int c = trip * 64 + lane;
        const bool ok = c < nvec;
        float4 q[S], k[S], v[S];
#pragma unroll
        for (int i = 0; i < S; ++i) {
            const float* row = qkv + ((size_t)i * T + t) * 3 * F;
            q[i] = ok ? ld4(row, c) : make_float4(0.f, 0.f, 0.f, 0.f);
            k[i] = ok ? ld4(row, nvec + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            v[i] = ok ? ld4(row, 2 * nvec + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < S; ++i) {                       // the same expressions as seqattn_fwd_kernel: the same bits
            float sc[S], mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < S; ++j) {
                sc[j] = group_sum(dot4(q[i], k[j]), lph) * scale;
                mx = fmaxf(mx, sc[j]);
            }
            float den = 0.f;
#pragma unroll
            for (int j = 0; j < S; ++j) {
                sc[j] = expf(sc[j] - mx);
                den += sc[j];
            }
            const float inv = 1.0f / den;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < S; ++j) {
                const float p = sc[j] * inv;
                o.x += p * v[j].x; o.y += p * v[j].y; o.z += p * v[j].z; o.w += p * v[j].w;
            }
            keep[trip][i] = o;
            if (ok) am[i] = max(am[i], mag4(o));
        }
    }
#pragma unroll
    for (int i = 0; i < S; ++i) {
        const unsigned m = wave_umax(am[i]);
        const int shift = grappa_amax_shift(m);
        uint16_t* pr = pairs + ((size_t)i * T + t) * ldp;
#pragma unroll
        for (int trip = 0; trip < 2; ++trip) {
            const int c = trip * 64 + lane;
            st_pairs4_paired(pr, c, keep[trip][i], shift, c < nvec);
        }
        if (lane == 0) row_amax[(size_t)i * T + t] = m;
    }
}

template <int S, typename TE>
__device__ __forceinline__ void seqattn_bwd_body(int T, int F, int dh, const TE* __restrict__ qkv, const TE* __restrict__ dout,
                                                 TE* __restrict__ dqkv, unsigned* __restrict__ row_amax, int vblock) {
    const int lane = threadIdx.x & 63;
    const int t = (vblock * blockDim.x + threadIdx.x) >> 6;
    if (t >= T) return;
    const int nvec = F >> 2, lph = dh >> 2;
    const float scale = 1.0f / sqrtf((float)dh);
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned am[S];
#pragma unroll
    for (int i = 0; i < S; ++i) am[i] = 0u;
    for (int c0 = 0; c0 < nvec; c0 += 64) {
        const int c = c0 + lane;
        const bool ok = c < nvec;
        float4 q[S], k[S], v[S], go[S], dq[S], dk[S], dv[S];
#pragma unroll
        for (int i = 0; i < S; ++i) {
            const TE* row = qkv + ((size_t)i * T + t) * 3 * F;
            q[i] = ok ? ld4(row, c) : zero;
            k[i] = ok ? ld4(row, nvec + c) : zero;
            v[i] = ok ? ld4(row, 2 * nvec + c) : zero;
            go[i] = ok ? ld4(dout + ((size_t)i * T + t) * F, c) : zero;
            dq[i] = zero; dk[i] = zero; dv[i] = zero;
        }
#pragma unroll
        for (int i = 0; i < S; ++i) {
            float p[S], dp[S], mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < S; ++j) {
                p[j] = group_sum(dot4(q[i], k[j]), lph) * scale;
                mx = fmaxf(mx, p[j]);
            }
            float den = 0.f;
#pragma unroll
            for (int j = 0; j < S; ++j) {
                p[j] = expf(p[j] - mx);
                den += p[j];
            }
            const float inv = 1.0f / den;
            float dsum = 0.f;
#pragma unroll
            for (int j = 0; j < S; ++j) {
                p[j] *= inv;
                dp[j] = group_sum(dot4(go[i], v[j]), lph);
                dsum += p[j] * dp[j];
            }
#pragma unroll
            for (int j = 0; j < S; ++j) {
                const float ds = p[j] * (dp[j] - dsum) * scale;
                dq[i].x += ds * k[j].x; dq[i].y += ds * k[j].y; dq[i].z += ds * k[j].z; dq[i].w += ds * k[j].w;
                dk[j].x += ds * q[i].x; dk[j].y += ds * q[i].y; dk[j].z += ds * q[i].z; dk[j].w += ds * q[i].w;
                dv[j].x += p[j] * go[i].x; dv[j].y += p[j] * go[i].y; dv[j].z += p[j] * go[i].z; dv[j].w += p[j] * go[i].w;
            }
        }
        if (ok) {
#pragma unroll
            for (int i = 0; i < S; ++i) {
                TE* row = dqkv + ((size_t)i * T + t) * 3 * F;
                st4(row, c, dq[i]);
                st4(row, nvec + c, dk[i]);
                st4(row, 2 * nvec + c, dv[i]);
                am[i] = max(am[i], max(mag4(dq[i]), max(mag4(dk[i]), mag4(dv[i]))));
            }
        }
    }
    if (row_amax) {
#pragma unroll
        for (int i = 0; i < S; ++i) {
            const unsigned m = wave_umax(am[i]);
            if (lane == 0) row_amax[(size_t)i * T + t] = m;
        }
    }
}

template <int S, typename TE>
__global__ __launch_bounds__(256) void seqattn_bwd_kernel(int T, int F, int dh, const TE* __restrict__ qkv, const TE* __restrict__ dout,
                                                          TE* __restrict__ dqkv, unsigned* __restrict__ row_amax) {
    seqattn_bwd_body<S, TE>(T, F, dh, qkv, dout, dqkv, row_amax, blockIdx.x);
}

__global__ __launch_bounds__(256) void seqattn_bwd_batched_kernel(SeqAttnBatch b) {
    int i = 0;
    while (i + 1 < b.count && (int)blockIdx.x >= b.blk_begin[i + 1]) ++i;
    const grappa_seqattn_item& t = b.it[i];
    const int vb = (int)blockIdx.x - b.blk_begin[i], F = t.nheads * t.dh;
    switch (t.s) {
        case 1: seqattn_bwd_body<1, float>(t.T, F, t.dh, t.qkv, t.dout, t.dqkv, t.amax, vb); break;
        case 2: seqattn_bwd_body<2, float>(t.T, F, t.dh, t.qkv, t.dout, t.dqkv, t.amax, vb); break;
        case 3: seqattn_bwd_body<3, float>(t.T, F, t.dh, t.qkv, t.dout, t.dqkv, t.amax, vb); break;
        default: seqattn_bwd_body<4, float>(t.T, F, t.dh, t.qkv, t.dout, t.dqkv, t.amax, vb); break;
    }
}

// ---- the same with E elements per lane (E = 8: 16-byte accesses of bf16 rows, see Chunk<E> in common.h)
template <int S, int E, typename TE>
__global__ __launch_bounds__(256) void seqattn_fwd_kernel_e(int T, int F, int dh, const TE* __restrict__ qkv, TE* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int t = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (t >= T) return;
    const int nvec = F / E, lph = dh / E;
    const float scale = 1.0f / sqrtf((float)dh);
    for (int c0 = 0; c0 < nvec; c0 += 64) {
        const int c = c0 + lane;
        const bool ok = c < nvec;
        Chunk<E> q[S], k[S], v[S];
#pragma unroll
        for (int i = 0; i < S; ++i) {
            const TE* row = qkv + ((size_t)i * T + t) * 3 * F;
            q[i] = ok ? ldc<E, TE>(row, c) : chunk_zero<E>();
            k[i] = ok ? ldc<E, TE>(row, nvec + c) : chunk_zero<E>();
            v[i] = ok ? ldc<E, TE>(row, 2 * nvec + c) : chunk_zero<E>();
        }
#pragma unroll
        for (int i = 0; i < S; ++i) {
            float sc[S], mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < S; ++j) {
                sc[j] = group_sum(cdot<E>(q[i], k[j]), lph) * scale;
                mx = fmaxf(mx, sc[j]);
            }
            float den = 0.f;
#pragma unroll
            for (int j = 0; j < S; ++j) {
                sc[j] = expf(sc[j] - mx);
                den += sc[j];
            }
            const float inv = 1.0f / den;
            Chunk<E> o = chunk_zero<E>();
#pragma unroll
            for (int j = 0; j < S; ++j) {
                const float p = sc[j] * inv;
#pragma unroll
                for (int e = 0; e < E; ++e) o.v[e] += p * v[j].v[e];
            }
            if (ok) stc<E, TE>(out + ((size_t)i * T + t) * F, c, o);
        }
    }
}

template <int S, int E, typename TE>
__global__ __launch_bounds__(256) void seqattn_bwd_kernel_e(int T, int F, int dh, const TE* __restrict__ qkv, const TE* __restrict__ dout,
                                                            TE* __restrict__ dqkv) {
    const int lane = threadIdx.x & 63;
    const int t = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (t >= T) return;
    const int nvec = F / E, lph = dh / E;
    const float scale = 1.0f / sqrtf((float)dh);
    for (int c0 = 0; c0 < nvec; c0 += 64) {
        const int c = c0 + lane;
        const bool ok = c < nvec;
        Chunk<E> q[S], k[S], v[S], go[S], dq[S], dk[S], dv[S];
#pragma unroll
        for (int i = 0; i < S; ++i) {
            const TE* row = qkv + ((size_t)i * T + t) * 3 * F;
            q[i] = ok ? ldc<E, TE>(row, c) : chunk_zero<E>();
            k[i] = ok ? ldc<E, TE>(row, nvec + c) : chunk_zero<E>();
            v[i] = ok ? ldc<E, TE>(row, 2 * nvec + c) : chunk_zero<E>();
            go[i] = ok ? ldc<E, TE>(dout + ((size_t)i * T + t) * F, c) : chunk_zero<E>();
            dq[i] = chunk_zero<E>(); dk[i] = chunk_zero<E>(); dv[i] = chunk_zero<E>();
        }
#pragma unroll
        for (int i = 0; i < S; ++i) {
            float p[S], dp[S], mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < S; ++j) {
                p[j] = group_sum(cdot<E>(q[i], k[j]), lph) * scale;
                mx = fmaxf(mx, p[j]);
            }
            float den = 0.f;
#pragma unroll
            for (int j = 0; j < S; ++j) {
                p[j] = expf(p[j] - mx);
                den += p[j];
            }
            const float inv = 1.0f / den;
            float dsum = 0.f;
#pragma unroll
            for (int j = 0; j < S; ++j) {
                p[j] *= inv;
                dp[j] = group_sum(cdot<E>(go[i], v[j]), lph);
                dsum += p[j] * dp[j];
            }
#pragma unroll
            for (int j = 0; j < S; ++j) {
                const float ds = p[j] * (dp[j] - dsum) * scale;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    dq[i].v[e] += ds * k[j].v[e];
                    dk[j].v[e] += ds * q[i].v[e];
                    dv[j].v[e] += p[j] * go[i].v[e];
                }
            }
        }
        if (ok) {
#pragma unroll
            for (int i = 0; i < S; ++i) {
                TE* row = dqkv + ((size_t)i * T + t) * 3 * F;
                stc<E, TE>(row, c, dq[i]);
                stc<E, TE>(row, nvec + c, dk[i]);
                stc<E, TE>(row, 2 * nvec + c, dv[i]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ symmetriser input
template <typename TE>
__global__ __launch_bounds__(256) void perm_concat_fwd_kernel(int s, int T, int F, int P, Perms perms, const TE* __restrict__ x,
                                                              TE* __restrict__ z) {
    const int lane = threadIdx.x & 63;
    const int nvec = F >> 2;
    const long total = (long)P * T * s;
    const long wave0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((long)gridDim.x * blockDim.x) >> 6;
    for (long w = wave0; w < total; w += nw) {
        const int j = (int)(w % s);
        const long pt = w / s;
        const int t = (int)(pt % T), p = (int)(pt / T);
        const int src_pos = perms.p[p][j];
        const TE* src = x + ((size_t)src_pos * T + t) * F;
        TE* dst = z + ((size_t)p * T + t) * ((size_t)s * F) + (size_t)j * F;
        for (int c = lane; c < nvec; c += 64) st4(dst, c, ld4(src, c));
    }
}

template <typename TE>
__global__ __launch_bounds__(256) void perm_concat_bwd_kernel(int s, int T, int F, int P, Perms inv, const TE* __restrict__ dz,
                                                              TE* __restrict__ dx) {
    const int lane = threadIdx.x & 63;
    const int nvec = F >> 2;
    const long total = (long)s * T;
    const long wave0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((long)gridDim.x * blockDim.x) >> 6;
    for (long w = wave0; w < total; w += nw) {
        const int i = (int)(w / T), t = (int)(w % T);
        TE* dst = dx + (size_t)w * F;
        for (int c = lane; c < nvec; c += 64) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int p = 0; p < P; ++p) {
                const int j = inv.p[p][i];
                const float4 v = ld4(dz + ((size_t)p * T + t) * ((size_t)s * F) + (size_t)j * F, c);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            st4(dst, c, acc);
        }
    }
}

// ------------------------------------------------------------------------------------------------ output maps
__device__ inline float to_positive(float c, float mos, float std, float mn) { return std * (grappa_elu(mos + c - 1.0f) + 1.0f) + mn; }
__device__ inline float to_positive_grad(float c, float mos, float std) {
    const float z = mos + c - 1.0f;
    return std * (z > 0.f ? 1.0f : expf(z));
}

__global__ __launch_bounds__(256) void param_out_fwd_kernel(int kind, int T, int P, int n_per, int gated, float cutoff,
                                                            const float* __restrict__ o, int ldo, const float* __restrict__ cst,
                                                            float* __restrict__ k, float* __restrict__ eq) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    if (kind == GRAPPA_OUT_TORSION) {
        for (int n = 0; n < n_per; ++n) {
            float c = 0.f, g = 0.f;
            for (int p = 0; p < P; ++p) {
                const float* row = o + ((size_t)p * T + t) * ldo;
                c += row[n];
                if (gated) g += row[n_per + n];
            }
            float v = gated ? c * grappa_sigmoid(g) * cst[n] : c * cst[n] + cst[n_per + n];
            if (cutoff > 0.f) v = fabsf(v) > cutoff ? v : 0.f;
            k[(size_t)t * n_per + n] = v;
        }
        return;
    }
    float c0 = 0.f, c1 = 0.f;
    for (int p = 0; p < P; ++p) {
        const float* row = o + ((size_t)p * T + t) * ldo;
        c0 += row[0];
        c1 += row[1];
    }
    if (kind == GRAPPA_OUT_BOND) eq[t] = to_positive(c0, cst[0], cst[1], cst[2]);
    else eq[t] = cst[1] * grappa_sigmoid(cst[0] * c0);
    k[t] = to_positive(c1, cst[3], cst[4], cst[5]);
}

__global__ __launch_bounds__(256) void param_out_bwd_kernel(int kind, int T, int P, int n_per, int gated, float cutoff, int nout,
                                                            const float* __restrict__ o, int ldo, const float* __restrict__ cst,
                                                            const float* __restrict__ dk, const float* __restrict__ deq,
                                                            float* __restrict__ d_o) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    if (kind == GRAPPA_OUT_TORSION) {
        for (int n = 0; n < n_per; ++n) {
            float c = 0.f, g = 0.f;
            for (int p = 0; p < P; ++p) {
                const float* row = o + ((size_t)p * T + t) * ldo;
                c += row[n];
                if (gated) g += row[n_per + n];
            }
            const float sg = gated ? grappa_sigmoid(g) : 0.f;
            const float v = gated ? c * sg * cst[n] : c * cst[n] + cst[n_per + n];
            float up = dk ? dk[(size_t)t * n_per + n] : 0.f;
            if (cutoff > 0.f && !(fabsf(v) > cutoff)) up = 0.f;
            const float dc = gated ? up * sg * cst[n] : up * cst[n];
            const float dg = gated ? up * c * cst[n] * sg * (1.0f - sg) : 0.f;
            for (int p = 0; p < P; ++p) {
                float* row = d_o + ((size_t)p * T + t) * ldo;
                row[n] = dc;
                if (gated) row[n_per + n] = dg;
            }
        }
        return;
    }
    float c0 = 0.f, c1 = 0.f;
    for (int p = 0; p < P; ++p) {
        const float* row = o + ((size_t)p * T + t) * ldo;
        c0 += row[0];
        c1 += row[1];
    }
    const float ueq = deq ? deq[t] : 0.f, uk = dk ? dk[t] : 0.f;
    float d0;
    if (kind == GRAPPA_OUT_BOND) d0 = ueq * to_positive_grad(c0, cst[0], cst[1]);
    else {
        const float sg = grappa_sigmoid(cst[0] * c0);
        d0 = ueq * cst[1] * cst[0] * sg * (1.0f - sg);
    }
    const float d1 = uk * to_positive_grad(c1, cst[3], cst[4]);
    for (int p = 0; p < P; ++p) {
        float* row = d_o + ((size_t)p * T + t) * ldo;
        row[0] = d0;
        row[1] = d1;
        for (int j = 2; j < nout; ++j) row[j] = 0.f;   // the harmonic gate column has no effect (reference quirk Q3)
    }
}

// learnable_statistics=True (reference models/final_layer.py:21-44, :64-88, interaction_parameters.py:463-470): gradient of the
// loss with respect to the statistics `cst` of an output map.  Per block: every thread's terms, summed in a fixed tree; the per-block
// partials part[block][16] are added by one wavefront afterwards (param_out_stats_final_kernel): reproducible.
//   bond    cst = [mos_eq, std_eq, min_eq, mos_k, std_k, min_k]   eq = std (elu(mos + c - 1) + 1) + min
//   angle   cst = [std_over_max, max, 0, mos_k, std_k, min_k]     eq = max sigmoid(std_over_max c)
//   torsion cst = [k_std[n_per], k_mean[n_per]]                   k = c k_std + k_mean  |  gated: c sigmoid(g) k_std
constexpr int STATS_MAX = 16;
__global__ __launch_bounds__(256) void param_out_stats_kernel(int kind, int T, int P, int n_per, int gated, float cutoff, const float* __restrict__ o,
                                                              int ldo, const float* __restrict__ cst, const float* __restrict__ dk,
                                                              const float* __restrict__ deq, float* __restrict__ part) {
    __shared__ float red[4][STATS_MAX];
    float acc[STATS_MAX];
#pragma unroll
    for (int i = 0; i < STATS_MAX; ++i) acc[i] = 0.f;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < T; t += gridDim.x * 256) {
        if (kind == GRAPPA_OUT_TORSION) {
            for (int n = 0; n < n_per; ++n) {
                float c = 0.f, g = 0.f;
                for (int p = 0; p < P; ++p) {
                    const float* row = o + ((size_t)p * T + t) * ldo;
                    c += row[n];
                    if (gated) g += row[n_per + n];
                }
                const float sg = gated ? grappa_sigmoid(g) : 0.f;
                const float v = gated ? c * sg * cst[n] : c * cst[n] + cst[n_per + n];
                float up = dk ? dk[(size_t)t * n_per + n] : 0.f;
                if (cutoff > 0.f && !(fabsf(v) > cutoff)) up = 0.f;
                acc[n] += gated ? up * c * sg : up * c;
                if (!gated) acc[n_per + n] += up;
            }
            continue;
        }
        float c0 = 0.f, c1 = 0.f;
        for (int p = 0; p < P; ++p) {
            const float* row = o + ((size_t)p * T + t) * ldo;
            c0 += row[0];
            c1 += row[1];
        }
        const float ueq = deq ? deq[t] : 0.f, uk = dk ? dk[t] : 0.f;
        if (kind == GRAPPA_OUT_BOND) {
            acc[0] += ueq * to_positive_grad(c0, cst[0], cst[1]);
            acc[1] += ueq * (grappa_elu(cst[0] + c0 - 1.0f) + 1.0f);
            acc[2] += ueq;
        } else {
            const float sg = grappa_sigmoid(cst[0] * c0);
            acc[0] += ueq * cst[1] * sg * (1.0f - sg) * c0;
            acc[1] += ueq * sg;
        }
        acc[3] += uk * to_positive_grad(c1, cst[3], cst[4]);
        acc[4] += uk * (grappa_elu(cst[3] + c1 - 1.0f) + 1.0f);
        acc[5] += uk;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < STATS_MAX; ++i) {
        const float v = wave_sum(acc[i]);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < STATS_MAX) part[(size_t)blockIdx.x * STATS_MAX + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ void param_out_stats_final_kernel(int nblocks, int ncst, const float* __restrict__ part, float* __restrict__ d_cst) {
    const int i = threadIdx.x;
    if (i >= ncst) return;
    float s = 0.f;
    for (int b = 0; b < nblocks; ++b) s += part[(size_t)b * STATS_MAX + i];
    d_cst[i] = s;
}

inline int wave_grid(long nwaves, int cap = 8192) {
    long b = (nwaves + 3) / 4;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}
inline bool pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

namespace {
template <typename T> inline bool aligned_el(const void* p) { return (reinterpret_cast<uintptr_t>(p) & (sizeof(T) == 4 ? 15 : 7)) == 0; }

template <typename TA, typename TX>
int tuple_gather_fwd_impl(void* stream, int T, int s, int W, const TA* a, int lda, const int* idx, const float* pe, TX* x, int ldx) {
    if (T < 0 || s < 1 || s > 4 || W <= 0 || (W & 3) || (lda & 3) || (ldx & 3)) return GRAPPA_ERR_ARG;
    if (T == 0) return GRAPPA_OK;
    if (!a || !idx || !x || !aligned_el<TA>(a) || !aligned_el<TX>(x)) return GRAPPA_ERR_ARG;
    GRAPPA_LAUNCH((tuple_gather_fwd_kernel<TA, TX>), dim3(wave_grid((long)s * T)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), T, s, W, a,
                       lda, idx, pe, x, ldx);
    return grappa_launch_status();
}

template <typename TX, typename TA>
int tuple_gather_bwd_impl(void* stream, int N, int W, const int* inv_ptr, const int* inv_rows, const TX* dx, int lddx, TA* da, int ldda, int has_pe,
                          int accumulate) {
    if (N < 0 || W <= 0 || (W & 3) || (lddx & 3) || (ldda & 3)) return GRAPPA_ERR_ARG;
    if (N == 0) return GRAPPA_OK;
    if (!inv_ptr || !da || !aligned_el<TA>(da) || (dx && !aligned_el<TX>(dx))) return GRAPPA_ERR_ARG;
    GRAPPA_LAUNCH((tuple_gather_bwd_kernel<TX, TA>), dim3((N + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), N, W, inv_ptr,
                       inv_rows, dx, lddx, da, ldda, has_pe, accumulate);
    return grappa_launch_status();
}

template <typename TE>
int seqattn_fwd_impl(void* stream, int s, int T, int nheads, int dh, const TE* qkv, TE* out, unsigned* row_amax = nullptr) {
    if (s < 1 || s > 4 || T < 0 || nheads <= 0 || dh <= 0 || (dh & 3) || !pow2(dh / 4) || dh / 4 > 64 || nheads * dh > 1024) return GRAPPA_ERR_ARG;
    if (T == 0) return GRAPPA_OK;
    if (!qkv || !out || !aligned_el<TE>(qkv) || !aligned_el<TE>(out)) return GRAPPA_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((T + 3) / 4);
    const int F = nheads * dh;
    if constexpr (sizeof(TE) == 2) {
        if ((dh & 7) == 0 && pow2(dh / 8) && ((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {      // 8 bf16 per lane
            switch (s) {
                case 1: GRAPPA_LAUNCH((seqattn_fwd_kernel_e<1, 8, TE>), grid, dim3(256), 0, st, T, F, dh, qkv, out); break;
                case 2: GRAPPA_LAUNCH((seqattn_fwd_kernel_e<2, 8, TE>), grid, dim3(256), 0, st, T, F, dh, qkv, out); break;
                case 3: GRAPPA_LAUNCH((seqattn_fwd_kernel_e<3, 8, TE>), grid, dim3(256), 0, st, T, F, dh, qkv, out); break;
                default: GRAPPA_LAUNCH((seqattn_fwd_kernel_e<4, 8, TE>), grid, dim3(256), 0, st, T, F, dh, qkv, out); break;
            }
            return grappa_launch_status();
        }
    }
    switch (s) {
        case 1: GRAPPA_LAUNCH((seqattn_fwd_kernel<1, TE>), grid, dim3(256), 0, st, T, F, dh, qkv, out, row_amax); break;
        case 2: GRAPPA_LAUNCH((seqattn_fwd_kernel<2, TE>), grid, dim3(256), 0, st, T, F, dh, qkv, out, row_amax); break;
        case 3: GRAPPA_LAUNCH((seqattn_fwd_kernel<3, TE>), grid, dim3(256), 0, st, T, F, dh, qkv, out, row_amax); break;
        default: GRAPPA_LAUNCH((seqattn_fwd_kernel<4, TE>), grid, dim3(256), 0, st, T, F, dh, qkv, out, row_amax); break;
    }
    return grappa_launch_status();
}

template <typename TE>
int seqattn_bwd_impl(void* stream, int s, int T, int nheads, int dh, const TE* qkv, const TE* dout, TE* dqkv, unsigned* row_amax = nullptr) {
    if (s < 1 || s > 4 || T < 0 || nheads <= 0 || dh <= 0 || (dh & 3) || !pow2(dh / 4) || dh / 4 > 64 || nheads * dh > 1024) return GRAPPA_ERR_ARG;
    if (T == 0) return GRAPPA_OK;
    if (!qkv || !dout || !dqkv || !aligned_el<TE>(qkv) || !aligned_el<TE>(dout) || !aligned_el<TE>(dqkv)) return GRAPPA_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((T + 3) / 4);
    const int F = nheads * dh;
    if constexpr (sizeof(TE) == 2) {
        if ((dh & 7) == 0 && pow2(dh / 8) &&
            ((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(dqkv)) & 15) == 0) {
            switch (s) {
                case 1: GRAPPA_LAUNCH((seqattn_bwd_kernel_e<1, 8, TE>), grid, dim3(256), 0, st, T, F, dh, qkv, dout, dqkv); break;
                case 2: GRAPPA_LAUNCH((seqattn_bwd_kernel_e<2, 8, TE>), grid, dim3(256), 0, st, T, F, dh, qkv, dout, dqkv); break;
                case 3: GRAPPA_LAUNCH((seqattn_bwd_kernel_e<3, 8, TE>), grid, dim3(256), 0, st, T, F, dh, qkv, dout, dqkv); break;
                default: GRAPPA_LAUNCH((seqattn_bwd_kernel_e<4, 8, TE>), grid, dim3(256), 0, st, T, F, dh, qkv, dout, dqkv); break;
            }
            return grappa_launch_status();
        }
    }
    switch (s) {
        case 1: GRAPPA_LAUNCH((seqattn_bwd_kernel<1, TE>), grid, dim3(256), 0, st, T, F, dh, qkv, dout, dqkv, row_amax); break;
        case 2: GRAPPA_LAUNCH((seqattn_bwd_kernel<2, TE>), grid, dim3(256), 0, st, T, F, dh, qkv, dout, dqkv, row_amax); break;
        case 3: GRAPPA_LAUNCH((seqattn_bwd_kernel<3, TE>), grid, dim3(256), 0, st, T, F, dh, qkv, dout, dqkv, row_amax); break;
        default: GRAPPA_LAUNCH((seqattn_bwd_kernel<4, TE>), grid, dim3(256), 0, st, T, F, dh, qkv, dout, dqkv, row_amax); break;
    }
    return grappa_launch_status();
}
}  // namespace

extern "C" int grappa_tuple_gather_fwd_f32(void* stream, int T, int s, int W, const float* a, int lda, const int* idx, const float* pe,
                                           float* x, int ldx) {
    return tuple_gather_fwd_impl<float, float>(stream, T, s, W, a, lda, idx, pe, x, ldx);
}
extern "C" int grappa_tuple_gather_fwd_bf16(void* stream, int T, int s, int W, const uint16_t* a, int lda, const int* idx, const float* pe,
                                            uint16_t* x, int ldx) {
    return tuple_gather_fwd_impl<grappa_bf16_t, grappa_bf16_t>(stream, T, s, W, a, lda, idx, pe, x, ldx);
}
extern "C" int grappa_tuple_gather_bwd_f32(void* stream, int N, int W, const int* inv_ptr, const int* inv_rows, const float* dx, int lddx,
                                           float* da, int ldda, int has_pe, int accumulate) {
    return tuple_gather_bwd_impl<float, float>(stream, N, W, inv_ptr, inv_rows, dx, lddx, da, ldda, has_pe, accumulate);
}
extern "C" int grappa_tuple_gather_bwd_bf16(void* stream, int N, int W, const int* inv_ptr, const int* inv_rows, const uint16_t* dx, int lddx,
                                            uint16_t* da, int ldda, int has_pe, int accumulate) {
    return tuple_gather_bwd_impl<grappa_bf16_t, grappa_bf16_t>(stream, N, W, inv_ptr, inv_rows, dx, lddx, da, ldda, has_pe, accumulate);
}
namespace {
int seqattn_batch(const grappa_seqattn_item* items, int count, bool bwd, SeqAttnBatch& b) {
    if (!items || count <= 0 || count > GRAPPA_ROW_BATCH_MAX) return GRAPPA_ERR_ARG;
    b.count = count;
    b.blk_begin[0] = 0;
    for (int i = 0; i < count; ++i) {
        const grappa_seqattn_item& t = items[i];
        if (t.s < 1 || t.s > 4 || t.T < 0 || t.nheads <= 0 || t.dh <= 0 || (t.dh & 3) || !pow2(t.dh / 4) || t.dh / 4 > 64 || t.nheads * t.dh > 1024) return GRAPPA_ERR_ARG;
        if (t.T > 0) {
            if (!t.qkv || !aligned_el<float>(t.qkv)) return GRAPPA_ERR_ARG;
            if (bwd ? (!t.dout || !t.dqkv || !aligned_el<float>(t.dout) || !aligned_el<float>(t.dqkv)) : (!t.out || !aligned_el<float>(t.out))) return GRAPPA_ERR_ARG;
        }
        b.it[i] = t;
        b.blk_begin[i + 1] = b.blk_begin[i] + (t.T + 3) / 4;
    }
    for (int i = count; i < GRAPPA_ROW_BATCH_MAX; ++i) b.blk_begin[i + 1] = b.blk_begin[count];
    return GRAPPA_OK;
}
}  // namespace

extern "C" int grappa_seqattn_fwd_batched_f32(void* stream, const grappa_seqattn_item* items, int count) {
    SeqAttnBatch b;
    const int rc = seqattn_batch(items, count, false, b);
    if (rc != GRAPPA_OK || b.blk_begin[count] == 0) return rc;
    GRAPPA_LAUNCH(seqattn_fwd_batched_kernel, dim3(b.blk_begin[count]), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), b);
    return grappa_launch_status();
}
extern "C" int grappa_seqattn_bwd_batched_f32(void* stream, const grappa_seqattn_item* items, int count) {
    SeqAttnBatch b;
    const int rc = seqattn_batch(items, count, true, b);
    if (rc != GRAPPA_OK || b.blk_begin[count] == 0) return rc;
    GRAPPA_LAUNCH(seqattn_bwd_batched_kernel, dim3(b.blk_begin[count]), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), b);
    return grappa_launch_status();
}

extern "C" int grappa_seqattn_fwd_f32(void* stream, int s, int T, int nheads, int dh, const float* qkv, float* out) {
    return seqattn_fwd_impl<float>(stream, s, T, nheads, dh, qkv, out);
}
extern "C" int grappa_seqattn_fwd_amax_f32(void* stream, int s, int T, int nheads, int dh, const float* qkv, float* out, uint32_t* out_amax) {
    return seqattn_fwd_impl<float>(stream, s, T, nheads, dh, qkv, out, out_amax);
}
extern "C" int grappa_seqattn_bwd_amax_f32(void* stream, int s, int T, int nheads, int dh, const float* qkv, const float* dout, float* dqkv,
                                           uint32_t* dqkv_amax) {
    return seqattn_bwd_impl<float>(stream, s, T, nheads, dh, qkv, dout, dqkv, dqkv_amax);
}
extern "C" int grappa_seqattn_fwd_pairs_f32(void* stream, int s, int T, int nheads, int dh, const float* qkv, uint16_t* pairs, int ldp,
                                            uint32_t* out_amax) {
    const int F = nheads * dh;
    if (s < 1 || s > 4 || T < 0 || nheads <= 0 || dh <= 0 || (dh & 3) || !pow2(dh / 4) || dh / 4 > 64 || F > 512 || (F & 31) || ldp < 2 * F || (ldp & 7))
        return GRAPPA_ERR_ARG;
    if (T == 0) return GRAPPA_OK;
    if (!qkv || !pairs || !out_amax || !aligned_el<float>(qkv) || (reinterpret_cast<uintptr_t>(pairs) & 15)) return GRAPPA_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((T + 3) / 4);
    switch (s) {
        case 1: GRAPPA_LAUNCH((seqattn_fwd_pairs_kernel<1>), grid, dim3(256), 0, st, T, F, dh, qkv, pairs, ldp, out_amax); break;
        case 2: GRAPPA_LAUNCH((seqattn_fwd_pairs_kernel<2>), grid, dim3(256), 0, st, T, F, dh, qkv, pairs, ldp, out_amax); break;
        case 3: GRAPPA_LAUNCH((seqattn_fwd_pairs_kernel<3>), grid, dim3(256), 0, st, T, F, dh, qkv, pairs, ldp, out_amax); break;
        default: GRAPPA_LAUNCH((seqattn_fwd_pairs_kernel<4>), grid, dim3(256), 0, st, T, F, dh, qkv, pairs, ldp, out_amax); break;
    }
    return grappa_launch_status();
}
extern "C" int grappa_seqattn_fwd_bf16(void* stream, int s, int T, int nheads, int dh, const uint16_t* qkv, uint16_t* out) {
    return seqattn_fwd_impl<grappa_bf16_t>(stream, s, T, nheads, dh, qkv, out);
}
extern "C" int grappa_seqattn_bwd_f32(void* stream, int s, int T, int nheads, int dh, const float* qkv, const float* dout, float* dqkv) {
    return seqattn_bwd_impl<float>(stream, s, T, nheads, dh, qkv, dout, dqkv);
}
extern "C" int grappa_seqattn_bwd_bf16(void* stream, int s, int T, int nheads, int dh, const uint16_t* qkv, const uint16_t* dout, uint16_t* dqkv) {
    return seqattn_bwd_impl<grappa_bf16_t>(stream, s, T, nheads, dh, qkv, dout, dqkv);
}

static int fill_perms(int s, int P, const int* h_perm, Perms& fwd, Perms& inv) {
    if (s < 1 || s > 4 || P < 1 || P > 6 || !h_perm) return GRAPPA_ERR_ARG;
    for (int p = 0; p < P; ++p) {
        int seen = 0;
        for (int j = 0; j < s; ++j) {
            const int v = h_perm[p * s + j];
            if (v < 0 || v >= s) return GRAPPA_ERR_ARG;
            fwd.p[p][j] = v;
            inv.p[p][v] = j;
            seen |= 1 << v;
        }
        if (seen != (1 << s) - 1) return GRAPPA_ERR_ARG;
    }
    return GRAPPA_OK;
}

namespace {
template <typename TE>
int perm_concat_fwd_impl(void* stream, int s, int T, int F, int P, const int* h_perm, const TE* x, TE* z) {
    Perms fwd, inv;
    if (int rc = fill_perms(s, P, h_perm, fwd, inv)) return rc;
    if (T < 0 || F <= 0 || (F & 3)) return GRAPPA_ERR_ARG;
    if (T == 0) return GRAPPA_OK;
    if (!x || !z || !aligned_el<TE>(x) || !aligned_el<TE>(z)) return GRAPPA_ERR_ARG;
    GRAPPA_LAUNCH(perm_concat_fwd_kernel<TE>, dim3(wave_grid((long)P * T * s)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), s, T, F, P,
                       fwd, x, z);
    return grappa_launch_status();
}
template <typename TE>
int perm_concat_bwd_impl(void* stream, int s, int T, int F, int P, const int* h_perm, const TE* dz, TE* dx) {
    Perms fwd, inv;
    if (int rc = fill_perms(s, P, h_perm, fwd, inv)) return rc;
    if (T < 0 || F <= 0 || (F & 3)) return GRAPPA_ERR_ARG;
    if (T == 0) return GRAPPA_OK;
    if (!dz || !dx || !aligned_el<TE>(dz) || !aligned_el<TE>(dx)) return GRAPPA_ERR_ARG;
    GRAPPA_LAUNCH(perm_concat_bwd_kernel<TE>, dim3(wave_grid((long)T * s)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), s, T, F, P, inv,
                       dz, dx);
    return grappa_launch_status();
}
}  // namespace

extern "C" int grappa_perm_concat_fwd_f32(void* stream, int s, int T, int F, int P, const int* h_perm, const float* x, float* z) {
    return perm_concat_fwd_impl<float>(stream, s, T, F, P, h_perm, x, z);
}
extern "C" int grappa_perm_concat_fwd_bf16(void* stream, int s, int T, int F, int P, const int* h_perm, const uint16_t* x, uint16_t* z) {
    return perm_concat_fwd_impl<grappa_bf16_t>(stream, s, T, F, P, h_perm, x, z);
}
extern "C" int grappa_perm_concat_bwd_f32(void* stream, int s, int T, int F, int P, const int* h_perm, const float* dz, float* dx) {
    return perm_concat_bwd_impl<float>(stream, s, T, F, P, h_perm, dz, dx);
}
extern "C" int grappa_perm_concat_bwd_bf16(void* stream, int s, int T, int F, int P, const int* h_perm, const uint16_t* dz, uint16_t* dx) {
    return perm_concat_bwd_impl<grappa_bf16_t>(stream, s, T, F, P, h_perm, dz, dx);
}

extern "C" int grappa_param_out_fwd_f32(void* stream, int kind, int T, int P, int n_per, int gated, float cutoff, const float* o, int ldo,
                                        const float* consts, float* k, float* eq) {
    if (T < 0 || P < 1 || kind < 0 || kind > 2) return GRAPPA_ERR_ARG;
    if (kind == GRAPPA_OUT_TORSION && (n_per < 1 || ldo < (gated ? 2 : 1) * n_per)) return GRAPPA_ERR_ARG;
    if (kind != GRAPPA_OUT_TORSION && ldo < 2) return GRAPPA_ERR_ARG;
    if (T == 0) return GRAPPA_OK;
    if (!o || !consts || !k || (kind != GRAPPA_OUT_TORSION && !eq)) return GRAPPA_ERR_ARG;
    GRAPPA_LAUNCH(param_out_fwd_kernel, dim3((T + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), kind, T, P, n_per, gated,
                       cutoff, o, ldo, consts, k, eq);
    return grappa_launch_status();
}

extern "C" size_t grappa_param_out_stats_workspace_bytes(int T) {
    const int blocks = (T + 255) / 256 > 256 ? 256 : (T + 255) / 256;
    return (size_t)(blocks < 1 ? 1 : blocks) * STATS_MAX * sizeof(float);
}
extern "C" int grappa_param_out_bwd_stats_f32(void* stream, int kind, int T, int P, int n_per, int gated, float cutoff, const float* o, int ldo,
                                              const float* consts, const float* dk, const float* deq, float* d_consts, void* ws, size_t ws_bytes) {
    const int ncst = kind == GRAPPA_OUT_TORSION ? 2 * n_per : 6;
    if (kind < 0 || kind > 2 || T < 0 || P < 1 || ncst > STATS_MAX || (kind == GRAPPA_OUT_TORSION && n_per < 1)) return GRAPPA_ERR_ARG;
    if (!d_consts) return GRAPPA_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (T == 0) return hipMemsetAsync(d_consts, 0, ncst * sizeof(float), st) == hipSuccess ? GRAPPA_OK : GRAPPA_ERR_LAUNCH;
    if (!o || !consts) return GRAPPA_ERR_ARG;
    const int blocks = (T + 255) / 256 > 256 ? 256 : (T + 255) / 256;
    if (!ws || ws_bytes < (size_t)blocks * STATS_MAX * sizeof(float)) return GRAPPA_ERR_WORKSPACE;
    float* part = reinterpret_cast<float*>(ws);
    GRAPPA_LAUNCH(param_out_stats_kernel, dim3(blocks), dim3(256), 0, st, kind, T, P, n_per, gated, cutoff, o, ldo, consts, dk, deq, part);
    GRAPPA_LAUNCH(param_out_stats_final_kernel, dim3(1), dim3(64), 0, st, blocks, ncst, part, d_consts);
    return grappa_launch_status();
}
extern "C" int grappa_param_out_bwd_f32(void* stream, int kind, int T, int P, int n_per, int gated, float cutoff, const float* o, int ldo,
                                        const float* consts, const float* dk, const float* deq, float* d_o) {
    if (T < 0 || P < 1 || kind < 0 || kind > 2) return GRAPPA_ERR_ARG;
    if (kind == GRAPPA_OUT_TORSION && (n_per < 1 || ldo < (gated ? 2 : 1) * n_per)) return GRAPPA_ERR_ARG;
    if (kind != GRAPPA_OUT_TORSION && ldo < 2) return GRAPPA_ERR_ARG;
    if (T == 0) return GRAPPA_OK;
    if (!o || !consts || !d_o) return GRAPPA_ERR_ARG;
    GRAPPA_LAUNCH(param_out_bwd_kernel, dim3((T + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), kind, T, P, n_per, gated,
                       cutoff, ldo, o, ldo, consts, dk, deq, d_o);
    return grappa_launch_status();
}
