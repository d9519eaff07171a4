// Graph message passing on the bonded-neighbour CSR (destination-major).  HBM-bound gathers:
// one wavefront owns one destination node; a feature row (F = H*D floats, 2 KB for the production
// model) is read as coalesced 16-byte chunks, chunk c = lane + 64*i; a head spans D/4 adjacent lanes,
// so per-head dot products are xor-shuffle reductions inside aligned lane groups.  Degree is 1..6 for
// molecules: softmax runs online in registers, the next neighbour row is prefetched while the current
// one is reduced.  The backward pass gathers through the reverse-edge index (symmetric graph): no atomics.
#include "common.h"

namespace {

constexpr int MAXC = 8;   // float4 chunks per lane: F <= 2048

struct Row {
    float4 v[MAXC];
};

template <int NC, typename T>
__device__ inline void load_row(const T* __restrict__ base, int node, int F, int lane, int nvec, float4 (&r)[NC]) {
    const T* p = base + (size_t)node * F;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + 64 * i;
        r[i] = (c < nvec) ? ld4(p, c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

__device__ inline float dot4(const float4& a, const float4& b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }

// ---------------------------------------------------------------------------------------------
// forward: NC = chunks per lane (compile-time: 1, 2, 4, 8); lanes_per_head = D/4
template <int NC, typename T>
__global__ __launch_bounds__(256) void gat_fwd_kernel(int N, int H, int D, const int* __restrict__ indptr, const int* __restrict__ indices,
                                                      const T* __restrict__ ft, T* __restrict__ out, float* __restrict__ alpha) {
    const int lane = threadIdx.x & 63;
    const int v = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (v >= N) return;
    const int F = H * D, nvec = F >> 2, lph = D >> 2;
    const float inv_sqrt_d = 1.0f / sqrtf((float)D);
    float4 fv[NC], cur[NC], nxt[NC], acc[NC];
    float mx[NC], den[NC];
    load_row<NC>(ft, v, F, lane, nvec, fv);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        mx[i] = -INFINITY;
        den[i] = 0.f;
    }
    const int e0 = indptr[v], e1 = indptr[v + 1];
    if (e0 < e1) load_row<NC>(ft, indices[e0], F, lane, nvec, nxt);
    for (int e = e0; e < e1; ++e) {
#pragma unroll
        for (int i = 0; i < NC; ++i) cur[i] = nxt[i];
        if (e + 1 < e1) load_row<NC>(ft, indices[e + 1], F, lane, nvec, nxt);
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const float s = group_sum(dot4(cur[i], fv[i]), lph) * inv_sqrt_d;
            const int c = lane + 64 * i;
            if (c < nvec && (lane % lph) == 0) alpha[(size_t)e * H + (c / lph)] = s;   // raw score, normalised below
            const float m_new = fmaxf(mx[i], s);
            const float corr = expf(mx[i] - m_new);     // exp(-inf) = 0 on the first edge
            const float w = expf(s - m_new);
            den[i] = den[i] * corr + w;
            acc[i].x = acc[i].x * corr + w * cur[i].x;
            acc[i].y = acc[i].y * corr + w * cur[i].y;
            acc[i].z = acc[i].z * corr + w * cur[i].z;
            acc[i].w = acc[i].w * corr + w * cur[i].w;
            mx[i] = m_new;
        }
    }
    T* o = out + (size_t)v * F;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + 64 * i;
        if (c < nvec) {
            const float inv = den[i] > 0.f ? 1.0f / den[i] : 0.f;
            st4(o, c, make_float4(acc[i].x * inv, acc[i].y * inv, acc[i].z * inv, acc[i].w * inv));
            if ((lane % lph) == 0) {
                const int h = c / lph;
                for (int e = e0; e < e1; ++e) {
                    const size_t a = (size_t)e * H + h;
                    alpha[a] = expf(alpha[a] - mx[i]) * inv;
                }
            }
        }
    }
}

// delta[v,h] = <dout_v[h], out_v[h]>
template <int NC, typename T>
__global__ __launch_bounds__(256) void gat_delta_kernel(int N, int H, int D, const T* __restrict__ out, const T* __restrict__ dout,
                                                        float* __restrict__ delta) {
    const int lane = threadIdx.x & 63;
    const int v = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (v >= N) return;
    const int F = H * D, nvec = F >> 2, lph = D >> 2;
    float4 a[NC], b[NC];
    load_row<NC>(out, v, F, lane, nvec, a);
    load_row<NC>(dout, v, F, lane, nvec, b);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const float s = group_sum(dot4(a[i], b[i]), lph);
        const int c = lane + 64 * i;
        if (c < nvec && (lane % lph) == 0) delta[(size_t)v * H + c / lph] = s;
    }
}

// dft_u = sum_{e in in(u), w = indices[e]}  alpha[rev e] * dout_w
//                                          + ( ds(u->w) + ds(w->u) ) * ft_w / sqrt(D)
//   ds(u->w)[h] = alpha[rev e][h] * (<dout_w[h], ft_u[h]> - delta[w,h])      (u is the source of edge u->w)
//   ds(w->u)[h] = alpha[e][h]     * (<dout_u[h], ft_w[h]> - delta[u,h])      (u is the destination of edge w->u)
template <int NC, typename T>
__global__ __launch_bounds__(256) void gat_bwd_kernel(int N, int H, int D, const int* __restrict__ indptr, const int* __restrict__ indices,
                                                      const int* __restrict__ rev, const T* __restrict__ ft,
                                                      const float* __restrict__ alpha, const T* __restrict__ dout,
                                                      const float* __restrict__ delta, T* __restrict__ dft) {
    const int lane = threadIdx.x & 63;
    const int u = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (u >= N) return;
    const int F = H * D, nvec = F >> 2, lph = D >> 2;
    const float inv_sqrt_d = 1.0f / sqrtf((float)D);
    float4 fu[NC], du[NC], fw[NC], dw[NC], acc[NC];
    float delta_u[NC];
    load_row<NC>(ft, u, F, lane, nvec, fu);
    load_row<NC>(dout, u, F, lane, nvec, du);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int c = lane + 64 * i;
        delta_u[i] = (c < nvec) ? delta[(size_t)u * H + c / lph] : 0.f;
    }
    const int e0 = indptr[u], e1 = indptr[u + 1];
    for (int e = e0; e < e1; ++e) {
        const int w = indices[e];
        const int er = rev[e];
        load_row<NC>(ft, w, F, lane, nvec, fw);
        load_row<NC>(dout, w, F, lane, nvec, dw);
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + 64 * i;
            const int h = (c < nvec) ? c / lph : 0;
            const float a_uw = (c < nvec) ? alpha[(size_t)er * H + h] : 0.f;   // weight of u in w's softmax
            const float a_wu = (c < nvec) ? alpha[(size_t)e * H + h] : 0.f;    // weight of w in u's softmax
            const float d_w = (c < nvec) ? delta[(size_t)w * H + h] : 0.f;
            const float dot_src = group_sum(dot4(dw[i], fu[i]), lph);          // <dout_w, ft_u>
            const float dot_dst = group_sum(dot4(du[i], fw[i]), lph);          // <dout_u, ft_w>
            const float coef = (a_uw * (dot_src - d_w) + a_wu * (dot_dst - delta_u[i])) * inv_sqrt_d;
            acc[i].x += a_uw * dw[i].x + coef * fw[i].x;
            acc[i].y += a_uw * dw[i].y + coef * fw[i].y;
            acc[i].z += a_uw * dw[i].z + coef * fw[i].z;
            acc[i].w += a_uw * dw[i].w + coef * fw[i].w;
        }
    }
    T* o = dft + (size_t)u * F;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + 64 * i;
        if (c < nvec) st4(o, c, acc[i]);
    }
}


// ---------------------------------------------------------------------------------------------
// bf16 rows with EIGHT elements (16 bytes) per lane: the same algorithms as above on Chunk<8>.  With 4 elements per lane a bf16 row
// costs as many memory instructions as an fp32 row and the kernel runs no faster on half the bytes; with 8 it needs half of them.
template <int NC, int E, typename T>
__device__ inline void load_row_e(const T* __restrict__ base, int node, int F, int lane, int nvec, Chunk<E> (&r)[NC]) {
    const T* p = base + (size_t)node * F;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + 64 * i;
        r[i] = (c < nvec) ? ldc<E, T>(p, c) : chunk_zero<E>();
    }
}

template <int NC, int E, typename T>
__global__ __launch_bounds__(256) void gat_fwd_kernel_e(int N, int H, int D, const int* __restrict__ indptr, const int* __restrict__ indices,
                                                        const T* __restrict__ ft, T* __restrict__ out, float* __restrict__ alpha) {
    const int lane = threadIdx.x & 63;
    const int v = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (v >= N) return;
    const int F = H * D, nvec = F / E, lph = D / E;
    const float inv_sqrt_d = 1.0f / sqrtf((float)D);
    Chunk<E> fv[NC], cur[NC], nxt[NC], acc[NC];
    float mx[NC], den[NC];
    load_row_e<NC, E>(ft, v, F, lane, nvec, fv);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        acc[i] = chunk_zero<E>();
        mx[i] = -INFINITY;
        den[i] = 0.f;
    }
    const int e0 = indptr[v], e1 = indptr[v + 1];
    if (e0 < e1) load_row_e<NC, E>(ft, indices[e0], F, lane, nvec, nxt);
    for (int e = e0; e < e1; ++e) {
#pragma unroll
        for (int i = 0; i < NC; ++i) cur[i] = nxt[i];
        if (e + 1 < e1) load_row_e<NC, E>(ft, indices[e + 1], F, lane, nvec, nxt);
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const float s = group_sum(cdot<E>(cur[i], fv[i]), lph) * inv_sqrt_d;
            const int c = lane + 64 * i;
            if (c < nvec && (lane % lph) == 0) alpha[(size_t)e * H + (c / lph)] = s;   // raw score, normalised below
            const float m_new = fmaxf(mx[i], s);
            const float corr = expf(mx[i] - m_new);
            const float w = expf(s - m_new);
            den[i] = den[i] * corr + w;
#pragma unroll
            for (int q = 0; q < E; ++q) acc[i].v[q] = acc[i].v[q] * corr + w * cur[i].v[q];
            mx[i] = m_new;
        }
    }
    T* o = out + (size_t)v * F;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + 64 * i;
        if (c < nvec) {
            const float inv = den[i] > 0.f ? 1.0f / den[i] : 0.f;
            Chunk<E> r;
#pragma unroll
            for (int q = 0; q < E; ++q) r.v[q] = acc[i].v[q] * inv;
            stc<E, T>(o, c, r);
            if ((lane % lph) == 0) {
                const int h = c / lph;
                for (int e = e0; e < e1; ++e) {
                    const size_t a = (size_t)e * H + h;
                    alpha[a] = expf(alpha[a] - mx[i]) * inv;
                }
            }
        }
    }
}

template <int NC, int E, typename T>
__global__ __launch_bounds__(256) void gat_delta_kernel_e(int N, int H, int D, const T* __restrict__ out, const T* __restrict__ dout,
                                                          float* __restrict__ delta) {
    const int lane = threadIdx.x & 63;
    const int v = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (v >= N) return;
    const int F = H * D, nvec = F / E, lph = D / E;
    Chunk<E> a[NC], b[NC];
    load_row_e<NC, E>(out, v, F, lane, nvec, a);
    load_row_e<NC, E>(dout, v, F, lane, nvec, b);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const float s = group_sum(cdot<E>(a[i], b[i]), lph);
        const int c = lane + 64 * i;
        if (c < nvec && (lane % lph) == 0) delta[(size_t)v * H + c / lph] = s;
    }
}

template <int NC, int E, typename T>
__global__ __launch_bounds__(256) void gat_bwd_kernel_e(int N, int H, int D, const int* __restrict__ indptr, const int* __restrict__ indices,
                                                        const int* __restrict__ rev, const T* __restrict__ ft,
                                                        const float* __restrict__ alpha, const T* __restrict__ dout,
                                                        const float* __restrict__ delta, T* __restrict__ dft) {
    const int lane = threadIdx.x & 63;
    const int u = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (u >= N) return;
    const int F = H * D, nvec = F / E, lph = D / E;
    const float inv_sqrt_d = 1.0f / sqrtf((float)D);
    Chunk<E> fu[NC], du[NC], fw[NC], dw[NC], acc[NC];
    float delta_u[NC];
    load_row_e<NC, E>(ft, u, F, lane, nvec, fu);
    load_row_e<NC, E>(dout, u, F, lane, nvec, du);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        acc[i] = chunk_zero<E>();
        const int c = lane + 64 * i;
        delta_u[i] = (c < nvec) ? delta[(size_t)u * H + c / lph] : 0.f;
    }
    const int e0 = indptr[u], e1 = indptr[u + 1];
    for (int e = e0; e < e1; ++e) {
        const int w = indices[e];
        const int er = rev[e];
        load_row_e<NC, E>(ft, w, F, lane, nvec, fw);
        load_row_e<NC, E>(dout, w, F, lane, nvec, dw);
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + 64 * i;
            const int h = (c < nvec) ? c / lph : 0;
            const float a_uw = (c < nvec) ? alpha[(size_t)er * H + h] : 0.f;
            const float a_wu = (c < nvec) ? alpha[(size_t)e * H + h] : 0.f;
            const float d_w = (c < nvec) ? delta[(size_t)w * H + h] : 0.f;
            const float dot_src = group_sum(cdot<E>(dw[i], fu[i]), lph);
            const float dot_dst = group_sum(cdot<E>(du[i], fw[i]), lph);
            const float coef = (a_uw * (dot_src - d_w) + a_wu * (dot_dst - delta_u[i])) * inv_sqrt_d;
#pragma unroll
            for (int q = 0; q < E; ++q) acc[i].v[q] += a_uw * dw[i].v[q] + coef * fw[i].v[q];
        }
    }
    T* o = dft + (size_t)u * F;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + 64 * i;
        if (c < nvec) stc<E, T>(o, c, acc[i]);
    }
}

template <int NC, typename T, typename U>
__global__ __launch_bounds__(256) void neighbor_mean_kernel(int N, int F, const int* __restrict__ indptr, const int* __restrict__ indices,
                                                            const T* __restrict__ x, U* __restrict__ out, int scale_by_neighbor) {
    const int lane = threadIdx.x & 63;
    const int v = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (v >= N) return;
    const int nvec = F >> 2;
    float4 acc[NC], r[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int e0 = indptr[v], e1 = indptr[v + 1];
    const float self_scale = e1 > e0 ? 1.0f / (float)(e1 - e0) : 0.f;
    for (int e = e0; e < e1; ++e) {
        const int u = indices[e];
        load_row<NC>(x, u, F, lane, nvec, r);
        float sc = self_scale;
        if (scale_by_neighbor) {
            const int du = indptr[u + 1] - indptr[u];
            sc = du > 0 ? 1.0f / (float)du : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            acc[i].x += sc * r[i].x; acc[i].y += sc * r[i].y; acc[i].z += sc * r[i].z; acc[i].w += sc * r[i].w;
        }
    }
    U* o = out + (size_t)v * F;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + 64 * i;
        if (c < nvec) st4(o, c, acc[i]);
    }
}

inline int chunks_for(int F) {
    const int nvec = F / 4;
    const int nc = (nvec + 63) / 64;
    return nc <= 1 ? 1 : nc <= 2 ? 2 : nc <= 4 ? 4 : 8;
}
inline bool pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }
inline int chunks_for8(int F) {
    const int nc = (F / 8 + 63) / 64;
    return nc <= 1 ? 1 : nc <= 2 ? 2 : 4;
}
// 8 bf16 per lane: a head = D/8 adjacent lanes, rows 16-byte aligned
inline bool wide_ok(int H, int D, const void* a, const void* b) {
    return (D & 7) == 0 && D / 8 > 0 && (((D / 8) & (D / 8 - 1)) == 0) && H * D <= 2048 &&
           ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
}
inline bool head_shape_ok(int H, int D) {
    if (H <= 0 || D <= 0 || (D & 3)) return false;
    const int lph = D / 4;
    return pow2(lph) && lph <= 64 && H * D <= 2048;
}

#define DISPATCH_NC(nc, KERN, grid, st, ...)                                                    \
    switch (nc) {                                                                               \
        case 1: GRAPPA_LAUNCH(KERN<1>, grid, dim3(256), 0, st, __VA_ARGS__); break;        \
        case 2: GRAPPA_LAUNCH(KERN<2>, grid, dim3(256), 0, st, __VA_ARGS__); break;        \
        case 4: GRAPPA_LAUNCH(KERN<4>, grid, dim3(256), 0, st, __VA_ARGS__); break;        \
        default: GRAPPA_LAUNCH(KERN<8>, grid, dim3(256), 0, st, __VA_ARGS__); break;       \
    }
#define DISPATCH_NC_T(nc, KERN, T, grid, st, ...)                                                     \
    switch (nc) {                                                                                     \
        case 1: GRAPPA_LAUNCH((KERN<1, T>), grid, dim3(256), 0, st, __VA_ARGS__); break;         \
        case 2: GRAPPA_LAUNCH((KERN<2, T>), grid, dim3(256), 0, st, __VA_ARGS__); break;         \
        case 4: GRAPPA_LAUNCH((KERN<4, T>), grid, dim3(256), 0, st, __VA_ARGS__); break;         \
        default: GRAPPA_LAUNCH((KERN<8, T>), grid, dim3(256), 0, st, __VA_ARGS__); break;        \
    }

template <typename T>
int gat_fwd_impl(void* stream, int N, int E, int H, int D, const int* indptr, const int* indices, const T* ft, T* out, float* alpha) {
    if (N < 0 || E < 0 || !head_shape_ok(H, D)) return GRAPPA_ERR_ARG;
    if (N == 0) return GRAPPA_OK;
    if (!indptr || !ft || !out || !alpha || (E > 0 && !indices)) return GRAPPA_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(ft) | reinterpret_cast<uintptr_t>(out)) & (sizeof(T) == 4 ? 15 : 7)) return GRAPPA_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((N + 3) / 4);
    if constexpr (sizeof(T) == 2) {
        if (wide_ok(H, D, ft, out)) {               // 16-byte accesses: 8 bf16 per lane
            const int nc8 = chunks_for8(H * D);
            if (nc8 == 1) GRAPPA_LAUNCH((gat_fwd_kernel_e<1, 8, T>), grid, dim3(256), 0, st, N, H, D, indptr, indices, ft, out, alpha);
            else if (nc8 == 2) GRAPPA_LAUNCH((gat_fwd_kernel_e<2, 8, T>), grid, dim3(256), 0, st, N, H, D, indptr, indices, ft, out, alpha);
            else GRAPPA_LAUNCH((gat_fwd_kernel_e<4, 8, T>), grid, dim3(256), 0, st, N, H, D, indptr, indices, ft, out, alpha);
            return grappa_launch_status();
        }
    }
    DISPATCH_NC_T(chunks_for(H * D), gat_fwd_kernel, T, grid, st, N, H, D, indptr, indices, ft, out, alpha);
    return grappa_launch_status();
}

template <typename T>
int gat_bwd_impl(void* stream, int N, int E, int H, int D, const int* indptr, const int* indices, const int* rev, const T* ft, const T* out,
                 const float* alpha, const T* dout, T* dft, float* delta) {
    if (N < 0 || E < 0 || !head_shape_ok(H, D)) return GRAPPA_ERR_ARG;
    if (N == 0) return GRAPPA_OK;
    if (!indptr || !ft || !out || !alpha || !dout || !dft || !delta || (E > 0 && (!indices || !rev))) return GRAPPA_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(ft) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(dft)) & (sizeof(T) == 4 ? 15 : 7))
        return GRAPPA_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((N + 3) / 4);
    if constexpr (sizeof(T) == 2) {
        if (wide_ok(H, D, ft, out) && wide_ok(H, D, dout, dft)) {
            const int nc8 = chunks_for8(H * D);
#define GRAPPA_GAT_BWD8(NC8)                                                                                                                  \
    GRAPPA_LAUNCH((gat_delta_kernel_e<NC8, 8, T>), grid, dim3(256), 0, st, N, H, D, out, dout, delta);                                    \
    GRAPPA_LAUNCH((gat_bwd_kernel_e<NC8, 8, T>), grid, dim3(256), 0, st, N, H, D, indptr, indices, rev, ft, alpha, dout, delta, dft)
            if (nc8 == 1) { GRAPPA_GAT_BWD8(1); }
            else if (nc8 == 2) { GRAPPA_GAT_BWD8(2); }
            else { GRAPPA_GAT_BWD8(4); }
#undef GRAPPA_GAT_BWD8
            return grappa_launch_status();
        }
    }
    const int nc = chunks_for(H * D);
    DISPATCH_NC_T(nc, gat_delta_kernel, T, grid, st, N, H, D, out, dout, delta);
    DISPATCH_NC_T(nc, gat_bwd_kernel, T, grid, st, N, H, D, indptr, indices, rev, ft, alpha, dout, delta, dft);
    return grappa_launch_status();
}

}  // namespace

extern "C" int grappa_gat_fwd_f32(void* stream, int N, int E, int H, int D, const int* indptr, const int* indices, const float* ft,
                                  float* out, float* alpha) {
    return gat_fwd_impl<float>(stream, N, E, H, D, indptr, indices, ft, out, alpha);
}
extern "C" int grappa_gat_fwd_bf16(void* stream, int N, int E, int H, int D, const int* indptr, const int* indices, const uint16_t* ft,
                                   uint16_t* out, float* alpha) {
    return gat_fwd_impl<grappa_bf16_t>(stream, N, E, H, D, indptr, indices, ft, out, alpha);
}

extern "C" int grappa_gat_bwd_f32(void* stream, int N, int E, int H, int D, const int* indptr, const int* indices, const int* rev,
                                  const float* ft, const float* out, const float* alpha, const float* dout, float* dft, float* delta) {
    return gat_bwd_impl<float>(stream, N, E, H, D, indptr, indices, rev, ft, out, alpha, dout, dft, delta);
}
extern "C" int grappa_gat_bwd_bf16(void* stream, int N, int E, int H, int D, const int* indptr, const int* indices, const int* rev,
                                   const uint16_t* ft, const uint16_t* out, const float* alpha, const uint16_t* dout, uint16_t* dft, float* delta) {
    return gat_bwd_impl<grappa_bf16_t>(stream, N, E, H, D, indptr, indices, rev, ft, out, alpha, dout, dft, delta);
}

extern "C" int grappa_neighbor_mean_f32(void* stream, int N, int F, const int* indptr, const int* indices, const float* x, float* out,
                                        int scale_by_neighbor) {
    if (N < 0 || F <= 0 || (F & 3) || F > 2048) return GRAPPA_ERR_ARG;
    if (N == 0) return GRAPPA_OK;
    if (!indptr || !indices || !x || !out) return GRAPPA_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) return GRAPPA_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((N + 3) / 4);
    switch (chunks_for(F)) {
        case 1: GRAPPA_LAUNCH((neighbor_mean_kernel<1, float, float>), grid, dim3(256), 0, st, N, F, indptr, indices, x, out, scale_by_neighbor); break;
        case 2: GRAPPA_LAUNCH((neighbor_mean_kernel<2, float, float>), grid, dim3(256), 0, st, N, F, indptr, indices, x, out, scale_by_neighbor); break;
        case 4: GRAPPA_LAUNCH((neighbor_mean_kernel<4, float, float>), grid, dim3(256), 0, st, N, F, indptr, indices, x, out, scale_by_neighbor); break;
        default: GRAPPA_LAUNCH((neighbor_mean_kernel<8, float, float>), grid, dim3(256), 0, st, N, F, indptr, indices, x, out, scale_by_neighbor); break;
    }
    return grappa_launch_status();
}

// bf16 storage configuration: bf16 rows in; out bf16 (the aggregate is the A operand of a dense product) or fp32 (out_f32 != 0: the
// transposed aggregation of the backward pass, an fp32 addend of the product that follows)
extern "C" int grappa_neighbor_mean_bf16(void* stream, int N, int F, const int* indptr, const int* indices, const uint16_t* x, void* out, int out_f32,
                                         int scale_by_neighbor) {
    if (N < 0 || F <= 0 || (F & 3) || F > 2048) return GRAPPA_ERR_ARG;
    if (N == 0) return GRAPPA_OK;
    if (!indptr || !indices || !x || !out) return GRAPPA_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(x) & 7) || (reinterpret_cast<uintptr_t>(out) & (out_f32 ? 15 : 7))) return GRAPPA_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((N + 3) / 4);
#define NM_LAUNCH(NC)                                                                                                                           \
    if (out_f32) GRAPPA_LAUNCH((neighbor_mean_kernel<NC, grappa_bf16_t, float>), grid, dim3(256), 0, st, N, F, indptr, indices, x,          \
                                    reinterpret_cast<float*>(out), scale_by_neighbor);                                                          \
    else GRAPPA_LAUNCH((neighbor_mean_kernel<NC, grappa_bf16_t, grappa_bf16_t>), grid, dim3(256), 0, st, N, F, indptr, indices, x,          \
                            reinterpret_cast<grappa_bf16_t*>(out), scale_by_neighbor);
    switch (chunks_for(F)) {
        case 1: NM_LAUNCH(1) break;
        case 2: NM_LAUNCH(2) break;
        case 4: NM_LAUNCH(4) break;
        default: NM_LAUNCH(8) break;
    }
#undef NM_LAUNCH
    return grappa_launch_status();
}
