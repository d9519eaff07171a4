// HBM-bound row-wise kernels: LayerNorm fwd/bwd (one wavefront per row, row held in registers as
// float4 chunks), column sums (bias gradients), activation/dropout backward, sum of squares, fused
// Adam with global-norm clipping, charge encoding.  All fp32; every global access is a 16-byte
// coalesced access where the shape allows it.
#include "common.h"

namespace {

constexpr int LN_MAX_CHUNKS = 8;   // 8 float4 per lane * 64 lanes = 2048 floats

// magnitudes as fp32 bit patterns with the sign cleared: their unsigned order is the order of the magnitudes (amax.hip)
__device__ inline unsigned mag4(const float4& v) {
    return max(max(__float_as_uint(v.x) & 0x7fffffffu, __float_as_uint(v.y) & 0x7fffffffu),
               max(__float_as_uint(v.z) & 0x7fffffffu, __float_as_uint(v.w) & 0x7fffffffu));
}
__device__ inline unsigned wave_umax(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o, 64));
    return v;
}

// NCH = float4 chunks per lane (row width <= 256 * NCH): the row lives in registers, so the chunk count is a template parameter --
// sized for the widest row (8 chunks) every width would carry ~130 live registers and run at 3 wavefronts per SIMD
template <int NCH, bool STORE_STATS, typename T>
__device__ __forceinline__ void layernorm_fwd_body(int M, int W, const T* __restrict__ x, int ldx,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   T* __restrict__ y, int ldy, float* __restrict__ mean_out,
                                                   float* __restrict__ rstd_out, unsigned* __restrict__ y_amax, int vblock, int vnblocks) {
    const int lane = threadIdx.x & 63;
    const int wave_global = (vblock * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (vnblocks * blockDim.x) >> 6;
    const int nvec = W >> 2;
    // one row per wavefront and trip.  (Two rows per trip measured 20 % faster on the widest tables, but that kernel is the one
    // that returned deviating rows when several queues ran this library's kernels at once -- DESIGN.md section 6 -- so it is not used.)
    for (int row = wave_global; row < M; row += nwaves) {
        const T* xr = x + (size_t)row * ldx;
        float4 v[NCH];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nvec) {
                v[i] = ld4(xr, c);
                s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
            }
        }
        const float mean = wave_sum(s) / (float)W;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nvec) {
                const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
                q += (a * a + b * b) + (cc * cc + d * d);
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)W + 1e-5f);
        T* yr = y + (size_t)row * ldy;
        unsigned am = 0u;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nvec) {
                const float4 g = reinterpret_cast<const float4*>(gamma)[c];
                const float4 b = reinterpret_cast<const float4*>(beta)[c];
                float4 o;
                o.x = grappa_ln_apply(v[i].x, mean, rstd, g.x, b.x);
                o.y = grappa_ln_apply(v[i].y, mean, rstd, g.y, b.y);
                o.z = grappa_ln_apply(v[i].z, mean, rstd, g.z, b.z);
                o.w = grappa_ln_apply(v[i].w, mean, rstd, g.w, b.w);
                st4(yr, c, o);
                am = max(am, mag4(o));
            }
        }
        if (STORE_STATS && lane == 0) {
            mean_out[row] = mean;
            rstd_out[row] = rstd;
        }
        if (y_amax) {                                    // largest |y| of the row (fp32 bit pattern): scale of an F32_F16X3 product
            am = wave_umax(am);
            if (lane == 0) y_amax[row] = am;
        }
    }
}

template <int NCH, bool STORE_STATS, typename T>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(int M, int W, const T* __restrict__ x, int ldx,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            T* __restrict__ y, int ldy, float* __restrict__ mean_out,
                                                            float* __restrict__ rstd_out, unsigned* __restrict__ y_amax) {
    layernorm_fwd_body<NCH, STORE_STATS, T>(M, W, x, ldx, gamma, beta, y, ldy, mean_out, rstd_out, y_amax, blockIdx.x, gridDim.x);
}

// several LayerNorms in ONE launch (the same LayerNorm of the four writer heads, layer-locked: ops.MultiTransformerLayerFn): every item has
// its own rows, width, gamma / beta; a workgroup finds its item by the prefix of workgroups
struct LnFwdBatch {
    grappa_ln_fwd_item it[GRAPPA_ROW_BATCH_MAX];
    int blk_begin[GRAPPA_ROW_BATCH_MAX + 1];
    int count;
};
template <int NCH>
__global__ __launch_bounds__(256) void layernorm_fwd_batched_kernel(LnFwdBatch b) {
    int i = 0;
    while (i + 1 < b.count && (int)blockIdx.x >= b.blk_begin[i + 1]) ++i;
    const grappa_ln_fwd_item& t = b.it[i];
    layernorm_fwd_body<NCH, true, float>(t.M, t.W, t.x, t.ldx, t.gamma, t.beta, t.y, t.ldy, t.mean, t.rstd, t.y_amax, (int)blockIdx.x - b.blk_begin[i],
                                         b.blk_begin[i + 1] - b.blk_begin[i]);
}

// The same rows also written in the PAIR format (fp16 hi / lo halves scaled by the row's largest magnitude, common.h st_pairs4): the A
// operand of a following F32_F16X3 product split ONCE, by the kernel that has the whole row -- and its maximum -- in registers.  y (fp32)
// is optional: inference needs it only where the normalised rows are also a residual.
template <int NCH>
__global__ __launch_bounds__(256) void layernorm_fwd_pairs_kernel(int M, int W, const float* __restrict__ x, int ldx,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  float* __restrict__ y, int ldy, float* __restrict__ mean_out,
                                                                  float* __restrict__ rstd_out, unsigned* __restrict__ y_amax,
                                                                  uint16_t* __restrict__ pairs, int ldp) {
    const int lane = threadIdx.x & 63;
    const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int nvec = W >> 2;
    for (int row = wave_global; row < M; row += nwaves) {
        const float* xr = x + (size_t)row * ldx;
        float4 v[NCH];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nvec) {
                v[i] = ld4(xr, c);
                s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
            }
        }
        const float mean = wave_sum(s) / (float)W;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nvec) {
                const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
                q += (a * a + b * b) + (cc * cc + d * d);
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)W + 1e-5f);
        unsigned am = 0u;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nvec) {
                const float4 g = reinterpret_cast<const float4*>(gamma)[c];
                const float4 b = reinterpret_cast<const float4*>(beta)[c];
                float4 o;                                    // the same expressions as layernorm_fwd_kernel: the same bits
                o.x = grappa_ln_apply(v[i].x, mean, rstd, g.x, b.x);
                o.y = grappa_ln_apply(v[i].y, mean, rstd, g.y, b.y);
                o.z = grappa_ln_apply(v[i].z, mean, rstd, g.z, b.z);
                o.w = grappa_ln_apply(v[i].w, mean, rstd, g.w, b.w);
                v[i] = o;
                am = max(am, mag4(o));
            }
        }
        am = wave_umax(am);
        const int shift = grappa_amax_shift(am);
        uint16_t* pr = pairs + (size_t)row * ldp;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nvec && y) st4(y + (size_t)row * ldy, c, v[i]);
            st_pairs4_paired(pr, c, v[i], shift, c < nvec);         // (W % 32 == 0: nvec is even, a lane and its partner are in range together)
        }
        if (lane == 0) {
            if (mean_out) mean_out[row] = mean;
            if (rstd_out) rstd_out[row] = rstd;
            y_amax[row] = am;
        }
    }
}

// the dropout backward of the tensor the LayerNorm's input came out of, applied to dx while the row is in registers: dz = mask * dx / (1 - p)
// and dz's row maxima (what act_dropout_bwd_rows_kernel would compute from dx in a launch of its own)
struct LnBwdDrop {
    float p, scale;
    uint64_t seed;
    const uint64_t* salt;
    float* dz;
    int lddz;
    unsigned* dz_amax;
};

// dx per row; per-block partial dgamma/dbeta into part[block][2][W]
template <int NCH, typename T, bool DROP = false>
__device__ __forceinline__ void layernorm_bwd_body(int M, int W, const T* __restrict__ dy, int lddy,
                                                   const T* __restrict__ x, int ldx, const float* __restrict__ mean,
                                                   const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                   T* __restrict__ dx, int lddx, float* __restrict__ part,
                                                   unsigned* __restrict__ dx_amax, int vblock, int vnblocks, const LnBwdDrop dr = LnBwdDrop{}) {
    extern __shared__ float red[];   // [4 waves][2][W]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wave_global = vblock * 4 + wave;
    const int nwaves = vnblocks * 4;
    const int nvec = W >> 2;
    float4 dg[NCH], db[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        dg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        db[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    uint64_t seed_ = 0;
    if constexpr (DROP) seed_ = grappa_salted(dr.seed, dr.salt);
    for (int row = wave_global; row < M; row += nwaves) {
        const T* xr = x + (size_t)row * ldx;
        const T* dyr = dy + (size_t)row * lddy;
        const float mu = mean[row], rs = rstd[row];
        float4 xh[NCH], g[NCH];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nvec) {
                const float4 xv = ld4(xr, c), dv = ld4(dyr, c), gm = reinterpret_cast<const float4*>(gamma)[c];
                xh[i].x = (xv.x - mu) * rs; xh[i].y = (xv.y - mu) * rs; xh[i].z = (xv.z - mu) * rs; xh[i].w = (xv.w - mu) * rs;
                g[i].x = dv.x * gm.x; g[i].y = dv.y * gm.y; g[i].z = dv.z * gm.z; g[i].w = dv.w * gm.w;
                s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
                s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
                dg[i].x += dv.x * xh[i].x; dg[i].y += dv.y * xh[i].y; dg[i].z += dv.z * xh[i].z; dg[i].w += dv.w * xh[i].w;
                db[i].x += dv.x; db[i].y += dv.y; db[i].z += dv.z; db[i].w += dv.w;
            }
        }
        const float m1 = wave_sum(s1) / (float)W, m2 = wave_sum(s2) / (float)W;
        T* dxr = dx + (size_t)row * lddx;
        unsigned am = 0u, amz = 0u;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nvec) {
                float4 o;
                o.x = rs * (g[i].x - m1 - xh[i].x * m2);
                o.y = rs * (g[i].y - m1 - xh[i].y * m2);
                o.z = rs * (g[i].z - m1 - xh[i].z * m2);
                o.w = rs * (g[i].w - m1 - xh[i].w * m2);
                st4(dxr, c, o);
                am = max(am, mag4(o));
                if constexpr (DROP) {
                    const uint64_t idx = (uint64_t)row * (uint64_t)W + (uint64_t)(c << 2);
                    float4 v;
                    v.x = grappa_keep(seed_, idx, dr.p) ? o.x * dr.scale : 0.f;
                    v.y = grappa_keep(seed_, idx + 1, dr.p) ? o.y * dr.scale : 0.f;
                    v.z = grappa_keep(seed_, idx + 2, dr.p) ? o.z * dr.scale : 0.f;
                    v.w = grappa_keep(seed_, idx + 3, dr.p) ? o.w * dr.scale : 0.f;
                    *reinterpret_cast<float4*>(dr.dz + (size_t)row * dr.lddz + (c << 2)) = v;
                    amz = max(amz, mag4(v));
                }
            }
        }
        if (dx_amax) {
            am = wave_umax(am);
            if (lane == 0) dx_amax[row] = am;
        }
        if constexpr (DROP) {
            amz = wave_umax(amz);
            if (lane == 0) dr.dz_amax[row] = amz;
        }
    }
    // block reduction of the per-wave partials (fixed order -> reproducible)
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * i;
        if (c < nvec) {
            reinterpret_cast<float4*>(red + (size_t)(wave * 2 + 0) * W)[c] = dg[i];
            reinterpret_cast<float4*>(red + (size_t)(wave * 2 + 1) * W)[c] = db[i];
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < 2 * W; j += 256) {
        const int which = j / W, col = j % W;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) s += red[(size_t)(w * 2 + which) * W + col];
        part[((size_t)vblock * 2 + which) * W + col] = s;
    }
}

template <int NCH, typename T>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(int M, int W, const T* __restrict__ dy, int lddy,
                                                            const T* __restrict__ x, int ldx, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                            T* __restrict__ dx, int lddx, float* __restrict__ part,
                                                            unsigned* __restrict__ dx_amax) {
    layernorm_bwd_body<NCH, T>(M, W, dy, lddy, x, ldx, mean, rstd, gamma, dx, lddx, part, dx_amax, blockIdx.x, gridDim.x);
}

template <int NCH>
__global__ __launch_bounds__(256) void layernorm_bwd_drop_kernel(int M, int W, const float* __restrict__ dy, int lddy,
                                                                 const float* __restrict__ x, int ldx, const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                 float* __restrict__ dx, int lddx, float* __restrict__ part,
                                                                 unsigned* __restrict__ dx_amax, LnBwdDrop dr) {
    layernorm_bwd_body<NCH, float, true>(M, W, dy, lddy, x, ldx, mean, rstd, gamma, dx, lddx, part, dx_amax, blockIdx.x, gridDim.x, dr);
}

struct LnBwdBatch {
    grappa_ln_bwd_item it[GRAPPA_ROW_BATCH_MAX];
    int blk_begin[GRAPPA_ROW_BATCH_MAX + 1];
    int count;
};
template <int NCH>
__global__ __launch_bounds__(256) void layernorm_bwd_batched_kernel(LnBwdBatch b) {
    int i = 0;
    while (i + 1 < b.count && (int)blockIdx.x >= b.blk_begin[i + 1]) ++i;
    const grappa_ln_bwd_item& t = b.it[i];
    layernorm_bwd_body<NCH, float>(t.M, t.W, t.dy, t.lddy, t.x, t.ldx, t.mean, t.rstd, t.gamma, t.dx, t.lddx, t.part, t.dx_amax,
                                   (int)blockIdx.x - b.blk_begin[i], b.blk_begin[i + 1] - b.blk_begin[i]);
}

// out[g*out_stride + j] (+)= sum over rows b in group g of part[b*stride + j]   (grid.y = number of groups)
__global__ __launch_bounds__(256) void reduce_rows_kernel(int nrows, int stride, int n, const float* __restrict__ part,
                                                          float* __restrict__ out, int out_stride, int accumulate, int rows_per_group,
                                                          float* __restrict__ out2, int n_first) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const int b0 = blockIdx.y * rows_per_group;
    const int b1 = min(nrows, b0 + rows_per_group);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = b0;
    for (; b + 3 < b1; b += 4) {       // four independent loads in flight
        s0 += part[(size_t)b * stride + j];
        s1 += part[(size_t)(b + 1) * stride + j];
        s2 += part[(size_t)(b + 2) * stride + j];
        s3 += part[(size_t)(b + 3) * stride + j];
    }
    for (; b < b1; ++b) s0 += part[(size_t)b * stride + j];
    const float s = (s0 + s1) + (s2 + s3);
    float* o = out + (size_t)blockIdx.y * out_stride + j;
    if (out2 && j >= n_first) o = out2 + (j - n_first);          // final stage with two destinations: columns [n_first, n) go to out2
    *o = accumulate ? *o + s : s;
}

constexpr int REDUCE_GROUPS = 32;

// fixed-order two-stage reduction of per-block partials; scratch holds REDUCE_GROUPS * n floats
// out2 != nullptr: the n columns are two vectors back to back, [0, n_first) -> out and [n_first, n) -> out2 (LayerNorm's dgamma | dbeta)
inline void reduce_rows(hipStream_t st, int nrows, int stride, int n, const float* part, float* out, int accumulate, float* scratch,
                        float* out2 = nullptr, int n_first = 0) {
    const dim3 gx((n + 255) / 256);
    if (nrows <= 2 * REDUCE_GROUPS) {
        GRAPPA_LAUNCH(reduce_rows_kernel, dim3(gx.x, 1), dim3(256), 0, st, nrows, stride, n, part, out, 0, accumulate, nrows, out2, n_first);
        return;
    }
    const int rpg = (nrows + REDUCE_GROUPS - 1) / REDUCE_GROUPS;
    const int groups = (nrows + rpg - 1) / rpg;
    GRAPPA_LAUNCH(reduce_rows_kernel, dim3(gx.x, groups), dim3(256), 0, st, nrows, stride, n, part, scratch, n, 0, rpg, (float*)nullptr, 0);
    GRAPPA_LAUNCH(reduce_rows_kernel, dim3(gx.x, 1), dim3(256), 0, st, groups, n, n, scratch, out, 0, accumulate, groups, out2, n_first);
}

// Column sums of many partial sets in one launch (the deferred LayerNorm parameter gradients of a backward pass): a block of 1024
// threads = 64 columns x 16 row groups of one item; group g adds rows g, g + 16, ... (four loads in flight), the groups are combined
// through the LDS in the order 0 .. 15: a fixed order, the same bits every run.
constexpr int COLSUM_BATCH_MAX = 64;
struct ColsumBatch {
    grappa_colsum_item it[COLSUM_BATCH_MAX];
    int blk_begin[COLSUM_BATCH_MAX + 1];
    int count;
};
__global__ __launch_bounds__(1024) void colsum_batched_kernel(ColsumBatch b) {
    __shared__ float red[16][64];
    int i = 0;
    while (i + 1 < b.count && (int)blockIdx.x >= b.blk_begin[i + 1]) ++i;
    const grappa_colsum_item& it = b.it[i];
    const int j = ((int)blockIdx.x - b.blk_begin[i]) * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (j < it.n) {
        int r = g;
        for (; r + 48 < it.nrows; r += 64) {
            s0 += it.part[(size_t)r * it.n + j];
            s1 += it.part[(size_t)(r + 16) * it.n + j];
            s2 += it.part[(size_t)(r + 32) * it.n + j];
            s3 += it.part[(size_t)(r + 48) * it.n + j];
        }
        for (; r < it.nrows; r += 16) s0 += it.part[(size_t)r * it.n + j];
    }
    red[g][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && j < it.n) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) s += red[q][threadIdx.x];
        float* o = (it.out2 && j >= it.n_first) ? it.out2 + (j - it.n_first) : it.out + j;
        *o = it.accumulate ? *o + s : s;
    }
}

// column sums: block = 256 threads = 64 columns x 4 row-lanes... simple: each block owns a row stripe,
// threads stride over columns; partials[block][N]
__global__ __launch_bounds__(256) void colsum_partial_kernel(int M, int N, const float* __restrict__ x, int ldx, int rows_per_block,
                                                             float* __restrict__ part) {
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    for (int n = threadIdx.x; n < N; n += 256) {
        float s = 0.f;
        for (int r = r0; r < r1; ++r) s += x[(size_t)r * ldx + n];
        part[(size_t)blockIdx.x * N + n] = s;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void act_dropout_bwd_kernel(int M, int N, const T* __restrict__ dy, int lddy,
                                                              const T* __restrict__ y, int ldy, float p, float scale, uint64_t seed, const uint64_t* __restrict__ salt,
                                                              T* __restrict__ dz, int lddz) {
    seed = grappa_salted(seed, salt);
    const size_t total = (size_t)M * N;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int m = (int)(i / N), n = (int)(i % N);
        float v = ld1(dy, (size_t)m * lddy + n);
        if (p > 0.f) v = grappa_keep(seed, i, p) ? v * scale : 0.f;
        if (y) v *= grappa_elu_grad_from_out(ld1(y, (size_t)m * ldy + n));
        st1(dz, (size_t)m * lddz + n, v);
    }
}

// the same with 16-byte accesses: N, the leading dimensions and the base addresses are multiples of 4 floats (host-checked)
template <typename T>
__global__ __launch_bounds__(256) void act_dropout_bwd_vec_kernel(int M, int N, const T* __restrict__ dy, int lddy,
                                                                  const T* __restrict__ y, int ldy, float p, float scale, uint64_t seed, const uint64_t* __restrict__ salt,
                                                                  T* __restrict__ dz, int lddz) {
    seed = grappa_salted(seed, salt);
    const int n4 = N >> 2;
    const size_t total = (size_t)M * n4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int m = (int)(i / n4), n = (int)(i - (size_t)m * n4) << 2;
        float4 v = ld4(dy + (size_t)m * lddy + n, 0);
        if (p > 0.f) {
            const uint64_t idx = (uint64_t)m * (uint64_t)N + (uint64_t)n;          // element index of the forward's mask
            v.x = grappa_keep(seed, idx, p) ? v.x * scale : 0.f;
            v.y = grappa_keep(seed, idx + 1, p) ? v.y * scale : 0.f;
            v.z = grappa_keep(seed, idx + 2, p) ? v.z * scale : 0.f;
            v.w = grappa_keep(seed, idx + 3, p) ? v.w * scale : 0.f;
        }
        if (y) {
            const float4 t = ld4(y + (size_t)m * ldy + n, 0);
            v.x *= grappa_elu_grad_from_out(t.x); v.y *= grappa_elu_grad_from_out(t.y);
            v.z *= grappa_elu_grad_from_out(t.z); v.w *= grappa_elu_grad_from_out(t.w);
        }
        st4(dz + (size_t)m * lddz + n, 0, v);
    }
}

// the same, one wavefront per row (as LayerNorm), so that the row's largest |dz| falls out of a wavefront reduction (dz_amax: the scale
// of the F32_F16X3 products that read dz).  N <= 256 * NCH, 16-byte accesses.
template <int NCH>
__device__ __forceinline__ void act_dropout_bwd_rows_body(int M, int N, const float* __restrict__ dy, int lddy,
                                                          const float* __restrict__ y, int ldy, float p, float scale, uint64_t seed, const uint64_t* __restrict__ salt,
                                                          float* __restrict__ dz, int lddz, unsigned* __restrict__ dz_amax, int vblock, int vnblocks) {
    seed = grappa_salted(seed, salt);
    const int lane = threadIdx.x & 63;
    const int wave_global = (vblock * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (vnblocks * blockDim.x) >> 6;
    const int nvec = N >> 2;
    for (int row = wave_global; row < M; row += nwaves) {
        unsigned am = 0u;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nvec) {
                float4 v = ld4(dy + (size_t)row * lddy, c);
                if (p > 0.f) {
                    const uint64_t idx = (uint64_t)row * (uint64_t)N + (uint64_t)(c << 2);
                    v.x = grappa_keep(seed, idx, p) ? v.x * scale : 0.f;
                    v.y = grappa_keep(seed, idx + 1, p) ? v.y * scale : 0.f;
                    v.z = grappa_keep(seed, idx + 2, p) ? v.z * scale : 0.f;
                    v.w = grappa_keep(seed, idx + 3, p) ? v.w * scale : 0.f;
                }
                if (y) {
                    const float4 t = ld4(y + (size_t)row * ldy, c);
                    v.x *= grappa_elu_grad_from_out(t.x); v.y *= grappa_elu_grad_from_out(t.y);
                    v.z *= grappa_elu_grad_from_out(t.z); v.w *= grappa_elu_grad_from_out(t.w);
                }
                st4(dz + (size_t)row * lddz, c, v);
                am = max(am, mag4(v));
            }
        }
        am = wave_umax(am);
        if (lane == 0) dz_amax[row] = am;
    }
}

template <int NCH>
__global__ __launch_bounds__(256) void act_dropout_bwd_rows_kernel(int M, int N, const float* __restrict__ dy, int lddy,
                                                                   const float* __restrict__ y, int ldy, float p, float scale, uint64_t seed, const uint64_t* __restrict__ salt,
                                                                   float* __restrict__ dz, int lddz, unsigned* __restrict__ dz_amax) {
    act_dropout_bwd_rows_body<NCH>(M, N, dy, lddy, y, ldy, p, scale, seed, salt, dz, lddz, dz_amax, blockIdx.x, gridDim.x);
}

struct ActDropBatch {
    grappa_act_dropout_item it[GRAPPA_ROW_BATCH_MAX];
    int blk_begin[GRAPPA_ROW_BATCH_MAX + 1];
    int count;
};
template <int NCH>
__global__ __launch_bounds__(256) void act_dropout_bwd_batched_kernel(ActDropBatch b) {
    int i = 0;
    while (i + 1 < b.count && (int)blockIdx.x >= b.blk_begin[i + 1]) ++i;
    const grappa_act_dropout_item& t = b.it[i];
    act_dropout_bwd_rows_body<NCH>(t.M, t.N, t.dy, t.lddy, t.y, t.ldy, t.drop_p, t.drop_p > 0.f ? 1.0f / (1.0f - t.drop_p) : 1.0f, t.drop_seed, t.drop_salt, t.dz, t.lddz,
                                   t.dz_amax, (int)blockIdx.x - b.blk_begin[i], b.blk_begin[i + 1] - b.blk_begin[i]);
}

// the same rows written in the PAIR format (dz itself optional): the A operand of the input-gradient product and an operand of the
// weight-gradient product behind it, split once by the kernel that holds the whole row
template <int NCH>
__global__ __launch_bounds__(256) void act_dropout_bwd_pairs_kernel(int M, int N, const float* __restrict__ dy, int lddy,
                                                                    const float* __restrict__ y, int ldy, float p, float scale, uint64_t seed, const uint64_t* __restrict__ salt,
                                                                    float* __restrict__ dz, int lddz, unsigned* __restrict__ dz_amax,
                                                                    uint16_t* __restrict__ pairs, int ldp) {
    seed = grappa_salted(seed, salt);
    const int lane = threadIdx.x & 63;
    const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int nvec = N >> 2;
    for (int row = wave_global; row < M; row += nwaves) {
        unsigned am = 0u;
        float4 keep[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < nvec) {
                v = ld4(dy + (size_t)row * lddy, c);
                if (p > 0.f) {
                    const uint64_t idx = (uint64_t)row * (uint64_t)N + (uint64_t)(c << 2);
                    v.x = grappa_keep(seed, idx, p) ? v.x * scale : 0.f;
                    v.y = grappa_keep(seed, idx + 1, p) ? v.y * scale : 0.f;
                    v.z = grappa_keep(seed, idx + 2, p) ? v.z * scale : 0.f;
                    v.w = grappa_keep(seed, idx + 3, p) ? v.w * scale : 0.f;
                }
                if (y) {
                    const float4 t = ld4(y + (size_t)row * ldy, c);
                    v.x *= grappa_elu_grad_from_out(t.x); v.y *= grappa_elu_grad_from_out(t.y);
                    v.z *= grappa_elu_grad_from_out(t.z); v.w *= grappa_elu_grad_from_out(t.w);
                }
                if (dz) st4(dz + (size_t)row * lddz, c, v);
                am = max(am, mag4(v));
            }
            keep[i] = v;
        }
        am = wave_umax(am);
        const int shift = grappa_amax_shift(am);
        uint16_t* pr = pairs + (size_t)row * ldp;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            st_pairs4_paired(pr, c, keep[i], shift, c < nvec);      // (N % 32 == 0: a lane and its partner are in range together)
        }
        if (lane == 0) dz_amax[row] = am;
    }
}

__global__ __launch_bounds__(256) void add_kernel(size_t n, const float* __restrict__ x, const float* __restrict__ z, float* __restrict__ y) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = x[i] + z[i];
}

__global__ __launch_bounds__(256) void sumsq_partial_kernel(size_t n, const float* __restrict__ x, float* __restrict__ part) {
    __shared__ float red[4];
    float s = 0.f;
    // 16-byte loads, four partial sums per thread (one dword per thread and trip read the 163 MB gradient buffer at 2.4 TB/s)
    const size_t n4 = (reinterpret_cast<uintptr_t>(x) & 15) == 0 ? n >> 2 : 0;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        s0 += v.x * v.x; s1 += v.y * v.y; s2 += v.z * v.z; s3 += v.w * v.w;
    }
    s = (s0 + s1) + (s2 + s3);
    for (size_t i = (n4 << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += x[i] * x[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void sumsq_final_kernel(int nblocks, const float* __restrict__ part, float* __restrict__ out, int accumulate) {
    float s = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 64) s += part[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[0] = accumulate ? out[0] + s : s;
}

__global__ __launch_bounds__(256) void adam_kernel(size_t n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, float lr, float b1, float b2, float eps, float wd,
                                                   float bc1, float bc2, float grad_scale, const float* __restrict__ sumsq, float max_norm) {
    float clip = 1.0f;
    if (sumsq) {
        const float norm = sqrtf(sumsq[0]) * grad_scale;
        clip = fminf(1.0f, max_norm / (norm + 1e-6f));
    }
    const float gs = grad_scale * clip;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float gi = g[i] * gs;
        const float pi = p[i];
        if (wd != 0.f) gi += wd * pi;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
        p[i] = pi - (lr / bc1) * (mi / denom);
    }
}

// the same with the learning rate and the step count read from device memory: a captured (hipGraph) train step replays with the values
// the host -- or a kernel of the graph itself -- wrote last
__global__ __launch_bounds__(256) void adam_dyn_kernel(size_t n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, const float* __restrict__ lr_dev, float b1, float b2, float eps, float wd,
                                                       const int* __restrict__ step_dev, float grad_scale, const float* __restrict__ sumsq, float max_norm) {
    const float lr = lr_dev[0];
    const int step = step_dev[0];
    const float bc1 = 1.0f - powf(b1, (float)step), bc2 = 1.0f - powf(b2, (float)step);
    float clip = 1.0f;
    if (sumsq) {
        const float norm = sqrtf(sumsq[0]) * grad_scale;
        clip = fminf(1.0f, max_norm / (norm + 1e-6f));
    }
    const float gs = grad_scale * clip;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float gi = g[i] * gs;
        const float pi = p[i];
        if (wd != 0.f) gi += wd * pi;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
        p[i] = pi - (lr / bc1) * (mi / denom);
    }
}

__global__ __launch_bounds__(256) void charge_encoding_kernel(int N, const float* __restrict__ q, int dim, float lo, float hi,
                                                              float* __restrict__ out, int ldo, int col0) {
    const int half = dim / 2;
    const int total = N * half;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int n = i / half, j = i % half;
        const float v = fminf(fmaxf(q[n], lo), hi);
        const float s = (v + hi) / (hi - lo);
        const float f = expf((float)j * -logf(10000.0f) / (float)half);
        out[(size_t)n * ldo + col0 + 2 * j] = sinf(s * f);
        out[(size_t)n * ldo + col0 + 2 * j + 1] = cosf(s * f);
    }
}

inline int grid_for(size_t n, int cap = 2048) {
    size_t b = (n + 255) / 256;
    if (b < 1) b = 1;
    if (b > (size_t)cap) b = cap;
    return (int)b;
}
inline int ln_blocks(int M) {
    int b = (M + 3) / 4;
    if (b > 1024) b = 1024;          // 4 wavefronts per SIMD on 256 CUs (2048 measured no faster)
    if (b < 1) b = 1;
    return b;
}
inline int colsum_blocks(int M) {
    int b = (M + 63) / 64;
    if (b > 512) b = 512;
    if (b < 1) b = 1;
    return b;
}

}  // namespace

namespace {
template <typename T>
int layernorm_fwd_impl(void* stream, int M, int W, const T* x, int ldx, const float* gamma, const float* beta, T* y, int ldy, float* mean,
                       float* rstd, unsigned* y_amax = nullptr) {
    if (M < 0 || W <= 0 || (W & 3) || W > 2048 || (ldx & 3) || (ldy & 3)) return GRAPPA_ERR_ARG;
    if (M == 0) return GRAPPA_OK;
    if (!x || !gamma || !beta || !y) return GRAPPA_ERR_ARG;
    const uintptr_t amask = sizeof(T) == 4 ? 15 : 7;
    if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & amask) || ((reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) & 15))
        return GRAPPA_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // the forward kernel is light (<= 54 registers up to W = 1024): 2048 blocks = 8 wavefronts per SIMD keep twice the rows in flight
    const int blocks = W <= 1024 ? ((M + 3) / 4 > 2048 ? 2048 : (M + 3) / 4) : ln_blocks(M);
#define GRAPPA_LN_FWD(NCH)                                                                                                              \
    if (mean && rstd)                                                                                                                   \
        GRAPPA_LAUNCH((layernorm_fwd_kernel<NCH, true, T>), dim3(blocks), dim3(256), 0, st, M, W, x, ldx, gamma, beta, y, ldy, mean, rstd, y_amax); \
    else                                                                                                                                \
        GRAPPA_LAUNCH((layernorm_fwd_kernel<NCH, false, T>), dim3(blocks), dim3(256), 0, st, M, W, x, ldx, gamma, beta, y, ldy, mean, rstd, y_amax)
    if (W <= 256) { GRAPPA_LN_FWD(1); }
    else if (W <= 512) { GRAPPA_LN_FWD(2); }
    else if (W <= 1024) { GRAPPA_LN_FWD(4); }
    else { GRAPPA_LN_FWD(8); }
#undef GRAPPA_LN_FWD
    return grappa_launch_status();
}

template <typename T>
int layernorm_bwd_impl(void* stream, int M, int W, const T* dy, int lddy, const T* x, int ldx, const float* mean, const float* rstd,
                       const float* gamma, T* dx, int lddx, float* dgamma, float* dbeta, int accumulate, void* ws, size_t ws_bytes,
                       unsigned* dx_amax = nullptr, const LnBwdDrop* drop = nullptr) {
    if (M < 0 || W <= 0 || (W & 3) || W > 2048 || (ldx & 3) || (lddy & 3) || (lddx & 3)) return GRAPPA_ERR_ARG;
    if (drop && (sizeof(T) != 4 || !drop->dz || !drop->dz_amax || (drop->lddz & 3) || drop->lddz < W || (reinterpret_cast<uintptr_t>(drop->dz) & 15) ||
                 !(drop->p > 0.f) || drop->p >= 1.f))
        return GRAPPA_ERR_ARG;
    if (M == 0) return GRAPPA_OK;
    if (!dy || !x || !mean || !rstd || !gamma || !dx || (accumulate != 2 && (!dgamma || !dbeta))) return GRAPPA_ERR_ARG;
    const uintptr_t amask = sizeof(T) == 4 ? 15 : 7;
    if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & amask) || (reinterpret_cast<uintptr_t>(gamma) & 15))
        return GRAPPA_ERR_ARG;
    const int blocks = ln_blocks(M);
    const size_t need = ((size_t)blocks * 2 * W + (size_t)REDUCE_GROUPS * 2 * W) * sizeof(float);
    if (!ws || ws_bytes < need) return GRAPPA_ERR_WORKSPACE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float* part = reinterpret_cast<float*>(ws);
    float* scratch = part + (size_t)blocks * 2 * W;
    const size_t smem = (size_t)4 * 2 * W * sizeof(float);
#define GRAPPA_LN_BWD(NCH)                                                                                                                     \
    do {                                                                                                                                       \
        if constexpr (sizeof(T) == 4) {                                                                                                        \
            if (drop) {                                                                                                                        \
                GRAPPA_LAUNCH((layernorm_bwd_drop_kernel<NCH>), dim3(blocks), dim3(256), smem, st, M, W, dy, lddy, x, ldx, mean, rstd, gamma, dx, lddx, part, dx_amax, *drop); \
                break;                                                                                                                         \
            }                                                                                                                                  \
        }                                                                                                                                      \
        GRAPPA_LAUNCH((layernorm_bwd_kernel<NCH, T>), dim3(blocks), dim3(256), smem, st, M, W, dy, lddy, x, ldx, mean, rstd, gamma, dx, lddx, part, dx_amax); \
    } while (0)
    if (W <= 256) GRAPPA_LN_BWD(1);
    else if (W <= 512) GRAPPA_LN_BWD(2);
    else if (W <= 1024) GRAPPA_LN_BWD(4);
    else GRAPPA_LN_BWD(8);
#undef GRAPPA_LN_BWD
    int rc = grappa_launch_status();
    if (rc) return rc;
    if (accumulate == 2) return GRAPPA_OK;                  // deferred: the caller reduces the partials (grappa_colsum_partials_batched)
    reduce_rows(st, blocks, 2 * W, 2 * W, part, dgamma, accumulate, scratch, dbeta, W);     // dgamma | dbeta in one pass (two launches, not four)
    return grappa_launch_status();
}

template <typename T>
int act_dropout_bwd_impl(void* stream, int M, int N, const T* dy, int lddy, const T* y, int ldy, float drop_p, uint64_t drop_seed, T* dz, int lddz, const uint64_t* drop_salt) {
    if (M < 0 || N < 0 || drop_p < 0.f || drop_p >= 1.f) return GRAPPA_ERR_ARG;
    if (M == 0 || N == 0) return GRAPPA_OK;
    if (!dy || !dz) return GRAPPA_ERR_ARG;
    const float scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    const uintptr_t amask = sizeof(T) == 4 ? 15 : 7;
    const bool vec = (N & 3) == 0 && (lddy & 3) == 0 && (lddz & 3) == 0 && (!y || (ldy & 3) == 0) &&
                     ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dz) | reinterpret_cast<uintptr_t>(y)) & amask) == 0;
    if (vec)
        GRAPPA_LAUNCH(act_dropout_bwd_vec_kernel<T>, dim3(grid_for((size_t)M * (N >> 2))), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                           M, N, dy, lddy, y, ldy, drop_p, scale, drop_seed, drop_salt, dz, lddz);
    else
        GRAPPA_LAUNCH(act_dropout_bwd_kernel<T>, dim3(grid_for((size_t)M * N)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                           M, N, dy, lddy, y, ldy, drop_p, scale, drop_seed, drop_salt, dz, lddz);
    return grappa_launch_status();
}

// element-type conversion of a 2-d view (the rare places where a bf16 tensor meets an fp32-only kernel or the reverse)
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void convert_kernel(int M, int N, const TI* __restrict__ x, int ldx, TO* __restrict__ y, int ldy) {
    const size_t total = (size_t)M * N;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int m = (int)(i / N), n = (int)(i % N);
        st1(y, (size_t)m * ldy + n, ld1(x, (size_t)m * ldx + n));
    }
}
}  // namespace

extern "C" int grappa_layernorm_fwd_f32(void* stream, int M, int W, const float* x, int ldx, const float* gamma, const float* beta,
                                        float* y, int ldy, float* mean, float* rstd) {
    return layernorm_fwd_impl<float>(stream, M, W, x, ldx, gamma, beta, y, ldy, mean, rstd);
}
extern "C" int grappa_layernorm_fwd_pairs_f32(void* stream, int M, int W, const float* x, int ldx, const float* gamma, const float* beta,
                                              float* y, int ldy, float* mean, float* rstd, uint32_t* y_amax, uint16_t* pairs, int ldp) {
    if (M < 0 || W <= 0 || (W & 31) || W > 2048 || (ldx & 3) || (y && (ldy & 3)) || ldp < 2 * W || (ldp & 7)) return GRAPPA_ERR_ARG;
    if (M == 0) return GRAPPA_OK;
    if (!x || !gamma || !beta || !y_amax || !pairs) return GRAPPA_ERR_ARG;
    if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta) |
          reinterpret_cast<uintptr_t>(pairs)) & 15) != 0)
        return GRAPPA_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int blocks = W <= 1024 ? ((M + 3) / 4 > 2048 ? 2048 : (M + 3) / 4) : ln_blocks(M);
#define GRAPPA_LN_FWD_PAIRS(NCH) GRAPPA_LAUNCH((layernorm_fwd_pairs_kernel<NCH>), dim3(blocks), dim3(256), 0, st, M, W, x, ldx, gamma, beta, y, ldy, mean, rstd, y_amax, pairs, ldp)
    if (W <= 256) GRAPPA_LN_FWD_PAIRS(1);
    else if (W <= 512) GRAPPA_LN_FWD_PAIRS(2);
    else if (W <= 1024) GRAPPA_LN_FWD_PAIRS(4);
    else GRAPPA_LN_FWD_PAIRS(8);
#undef GRAPPA_LN_FWD_PAIRS
    return grappa_launch_status();
}
extern "C" int grappa_layernorm_fwd_bf16(void* stream, int M, int W, const uint16_t* x, int ldx, const float* gamma, const float* beta,
                                         uint16_t* y, int ldy, float* mean, float* rstd) {
    return layernorm_fwd_impl<grappa_bf16_t>(stream, M, W, x, ldx, gamma, beta, y, ldy, mean, rstd);
}
extern "C" int grappa_convert_f32_to_bf16(void* stream, int M, int N, const float* x, int ldx, uint16_t* y, int ldy) {
    if (M < 0 || N < 0) return GRAPPA_ERR_ARG;
    if (M == 0 || N == 0) return GRAPPA_OK;
    if (!x || !y) return GRAPPA_ERR_ARG;
    GRAPPA_LAUNCH((convert_kernel<float, grappa_bf16_t>), dim3(grid_for((size_t)M * N)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), M, N, x, ldx, y, ldy);
    return grappa_launch_status();
}
extern "C" int grappa_convert_bf16_to_f32(void* stream, int M, int N, const uint16_t* x, int ldx, float* y, int ldy) {
    if (M < 0 || N < 0) return GRAPPA_ERR_ARG;
    if (M == 0 || N == 0) return GRAPPA_OK;
    if (!x || !y) return GRAPPA_ERR_ARG;
    GRAPPA_LAUNCH((convert_kernel<grappa_bf16_t, float>), dim3(grid_for((size_t)M * N)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), M, N, x, ldx, y, ldy);
    return grappa_launch_status();
}

extern "C" int grappa_layernorm_fwd_amax_f32(void* stream, int M, int W, const float* x, int ldx, const float* gamma, const float* beta,
                                             float* y, int ldy, float* mean, float* rstd, uint32_t* y_amax) {
    return layernorm_fwd_impl<float>(stream, M, W, x, ldx, gamma, beta, y, ldy, mean, rstd, y_amax);
}
extern "C" size_t grappa_layernorm_bwd_workspace_bytes(int M, int W) {
    return ((size_t)ln_blocks(M) * 2 * W + (size_t)REDUCE_GROUPS * 2 * W) * sizeof(float);
}

extern "C" int grappa_layernorm_bwd_partial_rows(int M) { return M > 0 ? ln_blocks(M) : 0; }

extern "C" int grappa_colsum_partials_batched(void* stream, const grappa_colsum_item* items, int count) {
    if (count < 0 || (count > 0 && !items)) return GRAPPA_ERR_ARG;
    for (int i = 0; i < count; ++i) {
        const grappa_colsum_item& it = items[i];
        if (it.nrows < 0 || it.n <= 0 || !it.out || (it.nrows > 0 && !it.part) || (it.out2 && (it.n_first <= 0 || it.n_first >= it.n))) return GRAPPA_ERR_ARG;
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    for (int i0 = 0; i0 < count; i0 += COLSUM_BATCH_MAX) {
        ColsumBatch b;
        b.count = count - i0 < COLSUM_BATCH_MAX ? count - i0 : COLSUM_BATCH_MAX;
        b.blk_begin[0] = 0;
        for (int i = 0; i < b.count; ++i) {
            b.it[i] = items[i0 + i];
            b.blk_begin[i + 1] = b.blk_begin[i] + (b.it[i].n + 63) / 64;
        }
        GRAPPA_LAUNCH(colsum_batched_kernel, dim3(b.blk_begin[b.count]), dim3(1024), 0, st, b);
    }
    return grappa_launch_status();
}

extern "C" int grappa_layernorm_bwd_f32(void* stream, int M, int W, const float* dy, int lddy, const float* x, int ldx,
                                        const float* mean, const float* rstd, const float* gamma, float* dx, int lddx,
                                        float* dgamma, float* dbeta, int accumulate, void* ws, size_t ws_bytes) {
    return layernorm_bwd_impl<float>(stream, M, W, dy, lddy, x, ldx, mean, rstd, gamma, dx, lddx, dgamma, dbeta, accumulate, ws, ws_bytes);
}
extern "C" int grappa_layernorm_bwd_amax_f32(void* stream, int M, int W, const float* dy, int lddy, const float* x, int ldx,
                                             const float* mean, const float* rstd, const float* gamma, float* dx, int lddx,
                                             float* dgamma, float* dbeta, int accumulate, void* ws, size_t ws_bytes, uint32_t* dx_amax) {
    return layernorm_bwd_impl<float>(stream, M, W, dy, lddy, x, ldx, mean, rstd, gamma, dx, lddx, dgamma, dbeta, accumulate, ws, ws_bytes, dx_amax);
}
extern "C" int grappa_layernorm_bwd_drop_f32(void* stream, int M, int W, const float* dy, int lddy, const float* x, int ldx,
                                             const float* mean, const float* rstd, const float* gamma, float* dx, int lddx,
                                             float* dgamma, float* dbeta, int accumulate, void* ws, size_t ws_bytes, uint32_t* dx_amax,
                                             float drop_p, uint64_t drop_seed, float* dz, int lddz, uint32_t* dz_amax, const uint64_t* drop_salt) {
    const LnBwdDrop dr{drop_p, drop_p > 0.f && drop_p < 1.f ? 1.0f / (1.0f - drop_p) : 1.0f, drop_seed, drop_salt, dz, lddz, dz_amax};
    return layernorm_bwd_impl<float>(stream, M, W, dy, lddy, x, ldx, mean, rstd, gamma, dx, lddx, dgamma, dbeta, accumulate, ws, ws_bytes, dx_amax, &dr);
}
extern "C" int grappa_layernorm_bwd_bf16(void* stream, int M, int W, const uint16_t* dy, int lddy, const uint16_t* x, int ldx,
                                         const float* mean, const float* rstd, const float* gamma, uint16_t* dx, int lddx,
                                         float* dgamma, float* dbeta, int accumulate, void* ws, size_t ws_bytes) {
    return layernorm_bwd_impl<grappa_bf16_t>(stream, M, W, dy, lddy, x, ldx, mean, rstd, gamma, dx, lddx, dgamma, dbeta, accumulate, ws, ws_bytes);
}

extern "C" size_t grappa_colsum_workspace_bytes(int M, int N) {
    return ((size_t)colsum_blocks(M) * N + (size_t)REDUCE_GROUPS * N) * sizeof(float);
}

extern "C" int grappa_colsum_f32(void* stream, int M, int N, const float* x, int ldx, float* out, int accumulate, void* ws, size_t ws_bytes) {
    if (M < 0 || N <= 0 || !out) return GRAPPA_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (M == 0) {
        if (!accumulate) { if (hipMemsetAsync(out, 0, (size_t)N * sizeof(float), st) != hipSuccess) return GRAPPA_ERR_LAUNCH; }
        return GRAPPA_OK;
    }
    if (!x) return GRAPPA_ERR_ARG;
    const int blocks = colsum_blocks(M);
    if (!ws || ws_bytes < ((size_t)blocks * N + (size_t)REDUCE_GROUPS * N) * sizeof(float)) return GRAPPA_ERR_WORKSPACE;
    const int rpb = (M + blocks - 1) / blocks;
    const int used = (M + rpb - 1) / rpb;
    float* part = reinterpret_cast<float*>(ws);
    GRAPPA_LAUNCH(colsum_partial_kernel, dim3(used), dim3(256), 0, st, M, N, x, ldx, rpb, part);
    reduce_rows(st, used, N, N, part, out, accumulate, part + (size_t)blocks * N);
    return grappa_launch_status();
}

extern "C" int grappa_act_dropout_bwd_f32(void* stream, int M, int N, const float* dy, int lddy, const float* y, int ldy,
                                          float drop_p, uint64_t drop_seed, float* dz, int lddz, const uint64_t* drop_salt) {
    return act_dropout_bwd_impl<float>(stream, M, N, dy, lddy, y, ldy, drop_p, drop_seed, dz, lddz, drop_salt);
}
extern "C" int grappa_act_dropout_bwd_amax_f32(void* stream, int M, int N, const float* dy, int lddy, const float* y, int ldy,
                                               float drop_p, uint64_t drop_seed, float* dz, int lddz, uint32_t* dz_amax, const uint64_t* drop_salt) {
    if (!dz_amax) return act_dropout_bwd_impl<float>(stream, M, N, dy, lddy, y, ldy, drop_p, drop_seed, dz, lddz, drop_salt);
    if (M < 0 || N < 0 || drop_p < 0.f || drop_p >= 1.f) return GRAPPA_ERR_ARG;
    if (M == 0 || N == 0) return GRAPPA_OK;
    if (!dy || !dz) return GRAPPA_ERR_ARG;
    const bool rows = (N & 3) == 0 && N <= 2048 && (lddy & 3) == 0 && (lddz & 3) == 0 && (!y || (ldy & 3) == 0) &&
                      ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dz) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
    if (!rows) {                                         // odd shapes: the plain kernel, then one pass over dz
        const int rc = act_dropout_bwd_impl<float>(stream, M, N, dy, lddy, y, ldy, drop_p, drop_seed, dz, lddz, drop_salt);
        return rc != GRAPPA_OK ? rc : grappa_amax_f32(stream, M, N, dz, lddz, dz_amax, nullptr, nullptr, 0);
    }
    const float scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int blocks = (M + 3) / 4 > 2048 ? 2048 : (M + 3) / 4;
#define GRAPPA_ADB(NCH) GRAPPA_LAUNCH((act_dropout_bwd_rows_kernel<NCH>), dim3(blocks), dim3(256), 0, st, M, N, dy, lddy, y, ldy, drop_p, scale, drop_seed, drop_salt, dz, lddz, dz_amax)
    if (N <= 256) GRAPPA_ADB(1);
    else if (N <= 512) GRAPPA_ADB(2);
    else if (N <= 1024) GRAPPA_ADB(4);
    else GRAPPA_ADB(8);
#undef GRAPPA_ADB
    return grappa_launch_status();
}
// ---- batched row-wise kernels (C ABI 8): the same kernel of up to GRAPPA_ROW_BATCH_MAX independent tensors in one launch
namespace {
template <typename Item>
bool row_items_ok(const Item* items, int count) { return items && count > 0 && count <= GRAPPA_ROW_BATCH_MAX; }
inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }
}  // namespace

extern "C" int grappa_layernorm_fwd_batched_f32(void* stream, const grappa_ln_fwd_item* items, int count) {
    if (!row_items_ok(items, count)) return GRAPPA_ERR_ARG;
    LnFwdBatch b;
    b.count = count;
    b.blk_begin[0] = 0;
    int wmax = 0;
    for (int i = 0; i < count; ++i) {
        const grappa_ln_fwd_item& t = items[i];
        if (t.M < 0 || t.W <= 0 || (t.W & 3) || t.W > 2048 || (t.ldx & 3) || (t.ldy & 3)) return GRAPPA_ERR_ARG;
        if (t.M > 0 && (!t.x || !t.gamma || !t.beta || !t.y || !t.mean || !t.rstd || !al16(t.x) || !al16(t.y) || !al16(t.gamma) || !al16(t.beta))) return GRAPPA_ERR_ARG;
        b.it[i] = t;
        const int blocks = t.W <= 1024 ? ((t.M + 3) / 4 > 2048 ? 2048 : (t.M + 3) / 4) : ln_blocks(t.M);
        b.blk_begin[i + 1] = b.blk_begin[i] + (t.M > 0 ? blocks : 0);
        wmax = t.W > wmax ? t.W : wmax;
    }
    for (int i = count; i < GRAPPA_ROW_BATCH_MAX; ++i) b.blk_begin[i + 1] = b.blk_begin[count];
    if (b.blk_begin[count] == 0) return GRAPPA_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define GRAPPA_LNB(NCH) GRAPPA_LAUNCH((layernorm_fwd_batched_kernel<NCH>), dim3(b.blk_begin[count]), dim3(256), 0, st, b)
    if (wmax <= 256) GRAPPA_LNB(1);
    else if (wmax <= 512) GRAPPA_LNB(2);
    else if (wmax <= 1024) GRAPPA_LNB(4);
    else GRAPPA_LNB(8);
#undef GRAPPA_LNB
    return grappa_launch_status();
}

extern "C" int grappa_layernorm_bwd_batched_f32(void* stream, const grappa_ln_bwd_item* items, int count) {
    if (!row_items_ok(items, count)) return GRAPPA_ERR_ARG;
    LnBwdBatch b;
    b.count = count;
    b.blk_begin[0] = 0;
    int wmax = 0;
    for (int i = 0; i < count; ++i) {
        const grappa_ln_bwd_item& t = items[i];
        if (t.M < 0 || t.W <= 0 || (t.W & 3) || t.W > 2048 || (t.ldx & 3) || (t.lddy & 3) || (t.lddx & 3)) return GRAPPA_ERR_ARG;
        if (t.M > 0 && (!t.dy || !t.x || !t.mean || !t.rstd || !t.gamma || !t.dx || !t.part || !al16(t.dy) || !al16(t.x) || !al16(t.dx) || !al16(t.gamma)))
            return GRAPPA_ERR_ARG;
        b.it[i] = t;
        b.blk_begin[i + 1] = b.blk_begin[i] + (t.M > 0 ? ln_blocks(t.M) : 0);          // = grappa_layernorm_bwd_partial_rows(M) rows of partials
        wmax = t.W > wmax ? t.W : wmax;
    }
    for (int i = count; i < GRAPPA_ROW_BATCH_MAX; ++i) b.blk_begin[i + 1] = b.blk_begin[count];
    if (b.blk_begin[count] == 0) return GRAPPA_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t smem = (size_t)4 * 2 * wmax * sizeof(float);
#define GRAPPA_LNBB(NCH) GRAPPA_LAUNCH((layernorm_bwd_batched_kernel<NCH>), dim3(b.blk_begin[count]), dim3(256), smem, st, b)
    if (wmax <= 256) GRAPPA_LNBB(1);
    else if (wmax <= 512) GRAPPA_LNBB(2);
    else if (wmax <= 1024) GRAPPA_LNBB(4);
    else GRAPPA_LNBB(8);
#undef GRAPPA_LNBB
    return grappa_launch_status();
}

extern "C" int grappa_act_dropout_bwd_batched_f32(void* stream, const grappa_act_dropout_item* items, int count) {
    if (!row_items_ok(items, count)) return GRAPPA_ERR_ARG;
    ActDropBatch b;
    b.count = count;
    b.blk_begin[0] = 0;
    int nmax = 0;
    for (int i = 0; i < count; ++i) {
        const grappa_act_dropout_item& t = items[i];
        if (t.M < 0 || t.N <= 0 || (t.N & 3) || t.N > 2048 || (t.lddy & 3) || (t.lddz & 3) || (t.y && (t.ldy & 3)) || t.drop_p < 0.f || t.drop_p >= 1.f) return GRAPPA_ERR_ARG;
        if (t.M > 0 && (!t.dy || !t.dz || !t.dz_amax || !al16(t.dy) || !al16(t.dz) || !al16(t.y))) return GRAPPA_ERR_ARG;
        b.it[i] = t;
        b.blk_begin[i + 1] = b.blk_begin[i] + (t.M > 0 ? ((t.M + 3) / 4 > 2048 ? 2048 : (t.M + 3) / 4) : 0);
        nmax = t.N > nmax ? t.N : nmax;
    }
    for (int i = count; i < GRAPPA_ROW_BATCH_MAX; ++i) b.blk_begin[i + 1] = b.blk_begin[count];
    if (b.blk_begin[count] == 0) return GRAPPA_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define GRAPPA_ADBB(NCH) GRAPPA_LAUNCH((act_dropout_bwd_batched_kernel<NCH>), dim3(b.blk_begin[count]), dim3(256), 0, st, b)
    if (nmax <= 256) GRAPPA_ADBB(1);
    else if (nmax <= 512) GRAPPA_ADBB(2);
    else if (nmax <= 1024) GRAPPA_ADBB(4);
    else GRAPPA_ADBB(8);
#undef GRAPPA_ADBB
    return grappa_launch_status();
}

extern "C" int grappa_act_dropout_bwd_pairs_f32(void* stream, int M, int N, const float* dy, int lddy, const float* y, int ldy, float drop_p,
                                                uint64_t drop_seed, float* dz, int lddz, uint32_t* dz_amax, uint16_t* pairs, int ldp, const uint64_t* drop_salt) {
    if (M < 0 || N <= 0 || (N & 31) || N > 2048 || drop_p < 0.f || drop_p >= 1.f || (lddy & 3) || (dz && (lddz & 3)) || (y && (ldy & 3)) || ldp < 2 * N || (ldp & 7))
        return GRAPPA_ERR_ARG;
    if (M == 0) return GRAPPA_OK;
    if (!dy || !dz_amax || !pairs) return GRAPPA_ERR_ARG;
    if (((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dz) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(pairs)) & 15) != 0)
        return GRAPPA_ERR_ARG;
    const float scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int blocks = (M + 3) / 4 > 2048 ? 2048 : (M + 3) / 4;
#define GRAPPA_ADBP(NCH) GRAPPA_LAUNCH((act_dropout_bwd_pairs_kernel<NCH>), dim3(blocks), dim3(256), 0, st, M, N, dy, lddy, y, ldy, drop_p, scale, drop_seed, drop_salt, dz, lddz, dz_amax, pairs, ldp)
    if (N <= 256) GRAPPA_ADBP(1);
    else if (N <= 512) GRAPPA_ADBP(2);
    else if (N <= 1024) GRAPPA_ADBP(4);
    else GRAPPA_ADBP(8);
#undef GRAPPA_ADBP
    return grappa_launch_status();
}
extern "C" int grappa_act_dropout_bwd_bf16(void* stream, int M, int N, const uint16_t* dy, int lddy, const uint16_t* y, int ldy,
                                           float drop_p, uint64_t drop_seed, uint16_t* dz, int lddz, const uint64_t* drop_salt) {
    return act_dropout_bwd_impl<grappa_bf16_t>(stream, M, N, dy, lddy, y, ldy, drop_p, drop_seed, dz, lddz, drop_salt);
}

extern "C" int grappa_add_f32(void* stream, size_t n, const float* x, const float* z, float* y) {
    if (n == 0) return GRAPPA_OK;
    if (!x || !z || !y) return GRAPPA_ERR_ARG;
    GRAPPA_LAUNCH(add_kernel, dim3(grid_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), n, x, z, y);
    return grappa_launch_status();
}

extern "C" size_t grappa_sumsq_workspace_bytes(size_t n) { return (size_t)grid_for(n, 1024) * sizeof(float); }

extern "C" int grappa_sumsq_f32(void* stream, size_t n, const float* x, float* out, int accumulate, void* ws, size_t ws_bytes) {
    if (!out) return GRAPPA_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int blocks = grid_for(n, 1024);
    if (!ws || ws_bytes < (size_t)blocks * sizeof(float)) return GRAPPA_ERR_WORKSPACE;
    if (n > 0 && !x) return GRAPPA_ERR_ARG;
    float* part = reinterpret_cast<float*>(ws);
    GRAPPA_LAUNCH(sumsq_partial_kernel, dim3(blocks), dim3(256), 0, st, n, x, part);
    GRAPPA_LAUNCH(sumsq_final_kernel, dim3(1), dim3(64), 0, st, blocks, part, out, accumulate);
    return grappa_launch_status();
}

extern "C" int grappa_adam_step_f32(void* stream, size_t n, float* p, const float* g, float* m, float* v, float lr, float beta1,
                                    float beta2, float eps, float weight_decay, int step, float grad_scale, const float* sumsq,
                                    float max_norm) {
    if (n == 0) return GRAPPA_OK;
    if (!p || !g || !m || !v || step < 1) return GRAPPA_ERR_ARG;
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    GRAPPA_LAUNCH(adam_kernel, dim3(grid_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), n, p, g, m, v, lr, beta1,
                       beta2, eps, weight_decay, bc1, bc2, grad_scale, sumsq, max_norm);
    return grappa_launch_status();
}

extern "C" int grappa_adam_step_dyn_f32(void* stream, size_t n, float* p, const float* g, float* m, float* v, const float* lr_dev, float beta1,
                                        float beta2, float eps, float weight_decay, const int* step_dev, float grad_scale, const float* sumsq, float max_norm) {
    if (n == 0) return GRAPPA_OK;
    if (!p || !g || !m || !v || !lr_dev || !step_dev) return GRAPPA_ERR_ARG;
    GRAPPA_LAUNCH(adam_dyn_kernel, dim3(grid_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), n, p, g, m, v, lr_dev, beta1, beta2, eps,
                  weight_decay, step_dev, grad_scale, sumsq, max_norm);
    return grappa_launch_status();
}

// dropout salt (include/grappa_hip.h): one 64-bit word in device memory mixed into every dropout seed of the kernels launched afterwards

extern "C" int grappa_charge_encoding_f32(void* stream, int N, const float* q, int dim, float lo, float hi, float* out, int ldo, int col0) {
    if (N < 0 || dim <= 0 || (dim & 1) || hi <= lo) return GRAPPA_ERR_ARG;
    if (N == 0) return GRAPPA_OK;
    if (!q || !out) return GRAPPA_ERR_ARG;
    GRAPPA_LAUNCH(charge_encoding_kernel, dim3(grid_for((size_t)N * dim / 2)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       N, q, dim, lo, hi, out, ldo, col0);
    return grappa_launch_status();
}

extern "C" int grappa_dropout_keep(uint64_t seed, uint64_t index, float p) { return grappa_keep(seed, index, p) ? 1 : 0; }
extern "C" int grappa_abi_version(void) { return GRAPPA_ABI_VERSION; }
extern "C" const char* grappa_build_arch(void) { return "gfx950"; }

#include <atomic>
static std::atomic<long long> g_launches{0};
extern "C" void grappa_count_launch(void) { g_launches.fetch_add(1, std::memory_order_relaxed); }
extern "C" long long grappa_launch_count(int reset) {
    return reset ? g_launches.exchange(0, std::memory_order_relaxed) : g_launches.load(std::memory_order_relaxed);
}
