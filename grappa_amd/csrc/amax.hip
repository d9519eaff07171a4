// Largest magnitude of every row and every column of an fp32 matrix, one pass: the power-of-two scales of the F32_F16X3 products
// (gemm_bf16x_impl.h, mode H3).  Magnitudes travel as fp32 bit patterns with the sign cleared: unsigned integer order is then the
// order of the magnitudes (NaN above Inf above every finite value), so the whole reduction is integer max -- exact, order-free.
// HBM-bound: R * C * 4 bytes read once; a wavefront walks rows (64 lanes x 16 B = 256 columns per load), four rows in flight.
#include "common.h"

namespace {

constexpr int AMAX_THREADS = 256;         // 4 wavefronts, one row each
constexpr int AMAX_CBLOCK = 2048;         // columns per grid.y slice (8 loads of 256 columns)
constexpr int AMAX_MAX_BLOCKS = 512;      // two workgroups per CU
constexpr int AMAX_ROWS_IN_FLIGHT = 4;

__device__ inline unsigned umax4(const uint4& v) { return max(max(v.x, v.y), max(v.z, v.w)); }

template <int NCH, bool COLS, bool VEC>
__global__ __launch_bounds__(AMAX_THREADS) void amax_kernel(const float* __restrict__ x, int R, int C, int ld, unsigned* __restrict__ row_amax,
                                                            unsigned* __restrict__ col_part, int part_stride, int row_atomic) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int cb = blockIdx.y * AMAX_CBLOCK;
    const int nw = gridDim.x * (AMAX_THREADS / 64);
    uint4 cm[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) cm[ch] = make_uint4(0u, 0u, 0u, 0u);
    for (int r0 = blockIdx.x * (AMAX_THREADS / 64) + wave; r0 < R; r0 += nw * AMAX_ROWS_IN_FLIGHT) {
        uint4 v[AMAX_ROWS_IN_FLIGHT][NCH];
#pragma unroll
        for (int u = 0; u < AMAX_ROWS_IN_FLIGHT; ++u) {
            const int r = r0 + u * nw;
            const unsigned* row = reinterpret_cast<const unsigned*>(x) + (size_t)min(r, R - 1) * ld;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int c = cb + ch * 256 + lane * 4;
                uint4 t = make_uint4(0u, 0u, 0u, 0u);
                if (r < R) {
                    if (VEC) {
                        if (c < C) t = *reinterpret_cast<const uint4*>(row + c);              // C % 4 == 0: c < C covers c + 3
                    } else {
                        if (c < C) t.x = row[c];
                        if (c + 1 < C) t.y = row[c + 1];
                        if (c + 2 < C) t.z = row[c + 2];
                        if (c + 3 < C) t.w = row[c + 3];
                    }
                }
                v[u][ch] = make_uint4(t.x & 0x7fffffffu, t.y & 0x7fffffffu, t.z & 0x7fffffffu, t.w & 0x7fffffffu);
            }
        }
#pragma unroll
        for (int u = 0; u < AMAX_ROWS_IN_FLIGHT; ++u) {
            unsigned rm = 0u;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                rm = max(rm, umax4(v[u][ch]));
                if (COLS) cm[ch] = make_uint4(max(cm[ch].x, v[u][ch].x), max(cm[ch].y, v[u][ch].y), max(cm[ch].z, v[u][ch].z), max(cm[ch].w, v[u][ch].w));
            }
            if (row_amax) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) rm = max(rm, (unsigned)__shfl_xor((int)rm, o, 64));
                const int r = r0 + u * nw;
                if (lane == 0 && r < R) {
                    if (row_atomic) atomicMax(row_amax + r, rm);
                    else row_amax[r] = rm;
                }
            }
        }
    }
    if (COLS) {
        __shared__ uint4 sm[AMAX_THREADS / 64][NCH * 64];
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) sm[wave][ch * 64 + lane] = cm[ch];
        __syncthreads();
        for (int q = threadIdx.x; q < NCH * 64; q += AMAX_THREADS) {
            uint4 m = sm[0][q];
#pragma unroll
            for (int w = 1; w < AMAX_THREADS / 64; ++w) {
                const uint4 t = sm[w][q];
                m = make_uint4(max(m.x, t.x), max(m.y, t.y), max(m.z, t.z), max(m.w, t.w));
            }
            const int c = cb + q * 4;                       // part_stride is a multiple of 4 and >= round_up(C, 4)
            if (c < part_stride) *reinterpret_cast<uint4*>(col_part + (size_t)blockIdx.x * part_stride + c) = m;
        }
    }
}

// column maxima of the per-workgroup partials: 64 columns per workgroup, 16 groups of lanes share the partial rows
__global__ __launch_bounds__(1024) void amax_colreduce_kernel(const unsigned* __restrict__ col_part, int nblocks, int part_stride, int C,
                                                              unsigned* __restrict__ col_amax) {
    __shared__ unsigned sm[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    unsigned m = 0u;
    if (c < C) {
#pragma unroll 8
        for (int b = ty; b < nblocks; b += 16) m = max(m, col_part[(size_t)b * part_stride + c]);
    }
    sm[ty][tx] = m;
    __syncthreads();
    if (ty == 0 && c < C) {
#pragma unroll
        for (int g = 1; g < 16; ++g) m = max(m, sm[g][tx]);
        col_amax[c] = m;
    }
}

// out[b] = max_i in_b[i] for up to 32 arrays, one workgroup each (the whole-tensor maxima of the operands of a group of weight-gradient
// products, from their row maxima)
constexpr int AMAX_BATCH = 32;
struct AmaxBatch {
    const unsigned* in[AMAX_BATCH];
    int n[AMAX_BATCH];
};
__global__ __launch_bounds__(1024) void amax_reduce_kernel(AmaxBatch b, unsigned* __restrict__ out) {
    __shared__ unsigned sm[16];
    const unsigned* __restrict__ in = b.in[blockIdx.x];
    const int n = b.n[blockIdx.x];
    unsigned m = 0u;
    if ((reinterpret_cast<uintptr_t>(in) & 15) == 0) {
        for (int i = threadIdx.x * 4; i < n; i += 4096) {
            if (i + 3 < n) m = max(m, umax4(*reinterpret_cast<const uint4*>(in + i)));
            else
                for (int q = i; q < n; ++q) m = max(m, in[q]);
        }
    } else {
        for (int i = threadIdx.x; i < n; i += 1024) m = max(m, in[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < 16; ++w) m = max(m, sm[w]);
        out[blockIdx.x] = m;
    }
}

// out[m] = max over segments of part[seg * M + m]: the per-segment row maxima the product kernels leave behind (gemm_common.h)
__global__ __launch_bounds__(256) void amax_combine_kernel(int M, int nseg, const unsigned* __restrict__ part, unsigned* __restrict__ out) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    unsigned v = 0u;
#pragma unroll 8
    for (int s = 0; s < nseg; ++s) v = max(v, part[(size_t)s * M + m]);
    out[m] = v;
}

// the same for up to four products of one grouped launch (grappa_gemm_f32_group) in one launch
struct AmaxCombine4 {
    const unsigned* part[4];
    unsigned* out[4];
    int M[4], nseg[4], blk_begin[5], count;
};
__global__ __launch_bounds__(256) void amax_combine4_kernel(AmaxCombine4 a) {
    int i = 0;
    while (i + 1 < a.count && (int)blockIdx.x >= a.blk_begin[i + 1]) ++i;
    const int m = ((int)blockIdx.x - a.blk_begin[i]) * 256 + threadIdx.x;
    if (m >= a.M[i]) return;
    unsigned v = 0u;
#pragma unroll 8
    for (int s = 0; s < a.nseg[i]; ++s) v = max(v, a.part[i][(size_t)s * a.M[i] + m]);
    a.out[i][m] = v;
}

// BATCH_SLICES workgroups per matrix (weights: at most a few MB each; one workgroup per matrix left 100 - 200 of 256 CUs idle for 0.2 ms
// per optimiser step), each a share of the rows; column maxima meet in LDS (ds_max_u32), then across the slices in the output array by
// atomic maxima of the bit patterns (order-free: the same bits every run), which amax_batched_zero_kernel cleared
constexpr int BATCH_SLICES = 8;
__global__ __launch_bounds__(256) void amax_batched_zero_kernel(const grappa_amax_item* __restrict__ descs) {
    const grappa_amax_item d = descs[blockIdx.x];
    for (int c = threadIdx.x; c < d.C; c += 256) d.col_amax[c] = 0u;
}
__global__ __launch_bounds__(1024) void amax_batched_kernel(const grappa_amax_item* __restrict__ descs) {
    __shared__ unsigned colmax[2048];
    const grappa_amax_item d = descs[blockIdx.x];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if ((int)blockIdx.y * 16 >= d.R) return;                  // (a slice without rows: nothing to add)
    for (int c = threadIdx.x; c < 2048; c += 1024) colmax[c] = 0u;
    __syncthreads();
    uint4 cm[8];
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) cm[ch] = make_uint4(0u, 0u, 0u, 0u);
    for (int r = blockIdx.y * 16 + wave; r < d.R; r += 16 * BATCH_SLICES) {
        const unsigned* row = reinterpret_cast<const unsigned*>(d.x) + (size_t)r * d.ld;
        unsigned rm = 0u;
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) {
            const int c = ch * 256 + lane * 4;
            if (c < d.C) {
                const uint4 t = *reinterpret_cast<const uint4*>(row + c);
                const uint4 v = make_uint4(t.x & 0x7fffffffu, t.y & 0x7fffffffu, t.z & 0x7fffffffu, t.w & 0x7fffffffu);
                rm = max(rm, umax4(v));
                cm[ch] = make_uint4(max(cm[ch].x, v.x), max(cm[ch].y, v.y), max(cm[ch].z, v.z), max(cm[ch].w, v.w));
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) rm = max(rm, (unsigned)__shfl_xor((int)rm, o, 64));
        if (lane == 0) d.row_amax[r] = rm;
    }
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
        const int c = ch * 256 + lane * 4;
        if (c < d.C) {
            atomicMax(&colmax[c], cm[ch].x);
            atomicMax(&colmax[c + 1], cm[ch].y);
            atomicMax(&colmax[c + 2], cm[ch].z);
            atomicMax(&colmax[c + 3], cm[ch].w);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < d.C; c += 1024) atomicMax(&d.col_amax[c], colmax[c]);
}

int amax_blocks(int R) {
    const int per = (AMAX_THREADS / 64) * AMAX_ROWS_IN_FLIGHT;
    const int b = (R + per - 1) / per;
    return b < 1 ? 1 : (b > AMAX_MAX_BLOCKS ? AMAX_MAX_BLOCKS : b);
}
int part_stride_of(int C) { return (C + 3) / 4 * 4; }

template <int NCH>
void launch_amax(hipStream_t st, dim3 grid, bool cols, bool vec, const float* x, int R, int C, int ld, unsigned* row_amax, unsigned* part, int stride, int row_atomic) {
    if (cols) {
        if (vec) GRAPPA_LAUNCH((amax_kernel<NCH, true, true>), grid, dim3(AMAX_THREADS), 0, st, x, R, C, ld, row_amax, part, stride, row_atomic);
        else GRAPPA_LAUNCH((amax_kernel<NCH, true, false>), grid, dim3(AMAX_THREADS), 0, st, x, R, C, ld, row_amax, part, stride, row_atomic);
    } else {
        if (vec) GRAPPA_LAUNCH((amax_kernel<NCH, false, true>), grid, dim3(AMAX_THREADS), 0, st, x, R, C, ld, row_amax, part, stride, row_atomic);
        else GRAPPA_LAUNCH((amax_kernel<NCH, false, false>), grid, dim3(AMAX_THREADS), 0, st, x, R, C, ld, row_amax, part, stride, row_atomic);
    }
}

}  // namespace

extern "C" size_t grappa_amax_f32_workspace_bytes(int R, int C) {
    if (R <= 0 || C <= 0) return 0;
    return (size_t)amax_blocks(R) * part_stride_of(C) * sizeof(unsigned);
}

extern "C" int grappa_amax_f32(void* stream, int R, int C, const float* x, int ldx, uint32_t* row_amax, uint32_t* col_amax, void* ws, size_t ws_bytes) {
    if (R < 0 || C < 0 || (!row_amax && !col_amax)) return GRAPPA_ERR_ARG;
    if (R == 0 || C == 0) return GRAPPA_OK;
    if (!x || ldx < C) return GRAPPA_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nb = amax_blocks(R), stride = part_stride_of(C);
    if (col_amax && (!ws || ws_bytes < (size_t)nb * stride * sizeof(unsigned))) return GRAPPA_ERR_WORKSPACE;
    const int ny = (C + AMAX_CBLOCK - 1) / AMAX_CBLOCK;
    const int row_atomic = row_amax && ny > 1;
    if (row_atomic && hipMemsetAsync(row_amax, 0, (size_t)R * sizeof(unsigned), st) != hipSuccess) return GRAPPA_ERR_LAUNCH;
    const bool vec = (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (ldx & 3) == 0 && (C & 3) == 0;
    const int cols_here = C < AMAX_CBLOCK ? C : AMAX_CBLOCK;
    const int nch = (cols_here + 255) / 256;
    unsigned* part = reinterpret_cast<unsigned*>(ws);
    const dim3 grid(nb, ny);
    if (nch <= 1) launch_amax<1>(st, grid, col_amax != nullptr, vec, x, R, C, ldx, row_amax, part, stride, row_atomic);
    else if (nch <= 2) launch_amax<2>(st, grid, col_amax != nullptr, vec, x, R, C, ldx, row_amax, part, stride, row_atomic);
    else if (nch <= 4) launch_amax<4>(st, grid, col_amax != nullptr, vec, x, R, C, ldx, row_amax, part, stride, row_atomic);
    else if (nch <= 6) launch_amax<6>(st, grid, col_amax != nullptr, vec, x, R, C, ldx, row_amax, part, stride, row_atomic);
    else launch_amax<8>(st, grid, col_amax != nullptr, vec, x, R, C, ldx, row_amax, part, stride, row_atomic);
    if (grappa_launch_status() != GRAPPA_OK) return GRAPPA_ERR_LAUNCH;
    if (col_amax) {
        GRAPPA_LAUNCH(amax_colreduce_kernel, dim3((C + 63) / 64), dim3(1024), 0, st, part, nb, stride, C, col_amax);
        return grappa_launch_status();
    }
    return GRAPPA_OK;
}

extern "C" int grappa_amax_reduce(void* stream, int count, const uint32_t* const* in, const int* n, uint32_t* out) {
    if (count < 0 || (count > 0 && (!in || !n || !out))) return GRAPPA_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    for (int b0 = 0; b0 < count; b0 += AMAX_BATCH) {
        AmaxBatch b;
        const int c = count - b0 < AMAX_BATCH ? count - b0 : AMAX_BATCH;
        for (int i = 0; i < c; ++i) {
            if (!in[b0 + i] || n[b0 + i] <= 0) return GRAPPA_ERR_ARG;
            b.in[i] = in[b0 + i];
            b.n[i] = n[b0 + i];
        }
        GRAPPA_LAUNCH(amax_reduce_kernel, dim3(c), dim3(1024), 0, st, b, out + b0);
    }
    return grappa_launch_status();
}

// called by grappa_gemm_f32 (gemm_f32.hip) after a product that wrote per-segment row maxima
int grappa_launch_amax_combine(hipStream_t st, int M, int nseg, const unsigned* part, unsigned* out) {
    GRAPPA_LAUNCH(amax_combine_kernel, dim3((M + 255) / 256), dim3(256), 0, st, M, nseg, part, out);
    return grappa_launch_status();
}

int grappa_launch_amax_combine4(hipStream_t st, int count, const int* M, const int* nseg, const unsigned* const* part, unsigned* const* out) {
    AmaxCombine4 a;
    a.count = count;
    a.blk_begin[0] = 0;
    for (int i = 0; i < 4; ++i) {
        a.M[i] = i < count ? M[i] : 0;
        a.nseg[i] = i < count ? nseg[i] : 0;
        a.part[i] = i < count ? part[i] : nullptr;
        a.out[i] = i < count ? out[i] : nullptr;
        a.blk_begin[i + 1] = a.blk_begin[i] + (i < count ? (M[i] + 255) / 256 : 0);
    }
    if (a.blk_begin[4] == 0) return GRAPPA_OK;
    GRAPPA_LAUNCH(amax_combine4_kernel, dim3(a.blk_begin[4]), dim3(256), 0, st, a);
    return grappa_launch_status();
}

extern "C" int grappa_amax_combine(void* stream, int M, int nseg, const uint32_t* parts, uint32_t* out) {
    if (M < 0 || nseg <= 0) return GRAPPA_ERR_ARG;
    if (M == 0) return GRAPPA_OK;
    if (!parts || !out) return GRAPPA_ERR_ARG;
    return grappa_launch_amax_combine(reinterpret_cast<hipStream_t>(stream), M, nseg, parts, out);
}

extern "C" int grappa_amax_f32_batched(void* stream, int count, const grappa_amax_item* descs) {
    if (count < 0 || (count > 0 && !descs)) return GRAPPA_ERR_ARG;
    if (count == 0) return GRAPPA_OK;
    GRAPPA_LAUNCH(amax_batched_zero_kernel, dim3(count), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), descs);
    GRAPPA_LAUNCH(amax_batched_kernel, dim3(count, BATCH_SLICES), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), descs);
    return grappa_launch_status();
}
