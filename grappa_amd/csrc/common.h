// Shared device helpers for libgrappa_hip (gfx950 / CDNA4 only: 64-wide wavefronts).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/grappa_hip.h"

#define GRAPPA_WAVE 64

// every kernel launch of the library goes through this macro: the process-wide counter behind grappa_launch_count() (include/grappa_hip.h)
// is what bench.py reports as launches per step
extern "C" void grappa_count_launch(void);
#define GRAPPA_LAUNCH(...)               \
    do {                                 \
        grappa_count_launch();           \
        hipLaunchKernelGGL(__VA_ARGS__); \
    } while (0)

static inline int grappa_launch_status() {
    return hipGetLastError() == hipSuccess ? GRAPPA_OK : GRAPPA_ERR_LAUNCH;
}

// counter-based dropout decision: a 32-bit finaliser (murmur3 fmix32) of the element index, keyed by the 64-bit seed;
// keep iff u >= p with u = 24 random bits / 2^24.  ~10 integer ops per element (it runs in GEMM epilogues).
__host__ __device__ inline uint32_t grappa_hash32(uint64_t seed, uint64_t idx) {
    uint32_t h = (uint32_t)idx * 0x9E3779B1u + (uint32_t)seed;
    h ^= (uint32_t)(idx >> 32) * 0x85EBCA77u;
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    h += (uint32_t)(seed >> 32);
    h ^= h >> 15;
    h *= 0x2C1B3C6Du;
    h ^= h >> 12;
    return h >> 8;   // 24 random bits
}
__host__ __device__ inline bool grappa_keep(uint64_t seed, uint64_t idx, float p) {
    return (float)grappa_hash32(seed, idx) * (1.0f / 16777216.0f) >= p;
}

// dropout salt: a 64-bit word in device memory mixed into the seeds (the drop_salt argument of every entry point that draws a mask, C ABI
// 10): the masks of a captured step then change from replay to replay although every seed is a constant of the graph.  nullptr: the seed as given.
__device__ inline uint64_t grappa_salted(uint64_t seed, const uint64_t* __restrict__ salt) {
    return salt ? seed + salt[0] * 0x9E3779B97F4A7C15ull : seed;
}

__device__ inline float grappa_elu(float x) { return x > 0.0f ? x : expm1f(x); }
// derivative of ELU expressed through its OUTPUT y: y>0 -> 1 ; else exp(z) = y+1
__device__ inline float grappa_elu_grad_from_out(float y) { return y > 0.0f ? 1.0f : y + 1.0f; }
__device__ inline float grappa_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---- element types of activation tensors: fp32, or bf16 storage (uint16_t bit patterns; the "bf16" configuration keeps every
// activation and activation gradient in bf16 in HBM; arithmetic, statistics and accumulation stay fp32).  ld4 / st4 move four
// consecutive elements (chunk c of a row whose base is 16-byte (fp32) / 8-byte (bf16) aligned).
typedef uint16_t grappa_bf16_t;
__device__ inline float4 ld4(const float* __restrict__ row, int c) { return reinterpret_cast<const float4*>(row)[c]; }
__device__ inline float4 ld4(const grappa_bf16_t* __restrict__ row, int c) {
    const uint2 u = reinterpret_cast<const uint2*>(row)[c];
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
__device__ inline void st4(float* __restrict__ row, int c, const float4& v) { reinterpret_cast<float4*>(row)[c] = v; }
__device__ inline void st4(grappa_bf16_t* __restrict__ row, int c, const float4& v) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    bf2 a, b;
    a[0] = (__bf16)v.x; a[1] = (__bf16)v.y;               // round to nearest even (v_cvt_pk_bf16_f32)
    b[0] = (__bf16)v.z; b[1] = (__bf16)v.w;
    reinterpret_cast<uint2*>(row)[c] = make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
}
// one element of LayerNorm's output -- ONE expression for the kernels that write it and the product epilogue that recomputes it as a
// residual (grappa_gemm_desc.res_ln_*): the same bits everywhere
__device__ inline float grappa_ln_apply(float x, float mean, float rstd, float g, float b) { return __builtin_fmaf((x - mean) * rstd, g, b); }

// ---- the PAIR format (include/grappa_hip.h, ABI 5): an fp32 row as fp16 (HI, LO) halves scaled by 2^shift, shift = 141 - the exponent
// field of the row's largest magnitude; blocks of 16 k as [16 x HI | 16 x LO].  st_pairs4: elements k = 4c .. 4c + 3 of one row.
__device__ inline int grappa_amax_shift(unsigned bits) { return 141 - (int)((bits >> 23) & 0xffu); }
__device__ inline void st_pairs4(uint16_t* __restrict__ row, int c, const float4& v, int shift) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const float r0 = __builtin_ldexpf(v.x, shift), r1 = __builtin_ldexpf(v.y, shift), r2 = __builtin_ldexpf(v.z, shift), r3 = __builtin_ldexpf(v.w, shift);
    h2 h01, h23, l01, l23;
    h01[0] = (_Float16)r0; h01[1] = (_Float16)r1; h23[0] = (_Float16)r2; h23[1] = (_Float16)r3;     // round to nearest even; |r| < 2^15
    l01[0] = (_Float16)(r0 - (float)h01[0]); l01[1] = (_Float16)(r1 - (float)h01[1]);
    l23[0] = (_Float16)(r2 - (float)h23[0]); l23[1] = (_Float16)(r3 - (float)h23[1]);
    uint16_t* p = row + 32 * (c >> 2) + 4 * (c & 3);
    *reinterpret_cast<uint2*>(p) = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
    *reinterpret_cast<uint2*>(p + 16) = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
}
// The same for a wavefront whose lanes hold CONSECUTIVE c (lane l: c = c0 + l, c0 even) and are all active up to an even count: lanes
// 2j and 2j + 1 trade halves so that each issues ONE 16-byte store (the even lane both HI quads, the odd lane both LO quads) instead of
// two 8-byte ones.  `ok` = this lane holds an element (lanes beyond the row end must still execute the exchange).
__device__ inline void st_pairs4_paired(uint16_t* __restrict__ row, int c, const float4& v, int shift, bool ok) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const float r0 = __builtin_ldexpf(v.x, shift), r1 = __builtin_ldexpf(v.y, shift), r2 = __builtin_ldexpf(v.z, shift), r3 = __builtin_ldexpf(v.w, shift);
    h2 h01, h23, l01, l23;
    h01[0] = (_Float16)r0; h01[1] = (_Float16)r1; h23[0] = (_Float16)r2; h23[1] = (_Float16)r3;
    l01[0] = (_Float16)(r0 - (float)h01[0]); l01[1] = (_Float16)(r1 - (float)h01[1]);
    l23[0] = (_Float16)(r2 - (float)h23[0]); l23[1] = (_Float16)(r3 - (float)h23[1]);
    const unsigned H0 = __builtin_bit_cast(unsigned, h01), H1 = __builtin_bit_cast(unsigned, h23);
    const unsigned L0 = __builtin_bit_cast(unsigned, l01), L1 = __builtin_bit_cast(unsigned, l23);
    const bool odd = (c & 1) != 0;
    const unsigned g0 = (unsigned)__shfl_xor((int)(odd ? H0 : L0), 1, 64), g1 = (unsigned)__shfl_xor((int)(odd ? H1 : L1), 1, 64);
    if (!ok) return;
    const int ce = c & ~1;                                       // the pair's even c: its quad starts the 16-byte run
    uint16_t* p = row + 32 * (ce >> 2) + 4 * (ce & 3) + (odd ? 16 : 0);
    *reinterpret_cast<uint4*>(p) = odd ? make_uint4(g0, g1, L0, L1) : make_uint4(H0, H1, g0, g1);
}
__device__ inline float ld1(const float* __restrict__ p, size_t i) { return p[i]; }
__device__ inline float ld1(const grappa_bf16_t* __restrict__ p, size_t i) { return __uint_as_float((unsigned)p[i] << 16); }
__device__ inline void st1(float* __restrict__ p, size_t i, float v) { p[i] = v; }
__device__ inline void st1(grappa_bf16_t* __restrict__ p, size_t i, float v) {
    const __bf16 h = (__bf16)v;
    p[i] = __builtin_bit_cast(grappa_bf16_t, h);
}

// A lane's chunk of E consecutive elements held as fp32.  E = 4: 16 B of fp32 or 8 B of bf16; E = 8 (bf16 only): 16 B per lane --
// the bf16 kernels are otherwise bound by the NUMBER of memory instructions, not by bytes (same launch time as fp32 with half the bytes)
template <int E> struct Chunk { float v[E]; };
template <int E> __device__ inline Chunk<E> chunk_zero() {
    Chunk<E> c;
#pragma unroll
    for (int e = 0; e < E; ++e) c.v[e] = 0.f;
    return c;
}
template <int E, typename T> __device__ inline Chunk<E> ldc(const T* __restrict__ row, int c);
template <> __device__ inline Chunk<4> ldc<4, float>(const float* __restrict__ row, int c) {
    const float4 f = reinterpret_cast<const float4*>(row)[c];
    return Chunk<4>{{f.x, f.y, f.z, f.w}};
}
template <> __device__ inline Chunk<4> ldc<4, grappa_bf16_t>(const grappa_bf16_t* __restrict__ row, int c) {
    const float4 f = ld4(row, c);
    return Chunk<4>{{f.x, f.y, f.z, f.w}};
}
template <> __device__ inline Chunk<8> ldc<8, grappa_bf16_t>(const grappa_bf16_t* __restrict__ row, int c) {
    const uint4 u = reinterpret_cast<const uint4*>(row)[c];
    return Chunk<8>{{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u),
                     __uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u), __uint_as_float(u.w << 16), __uint_as_float(u.w & 0xffff0000u)}};
}
template <int E, typename T> __device__ inline void stc(T* __restrict__ row, int c, const Chunk<E>& x);
template <> __device__ inline void stc<4, float>(float* __restrict__ row, int c, const Chunk<4>& x) {
    reinterpret_cast<float4*>(row)[c] = make_float4(x.v[0], x.v[1], x.v[2], x.v[3]);
}
template <> __device__ inline void stc<4, grappa_bf16_t>(grappa_bf16_t* __restrict__ row, int c, const Chunk<4>& x) {
    st4(row, c, make_float4(x.v[0], x.v[1], x.v[2], x.v[3]));
}
template <> __device__ inline void stc<8, grappa_bf16_t>(grappa_bf16_t* __restrict__ row, int c, const Chunk<8>& x) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    unsigned w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        bf2 h;
        h[0] = (__bf16)x.v[2 * e];
        h[1] = (__bf16)x.v[2 * e + 1];
        w[e] = __builtin_bit_cast(unsigned, h);
    }
    reinterpret_cast<uint4*>(row)[c] = make_uint4(w[0], w[1], w[2], w[3]);
}
template <int E> __device__ inline float cdot(const Chunk<E>& a, const Chunk<E>& b) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < E; e += 2) s += a.v[e] * b.v[e] + a.v[e + 1] * b.v[e + 1];
    return s;
}

__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// sum over aligned groups of `g` lanes (g power of two <= 64); every lane of the group gets the sum
__device__ inline float group_sum(float v, int g) {
    for (int o = g >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
