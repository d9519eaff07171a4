// Shared device helpers for libgrappa_hip (gfx950 / CDNA4 only: 64-wide wavefronts).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/grappa_hip.h"

#define GRAPPA_WAVE 64

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
