// Shared device helpers for libgrappa_hip (gfx950 / CDNA4 only: 64-wide wavefronts).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/grappa_hip.h"

#define GRAPPA_WAVE 64

static inline int grappa_launch_status() {
    return hipGetLastError() == hipSuccess ? GRAPPA_OK : GRAPPA_ERR_LAUNCH;
}

// counter-based dropout decision: splitmix64 of (seed + index*golden); keep iff u >= p, u in [0,1) with 24 bits
__host__ __device__ inline uint32_t grappa_hash32(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + idx * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z = z ^ (z >> 31);
    return (uint32_t)(z >> 40);   // 24 random bits
}
__host__ __device__ inline bool grappa_keep(uint64_t seed, uint64_t idx, float p) {
    return (float)grappa_hash32(seed, idx) * (1.0f / 16777216.0f) >= p;
}

__device__ inline float grappa_elu(float x) { return x > 0.0f ? x : expm1f(x); }
// derivative of ELU expressed through its OUTPUT y: y>0 -> 1 ; else exp(z) = y+1
__device__ inline float grappa_elu_grad_from_out(float y) { return y > 0.0f ? 1.0f : y + 1.0f; }
__device__ inline float grappa_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

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
