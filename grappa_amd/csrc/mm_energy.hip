// Molecular-mechanics energy, forces and their backward over index tuples (fp32).
//   energy   : one workgroup per molecule; a thread owns one conformation column and a strided share of the
//              molecule's tuples; per-level sums are combined through LDS in a fixed order (reproducible).
//   gradient : one thread per (atom, conformation); walks the atom's incidence list (atom -> (level, pos, tuple))
//              and adds coef * d(internal coordinate)/d(x_atom) in closed form -- a gather, no atomics.
//   backward : dL/dk, dL/deq for upstream gE (B,C) and gG (N,C,3): the closed form of the double backward the
//              reference obtains from autograd.grad(..., create_graph=True) (models/energy.py:139).  Forces are
//              linear in k and affine in eq, so with D = sum_a gG_a . dx/dx_a :
//                 harmonic: gk += gE*(x-eq)^2/2 + (x-eq)*D ;  geq += -k*(x-eq)*gE - k*D
//                 torsion : gk_n += gE*cos(n phi) - n sin(n phi) * D
// Geometry follows models/internal_coordinates.py:150-210 (distance, atan2 angle, timemachine dihedral) without
// the reference's random dihedral noise; conformations are the fast axis of xyz[N,C,3] so lanes read 12-byte
// strided but contiguous segments.
#include "common.h"

namespace {

struct V3 {
    float x, y, z;
};
__device__ inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ inline V3 ldv(const float* __restrict__ p, int atom, int C, int c) {
    const float* q = p + ((size_t)atom * C + c) * 3;
    return {q[0], q[1], q[2]};
}
constexpr float TINY = 1e-20f;

// bond: r = |x0 - x1| ; dr/dx0 = u, dr/dx1 = -u
__device__ inline float bond_geom(V3 p0, V3 p1, V3& u) {
    const V3 d = p0 - p1;
    const float r = sqrtf(dot(d, d));
    u = (1.0f / fmaxf(r, TINY)) * d;
    return r;
}
// angle at p1: theta = atan2(|u x v|, u.v), u = p0-p1, v = p2-p1 ; e0 = dtheta/dp0, e2 = dtheta/dp2, dtheta/dp1 = -(e0+e2)
__device__ inline float angle_geom(V3 p0, V3 p1, V3 p2, V3& e0, V3& e2) {
    const V3 u = p0 - p1, v = p2 - p1;
    const V3 w = cross(u, v);
    const float wl = sqrtf(dot(w, w));
    const float theta = atan2f(wl, dot(u, v));
    const float iw = 1.0f / fmaxf(wl, TINY);
    e0 = (iw / fmaxf(dot(u, u), TINY)) * cross(u, w);
    e2 = (iw / fmaxf(dot(v, v), TINY)) * cross(w, v);
    return theta;
}
// dihedral (reference convention): a = p1-p0, b = p1-p2, c = p3-p2, n1 = a x b, n2 = b x c,
// phi = atan2((n1 x n2).b/|b|, n1.n2);  d0 = -|b| n1/|n1|^2, d3 = |b| n2/|n2|^2,
// d1 = (p-1) d0 - q d3, d2 = (q-1) d3 - p d0 with p = a.b/|b|^2, q = c.b/|b|^2
__device__ inline float dihedral_geom(V3 p0, V3 p1, V3 p2, V3 p3, V3& d0, V3& d1, V3& d2, V3& d3) {
    const V3 a = p1 - p0, b = p1 - p2, c = p3 - p2;
    const V3 n1 = cross(a, b), n2 = cross(b, c);
    const float b2 = dot(b, b);
    const float bl = sqrtf(b2);
    const float y = dot(cross(n1, n2), b) / fmaxf(bl, TINY);
    const float x = dot(n1, n2);
    const float phi = atan2f(y, x);
    d0 = (-bl / fmaxf(dot(n1, n1), TINY)) * n1;
    d3 = (bl / fmaxf(dot(n2, n2), TINY)) * n2;
    const float ib2 = 1.0f / fmaxf(b2, TINY);
    const float p = dot(a, b) * ib2, q = dot(c, b) * ib2;
    d1 = (p - 1.0f) * d0 - q * d3;
    d2 = (q - 1.0f) * d3 - p * d0;
    return phi;
}

__device__ inline float torsion_energy(const float* __restrict__ k, int n_per, float phi, int offset) {
    float e = 0.f;
    for (int n = 1; n <= n_per; ++n) {
        const float kn = k[n - 1];
        e += kn * cosf((float)n * phi);
        if (offset) e += fabsf(kn);
    }
    return e;
}

struct MMArgs {
    grappa_mm_desc d;
    float* energy;
    float* term_energy;
    float* tuple_e[4];
    float* tuple_x[4];
};

// ------------------------------------------------------------------------------------------------ energy
// one workgroup per molecule; 1024 threads = (tuple slot, conformation): a 30-atom molecule's ~80 torsions take 3 rounds instead of the 10 of
// a 256-thread workgroup (the kernel is a chain of dependent rounds: 74 -> 30 us on a 32-molecule batch, where it runs alone on 32 CUs)
constexpr int MME_NT = 1024;
__global__ __launch_bounds__(MME_NT) void mm_energy_kernel(MMArgs a) {
    __shared__ float red[MME_NT];
    const grappa_mm_desc& d = a.d;
    const int b = blockIdx.x, C = d.C, tid = threadIdx.x;
    const int cs = C < MME_NT ? C : MME_NT;
    const int tpb = MME_NT / cs;
    const int j = tid / cs, cl = tid - j * cs;
    const bool active = j < tpb;
    for (int cbase = 0; cbase < C; cbase += cs) {
        const int c = cbase + cl;
        const bool cok = active && c < C;
        float total = 0.f;
        for (int l = 0; l < 4; ++l) {
            float acc = 0.f;
            const int t0 = d.mol_ptr[l][b], t1 = d.mol_ptr[l][b + 1];
            if (cok) {
                for (int t = t0 + j; t < t1; t += tpb) {
                    float e, x;
                    if (l == 0) {
                        V3 u;
                        x = bond_geom(ldv(d.xyz, d.idx[0][2 * t], C, c), ldv(d.xyz, d.idx[0][2 * t + 1], C, c), u);
                        const float dx = x - d.eq[0][t];
                        e = 0.5f * d.k[0][t] * dx * dx;
                    } else if (l == 1) {
                        V3 e0, e2;
                        x = angle_geom(ldv(d.xyz, d.idx[1][3 * t], C, c), ldv(d.xyz, d.idx[1][3 * t + 1], C, c),
                                       ldv(d.xyz, d.idx[1][3 * t + 2], C, c), e0, e2);
                        const float dx = x - d.eq[1][t];
                        e = 0.5f * d.k[1][t] * dx * dx;
                    } else {
                        V3 d0, d1, d2, d3;
                        const int* id = d.idx[l] + 4 * (size_t)t;
                        x = dihedral_geom(ldv(d.xyz, id[0], C, c), ldv(d.xyz, id[1], C, c), ldv(d.xyz, id[2], C, c), ldv(d.xyz, id[3], C, c),
                                          d0, d1, d2, d3);
                        e = torsion_energy(d.k[l] + (size_t)t * d.n_per[l], d.n_per[l], x, d.offset_torsion);
                    }
                    if (a.tuple_e[l]) a.tuple_e[l][(size_t)t * C + c] = e;
                    if (a.tuple_x[l]) a.tuple_x[l][(size_t)t * C + c] = x;
                    acc += e;
                }
            }
            red[tid] = acc;
            __syncthreads();
            if (cok && j == 0) {
                float s = 0.f;
                for (int jj = 0; jj < tpb; ++jj) s += red[jj * cs + cl];
                if (a.term_energy) a.term_energy[((size_t)l * d.B + b) * C + c] = s;
                total += s;
            }
            __syncthreads();
        }
        if (cok && j == 0) a.energy[(size_t)b * C + c] = total;
    }
}

// ------------------------------------------------------------------------------------------------ gradient
// MMG_SUB lanes per (atom, conformation) share the atom's incidences (an atom of a small molecule sits in ~60 tuples, each a dependent chain
// of geometry + trigonometry: one thread per (atom, conformation) took 130 - 150 us whatever the batch); their partial sums meet in a
// fixed-order butterfly
constexpr int MMG_SUB = 4;
__global__ __launch_bounds__(256) void mm_gradient_kernel(grappa_mm_desc d, float* __restrict__ grad) {
    const size_t tidg = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t gid = tidg / MMG_SUB;
    const int sub = (int)(tidg % MMG_SUB);
    const int C = d.C;
    const bool ok = gid < (size_t)d.N * C;                         // (whole groups of MMG_SUB lanes are in or out: 256 % MMG_SUB == 0)
    const int atom = ok ? (int)(gid / C) : 0, c = ok ? (int)(gid % C) : 0;
    V3 g = {0.f, 0.f, 0.f};
    const int i0 = ok ? d.inc_ptr[atom] : 0, i1 = ok ? d.inc_ptr[atom + 1] : 0;
    for (int i = i0 + sub; i < i1; i += MMG_SUB) {
        const int code = d.inc_code[i];
        const int pos = code & 3, l = (code >> 2) & 3, t = code >> 4;
        if (l == 0) {
            V3 u;
            const float r = bond_geom(ldv(d.xyz, d.idx[0][2 * t], C, c), ldv(d.xyz, d.idx[0][2 * t + 1], C, c), u);
            const float coef = d.k[0][t] * (r - d.eq[0][t]);
            g = g + (pos == 0 ? coef : -coef) * u;
        } else if (l == 1) {
            V3 e0, e2;
            const float th = angle_geom(ldv(d.xyz, d.idx[1][3 * t], C, c), ldv(d.xyz, d.idx[1][3 * t + 1], C, c),
                                        ldv(d.xyz, d.idx[1][3 * t + 2], C, c), e0, e2);
            const float coef = d.k[1][t] * (th - d.eq[1][t]);
            const V3 dv = pos == 0 ? e0 : (pos == 2 ? e2 : (-1.0f) * (e0 + e2));
            g = g + coef * dv;
        } else {
            V3 d0, d1, d2, d3;
            const int* id = d.idx[l] + 4 * (size_t)t;
            const float phi = dihedral_geom(ldv(d.xyz, id[0], C, c), ldv(d.xyz, id[1], C, c), ldv(d.xyz, id[2], C, c), ldv(d.xyz, id[3], C, c),
                                            d0, d1, d2, d3);
            const float* k = d.k[l] + (size_t)t * d.n_per[l];
            float coef = 0.f;
            for (int n = 1; n <= d.n_per[l]; ++n) coef -= (float)n * k[n - 1] * sinf((float)n * phi);
            const V3 dv = pos == 0 ? d0 : (pos == 1 ? d1 : (pos == 2 ? d2 : d3));
            g = g + coef * dv;
        }
    }
#pragma unroll
    for (int m = 1; m < MMG_SUB; m <<= 1) {
        g.x += __shfl_xor(g.x, m);
        g.y += __shfl_xor(g.y, m);
        g.z += __shfl_xor(g.z, m);
    }
    if (ok && sub == 0) {
        float* o = grad + gid * 3;
        o[0] = g.x; o[1] = g.y; o[2] = g.z;
    }
}

// ------------------------------------------------------------------------------------------------ backward
__device__ inline int mol_of(const int* __restrict__ ptr, int B, int t) {   // largest b with ptr[b] <= t
    int lo = 0, hi = B;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (ptr[mid] <= t) lo = mid; else hi = mid;
    }
    return lo;
}

struct BwdArgs {
    grappa_mm_desc d;
    const float* gE;
    const float* gG;
    float* gk[4];
    float* geq[4];
};

template <int L>   // lanes per tuple
__global__ __launch_bounds__(256) void mm_bwd_kernel(BwdArgs a, int level) {
    const grappa_mm_desc& d = a.d;
    const int C = d.C, T = d.T[level];
    const int tpw = 64 / L;
    const int gw = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const int t = gw * tpw + lane / L;
    const int cl = lane % L;
    const bool ok = t < T;
    const int tt = ok ? t : 0;
    if (T == 0) return;
    const int b = mol_of(d.mol_ptr[level], d.B, tt);
    if (level <= 1) {
        const int s = level == 0 ? 2 : 3;
        const int* id = d.idx[level] + (size_t)s * tt;
        const float k = d.k[level][tt], eq = d.eq[level][tt];
        float gk = 0.f, geq = 0.f;
        if (ok) {
            for (int c = cl; c < C; c += L) {
                float x, D = 0.f;
                if (level == 0) {
                    V3 u;
                    x = bond_geom(ldv(d.xyz, id[0], C, c), ldv(d.xyz, id[1], C, c), u);
                    if (a.gG) D = dot(ldv(a.gG, id[0], C, c) - ldv(a.gG, id[1], C, c), u);
                } else {
                    V3 e0, e2;
                    x = angle_geom(ldv(d.xyz, id[0], C, c), ldv(d.xyz, id[1], C, c), ldv(d.xyz, id[2], C, c), e0, e2);
                    if (a.gG) {
                        const V3 g1 = ldv(a.gG, id[1], C, c);
                        D = dot(ldv(a.gG, id[0], C, c) - g1, e0) + dot(ldv(a.gG, id[2], C, c) - g1, e2);
                    }
                }
                const float dx = x - eq;
                const float ge = a.gE ? a.gE[(size_t)b * C + c] : 0.f;
                gk += ge * 0.5f * dx * dx + dx * D;
                geq += -ge * k * dx - k * D;
            }
        }
        gk = group_sum(gk, L);
        geq = group_sum(geq, L);
        if (ok && cl == 0) {
            a.gk[level][t] = gk;
            a.geq[level][t] = geq;
        }
    } else {
        const int np = d.n_per[level];
        const int* id = d.idx[level] + 4 * (size_t)tt;
        const float* k = d.k[level] + (size_t)tt * np;
        float gk[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) gk[n] = 0.f;
        if (ok) {
            for (int c = cl; c < C; c += L) {
                V3 d0, d1, d2, d3;
                const float phi = dihedral_geom(ldv(d.xyz, id[0], C, c), ldv(d.xyz, id[1], C, c), ldv(d.xyz, id[2], C, c),
                                                ldv(d.xyz, id[3], C, c), d0, d1, d2, d3);
                float D = 0.f;
                if (a.gG)
                    D = dot(ldv(a.gG, id[0], C, c), d0) + dot(ldv(a.gG, id[1], C, c), d1) + dot(ldv(a.gG, id[2], C, c), d2) +
                        dot(ldv(a.gG, id[3], C, c), d3);
                const float ge = a.gE ? a.gE[(size_t)b * C + c] : 0.f;
#pragma unroll
                for (int n = 1; n <= 8; ++n) {
                    if (n <= np) {
                        float v = ge * cosf((float)n * phi) - (float)n * sinf((float)n * phi) * D;
                        if (d.offset_torsion) {
                            const float kn = k[n - 1];
                            v += ge * (kn > 0.f ? 1.0f : (kn < 0.f ? -1.0f : 0.f));
                        }
                        gk[n - 1] += v;
                    }
                }
            }
        }
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            if (n < np) {
                const float s = group_sum(gk[n], L);
                if (ok && cl == 0) a.gk[level][(size_t)t * np + n] = s;
            }
        }
    }
}

inline int lanes_per_tuple(int C) { return C > 32 ? 64 : C > 16 ? 32 : C > 8 ? 16 : 8; }

int validate(const grappa_mm_desc* d, bool need_inc) {
    if (!d || d->N < 0 || d->C <= 0 || d->B <= 0 || !d->xyz) return GRAPPA_ERR_ARG;
    for (int l = 0; l < 4; ++l) {
        if (d->T[l] < 0 || !d->mol_ptr[l]) return GRAPPA_ERR_ARG;
        if (d->T[l] > 0 && (!d->idx[l] || !d->k[l])) return GRAPPA_ERR_ARG;
        if (l < 2 && d->T[l] > 0 && !d->eq[l]) return GRAPPA_ERR_ARG;
        if (l >= 2 && (d->n_per[l] < 1 || d->n_per[l] > 8)) return GRAPPA_ERR_ARG;
        if (d->T[l] >= (1 << 27)) return GRAPPA_ERR_ARG;
    }
    if (need_inc && (!d->inc_ptr || (!d->inc_code && (d->T[0] + d->T[1] + d->T[2] + d->T[3]) > 0))) return GRAPPA_ERR_ARG;
    return GRAPPA_OK;
}

}  // namespace

extern "C" int grappa_mm_energy_fwd_f32(void* stream, const grappa_mm_desc* d, float* energy, float* term_energy, float* const tuple_e[4],
                                        float* const tuple_x[4]) {
    if (int rc = validate(d, false)) return rc;
    if (!energy) return GRAPPA_ERR_ARG;
    MMArgs a;
    a.d = *d;
    a.energy = energy;
    a.term_energy = term_energy;
    for (int l = 0; l < 4; ++l) {
        a.tuple_e[l] = tuple_e ? tuple_e[l] : nullptr;
        a.tuple_x[l] = tuple_x ? tuple_x[l] : nullptr;
    }
    GRAPPA_LAUNCH(mm_energy_kernel, dim3(d->B), dim3(MME_NT), 0, reinterpret_cast<hipStream_t>(stream), a);
    return grappa_launch_status();
}

extern "C" int grappa_mm_gradient_fwd_f32(void* stream, const grappa_mm_desc* d, float* grad) {
    if (int rc = validate(d, true)) return rc;
    if (d->N == 0) return GRAPPA_OK;
    if (!grad) return GRAPPA_ERR_ARG;
    const size_t total = (size_t)d->N * d->C;
    GRAPPA_LAUNCH(mm_gradient_kernel, dim3((unsigned)((total * MMG_SUB + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), *d, grad);
    return grappa_launch_status();
}

extern "C" int grappa_mm_bwd_f32(void* stream, const grappa_mm_desc* d, const float* gE, const float* gG, float* const gk[4],
                                 float* const geq[4]) {
    if (int rc = validate(d, false)) return rc;
    if (!gk || !geq) return GRAPPA_ERR_ARG;
    BwdArgs a;
    a.d = *d;
    a.gE = gE;
    a.gG = gG;
    for (int l = 0; l < 4; ++l) {
        a.gk[l] = gk[l];
        a.geq[l] = geq[l];
        if (d->T[l] > 0 && (!gk[l] || (l < 2 && !geq[l]))) return GRAPPA_ERR_ARG;
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int L = lanes_per_tuple(d->C);
    for (int l = 0; l < 4; ++l) {
        if (d->T[l] == 0) continue;
        const int tpw = 64 / L;
        const int waves = (d->T[l] + tpw - 1) / tpw;
        const dim3 grid((waves + 3) / 4);
        switch (L) {
            case 64: GRAPPA_LAUNCH(mm_bwd_kernel<64>, grid, dim3(256), 0, st, a, l); break;
            case 32: GRAPPA_LAUNCH(mm_bwd_kernel<32>, grid, dim3(256), 0, st, a, l); break;
            case 16: GRAPPA_LAUNCH(mm_bwd_kernel<16>, grid, dim3(256), 0, st, a, l); break;
            default: GRAPPA_LAUNCH(mm_bwd_kernel<8>, grid, dim3(256), 0, st, a, l); break;
        }
    }
    return grappa_launch_status();
}
