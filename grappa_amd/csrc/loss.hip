// MolwiseLoss forward + its gradient in one pass, one workgroup per molecule (training/loss.py:45-167).
// The reference loops over molecules in Python (dgl.unbatch + ~10 tiny kernels per molecule); here a
// molecule is a segment [ptr[b], ptr[b+1]) of the flat tables, reduced in a fixed order (reproducible).
#include "common.h"

namespace {

__device__ inline float block_sum(float v, float* red) {   // 256 threads; result in every thread
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void loss_ef_kernel(int B, int C, const int* __restrict__ atom_molptr, const float* __restrict__ energy,
                                                      const float* __restrict__ energy_ref, const float* __restrict__ is_dummy,
                                                      const float* __restrict__ grad, const float* __restrict__ grad_ref, float wE, float wG,
                                                      float inv_B, float* __restrict__ loss_mol, float* __restrict__ gE, float* __restrict__ gG) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    // ---- real-conformation count
    float nr = 0.f;
    for (int c = tid; c < C; c += 256) nr += (is_dummy && is_dummy[(size_t)b * C + c] != 0.f) ? 0.f : 1.f;
    const float nreal = block_sum(nr, red);
    float loss = 0.f;
    if (wE != 0.f && energy && energy_ref) {
        float se = 0.f, sr = 0.f;
        for (int c = tid; c < C; c += 256) {
            const float m = (is_dummy && is_dummy[(size_t)b * C + c] != 0.f) ? 0.f : 1.f;
            se += m * energy[(size_t)b * C + c];
            sr += m * energy_ref[(size_t)b * C + c];
        }
        const float me = block_sum(se, red) / nreal, mr = block_sum(sr, red) / nreal;
        float sq = 0.f, sd = 0.f;
        for (int c = tid; c < C; c += 256) {
            const float m = (is_dummy && is_dummy[(size_t)b * C + c] != 0.f) ? 0.f : 1.f;
            const float diff = (energy[(size_t)b * C + c] - me) - (energy_ref[(size_t)b * C + c] - mr);
            sq += m * diff * diff;
            sd += m * diff;
        }
        const float tsq = block_sum(sq, red), tsd = block_sum(sd, red);
        loss += wE * tsq / nreal;
        if (gE) {
            for (int c = tid; c < C; c += 256) {
                const float m = (is_dummy && is_dummy[(size_t)b * C + c] != 0.f) ? 0.f : 1.f;
                const float diff = (energy[(size_t)b * C + c] - me) - (energy_ref[(size_t)b * C + c] - mr);
                gE[(size_t)b * C + c] = inv_B * wE * 2.0f / nreal * m * (diff - tsd / nreal);
            }
        }
    } else if (gE) {
        for (int c = tid; c < C; c += 256) gE[(size_t)b * C + c] = 0.f;
    }
    const int a0 = atom_molptr[b], a1 = atom_molptr[b + 1];
    const size_t base = (size_t)a0 * C * 3, n = (size_t)(a1 - a0) * C * 3;
    if (wG != 0.f && grad && grad_ref) {
        const float denom = (float)(a1 - a0) * nreal * 3.0f;
        float sq = 0.f;
        for (size_t i = tid; i < n; i += 256) {
            const int c = (int)((i / 3) % C);
            const float m = (is_dummy && is_dummy[(size_t)b * C + c] != 0.f) ? 0.f : 1.f;
            const float diff = grad[base + i] - grad_ref[base + i];
            sq += m * diff * diff;
            if (gG) gG[base + i] = inv_B * wG * 2.0f * m * diff / denom;
        }
        loss += wG * block_sum(sq, red) / denom;
    } else if (gG) {
        for (size_t i = tid; i < n; i += 256) gG[base + i] = 0.f;
    }
    if (tid == 0) loss_mol[b] = loss;
}

// FastEvaluator.step (training/evaluation.py:53-113): per molecule the sums the reference takes after dgl.unbatch --
// sum_c ((E - mean E) - (E_ref - mean E_ref))^2 over the real conformations (get_energies centres per molecule,
// utils/graph_utils.py:35-63), sum over atoms, real conformations and xyz of (G - G_ref)^2, and the two counts
// (conformations; atoms x conformations = 3-vectors).  out[b] = {se_E, n_E, se_G, n_G}.
__global__ __launch_bounds__(256) void eval_se_kernel(int B, int C, const int* __restrict__ atom_molptr, const float* __restrict__ energy,
                                                      const float* __restrict__ energy_ref, const float* __restrict__ is_dummy,
                                                      const float* __restrict__ grad, const float* __restrict__ grad_ref, float* __restrict__ out) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    float nr = 0.f, se = 0.f, sr = 0.f;
    for (int c = tid; c < C; c += 256) {
        const float m = (is_dummy && is_dummy[(size_t)b * C + c] != 0.f) ? 0.f : 1.f;
        nr += m;
        se += m * energy[(size_t)b * C + c];
        sr += m * energy_ref[(size_t)b * C + c];
    }
    const float nreal = block_sum(nr, red);
    const float me = block_sum(se, red) / nreal, mr = block_sum(sr, red) / nreal;
    float sq = 0.f;
    for (int c = tid; c < C; c += 256) {
        const float m = (is_dummy && is_dummy[(size_t)b * C + c] != 0.f) ? 0.f : 1.f;
        const float diff = (energy[(size_t)b * C + c] - me) - (energy_ref[(size_t)b * C + c] - mr);
        sq += m * diff * diff;
    }
    const float se_e = block_sum(sq, red);
    float se_g = 0.f;
    const int a0 = atom_molptr[b], a1 = atom_molptr[b + 1];
    if (grad && grad_ref) {
        const size_t base = (size_t)a0 * C * 3, n = (size_t)(a1 - a0) * C * 3;
        float gq = 0.f;
        for (size_t i = tid; i < n; i += 256) {
            const int c = (int)((i / 3) % C);
            const float m = (is_dummy && is_dummy[(size_t)b * C + c] != 0.f) ? 0.f : 1.f;
            const float diff = grad[base + i] - grad_ref[base + i];
            gq += m * diff * diff;
        }
        se_g = block_sum(gq, red);
    }
    if (tid == 0) {
        out[4 * (size_t)b] = se_e;
        out[4 * (size_t)b + 1] = nreal;
        out[4 * (size_t)b + 2] = se_g;
        out[4 * (size_t)b + 3] = (grad && grad_ref) ? (float)(a1 - a0) * nreal : 0.f;
    }
}

struct PLArgs {
    grappa_ploss_desc d;
    float* loss_mol;
    float* gp[6];
};

__global__ __launch_bounds__(256) void loss_param_kernel(PLArgs a) {
    __shared__ float red[4];
    const grappa_ploss_desc& d = a.d;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float pw = d.pw ? d.pw[b] : 0.f;
    float den = 0.f;
    for (int l = 0; l < 6; ++l)
        if (d.ref[l] && d.p[l]) den += (float)(d.mol_ptr[l][b + 1] - d.mol_ptr[l][b]) * (float)d.width[l];
    float num = 0.f, regsum = 0.f;
    for (int l = 0; l < 6; ++l) {
        if (!d.p[l]) continue;
        const int t0 = d.mol_ptr[l][b], t1 = d.mol_ptr[l][b + 1];
        const int w = d.width[l], rw = d.ref_width[l];
        const int cnt = (t1 - t0) * w;
        const bool mse = d.ref[l] && pw != 0.f;
        const float fac2 = d.fac[l] * d.fac[l];
        const float regc = (d.reg[l] > 0.f && cnt > 0) ? d.reg[l] / (float)cnt : 0.f;
        float lreg = 0.f;
        for (int i = tid; i < cnt; i += 256) {
            const int t = t0 + i / w, col = i % w;
            const float p = d.p[l][(size_t)t * w + col];
            float g = 0.f;
            if (mse) {
                const float r = col < rw ? d.ref[l][(size_t)t * rw + col] : 0.f;
                if (!isnan(r)) {
                    const float diff = p - r;
                    num += fac2 * diff * diff;
                    g += pw * fac2 * 2.0f * diff / den;
                }
            }
            if (regc > 0.f) {
                lreg += p * p;
                g += regc * 2.0f * p;
            }
            if (a.gp[l]) a.gp[l][(size_t)t * w + col] = d.inv_B * g;
        }
        regsum += regc * block_sum(lreg, red);
    }
    const float tnum = block_sum(num, red);
    if (tid == 0) {
        float loss = regsum;
        if (pw != 0.f && den > 0.f) loss += pw * tnum / den;
        a.loss_mol[b] += loss;
    }
}

}  // namespace

extern "C" int grappa_loss_ef_fwd_bwd_f32(void* stream, int B, int C, int N, const int* atom_molptr, const float* energy, const float* energy_ref,
                                          const float* is_dummy, const float* grad, const float* grad_ref, float wE, float wG, float inv_B,
                                          float* loss_mol, float* gE, float* gG) {
    if (B <= 0 || C <= 0 || N < 0 || !atom_molptr || !loss_mol) return GRAPPA_ERR_ARG;
    if (wE != 0.f && (!energy || !energy_ref)) return GRAPPA_ERR_ARG;
    if (wG != 0.f && (!grad || !grad_ref)) return GRAPPA_ERR_ARG;
    GRAPPA_LAUNCH(loss_ef_kernel, dim3(B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), B, C, atom_molptr, energy, energy_ref,
                       is_dummy, grad, grad_ref, wE, wG, inv_B, loss_mol, gE, gG);
    return grappa_launch_status();
}

extern "C" int grappa_loss_param_fwd_bwd_f32(void* stream, const grappa_ploss_desc* d, float* loss_mol, float* const gp[6]) {
    if (!d || d->B <= 0 || !loss_mol) return GRAPPA_ERR_ARG;
    PLArgs a;
    a.d = *d;
    a.loss_mol = loss_mol;
    for (int l = 0; l < 6; ++l) {
        a.gp[l] = gp ? gp[l] : nullptr;
        if (d->p[l] && (!d->mol_ptr[l] || d->width[l] < 1)) return GRAPPA_ERR_ARG;
        if (d->ref[l] && d->ref_width[l] < 1) return GRAPPA_ERR_ARG;
    }
    GRAPPA_LAUNCH(loss_param_kernel, dim3(d->B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    return grappa_launch_status();
}

extern "C" int grappa_eval_se_f32(void* stream, int B, int C, int N, const int* atom_molptr, const float* energy, const float* energy_ref,
                                  const float* is_dummy, const float* grad, const float* grad_ref, float* out) {
    if (B <= 0 || C <= 0 || N < 0 || !atom_molptr || !energy || !energy_ref || !out) return GRAPPA_ERR_ARG;
    if ((grad == nullptr) != (grad_ref == nullptr)) return GRAPPA_ERR_ARG;
    GRAPPA_LAUNCH(eval_se_kernel, dim3(B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), B, C, atom_molptr, energy, energy_ref, is_dummy,
                       grad, grad_ref, out);
    return grappa_launch_status();
}
