"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product package `grappa_amd`.

`RefBackend`: plain-PyTorch restatement, op by op, of the C ABI in include/grappa_hip.h (same method
names and argument meaning as grappa_amd.backend.HipBackend, results written into the caller's output
tensors).  Two uses, both under tests/:
  * `-m gpu` parity tests run every HIP entry point and this restatement on the same seeded inputs;
  * `-m "not gpu"` tests install it as a test-only backend so that the product's host logic
    (autograd wiring, buffer management, batching, optimiser, data-parallel sharding) runs on a
    CPU-only machine.  The hand-derived backward formulas here are themselves checked against
    torch.autograd in tests/test_ops_ref_autograd.py.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

M32 = 0xFFFFFFFF


def dropout_keep(seed: int, idx: torch.Tensor, p: float) -> torch.Tensor:
    """csrc/common.h grappa_keep: murmur3-fmix32 style hash of the element index keyed by the 64-bit seed; 24-bit uniform;
    keep iff u >= p.  int64 arithmetic masked to 32 bits."""
    seed &= (1 << 64) - 1
    lo, hi = seed & M32, (seed >> 32) & M32
    i = idx.to(torch.int64)
    h = ((i & M32) * 0x9E3779B1 + lo) & M32
    h = h ^ ((((i >> 32) & M32) * 0x85EBCA77) & M32)
    h = h ^ (h >> 16)
    h = (h * 0x85EBCA6B) & M32
    h = h ^ (h >> 13)
    h = (h * 0xC2B2AE35) & M32
    h = h ^ (h >> 16)
    h = (h + hi) & M32
    h = h ^ (h >> 15)
    h = (h * 0x2C1B3C6D) & M32
    h = h ^ (h >> 12)
    u = (h >> 8).to(torch.float32) * (1.0 / 16777216.0)
    return u >= p


def elu_grad_from_out(y):
    return torch.where(y > 0, torch.ones_like(y), y + 1.0)


class RefBackend:
    name = "ref"

    # ------------------------------------------------------------------ dense
    def gemm(self, a, b, out, *, M, N, K, a_kcontig=True, b_kcontig=True, bias=None, res=None, aux=None, pre=None, act=0,
             drop_p=0.0, drop_seed=0, accumulate=False, out2=None, a_colsum=None, precision=None, a_scales=None, b_scales=None, out_amax=False):
        A = a if a_kcontig else a.t()
        Bm = b if b_kcontig else b.t()
        assert tuple(A.shape) == (M, K) and tuple(Bm.shape) == (N, K) and tuple(out.shape) == (M, N)
        if a_colsum is not None:
            assert not a_kcontig
            a_colsum.add_(A.sum(1))
        v = A @ Bm.t()
        if pre is not None:
            v = v + pre
        if bias is not None:
            v = v + bias
        if act == 1:
            v = F.elu(v)
        if aux is not None:
            v = v * elu_grad_from_out(aux)
        target = out
        if out2 is not None:
            out.copy_(v)
            target = out2
        if drop_p > 0:
            idx = torch.arange(M * N, device=v.device).view(M, N)
            v = torch.where(dropout_keep(drop_seed, idx, drop_p), v * (1.0 / (1.0 - drop_p)), torch.zeros_like(v))
        if res is not None:
            v = v + res
        if accumulate:
            v = v + target
        target.copy_(v)
        return (None, None) if out_amax else None      # (the HIP backend's fp16-split products hand back operand maxima here)

    def colsum(self, x, out, accumulate=False):
        s = x.sum(0)
        out.copy_(out + s if accumulate else s)

    def act_dropout_bwd(self, dy, y, drop_p, drop_seed, dz):
        M, N = dy.shape
        v = dy.clone()
        if drop_p > 0:
            idx = torch.arange(M * N, device=v.device).view(M, N)
            v = torch.where(dropout_keep(drop_seed, idx, drop_p), v * (1.0 / (1.0 - drop_p)), torch.zeros_like(v))
        if y is not None:
            v = v * elu_grad_from_out(y)
        dz.copy_(v)

    def add(self, x, z, y):
        y.copy_(x + z)

    # ------------------------------------------------------------------ layer norm
    def layernorm_fwd(self, x, gamma, beta, y, mean, rstd):
        mu = x.mean(1)
        var = ((x - mu[:, None]) ** 2).mean(1)
        rs = 1.0 / torch.sqrt(var + 1e-5)
        y.copy_((x - mu[:, None]) * rs[:, None] * gamma + beta)
        if mean is not None:
            mean.copy_(mu)
            rstd.copy_(rs)

    def layernorm_bwd(self, dy, x, mean, rstd, gamma, dx, dgamma, dbeta, accumulate=True):
        xh = (x - mean[:, None]) * rstd[:, None]
        g = dy * gamma
        m1 = g.mean(1, keepdim=True)
        m2 = (g * xh).mean(1, keepdim=True)
        dgv, dbv = (dy * xh).sum(0), dy.sum(0)
        dx.copy_(rstd[:, None] * (g - m1 - xh * m2))
        dgamma.copy_(dgamma + dgv if accumulate else dgv)
        dbeta.copy_(dbeta + dbv if accumulate else dbv)

    # ------------------------------------------------------------------ graph
    @staticmethod
    def _edges(plan):
        deg = (plan.indptr[1:] - plan.indptr[:-1]).long()
        dst = torch.repeat_interleave(torch.arange(plan.N, device=plan.indptr.device), deg)
        return plan.indices.long(), dst

    def gat_fwd(self, plan, ft, H, D, out, alpha):
        src, dst = self._edges(plan)
        N = ft.shape[0]
        f = ft.view(N, H, D)
        a = (f[src] * f[dst]).sum(-1) / math.sqrt(D)
        mx = torch.full((N, H), float("-inf"), dtype=ft.dtype, device=ft.device).scatter_reduce(0, dst[:, None].expand(-1, H), a, "amax")
        ex = torch.exp(a - mx[dst])
        den = torch.zeros((N, H), dtype=ft.dtype, device=ft.device).index_add(0, dst, ex)
        al = ex / den[dst]
        alpha.view(-1, H).copy_(al)
        out.copy_(torch.zeros_like(f).index_add(0, dst, f[src] * al[..., None]).view(N, H * D))

    def gat_bwd(self, plan, ft, out, alpha, dout, H, D, dft):
        src, dst = self._edges(plan)
        N = ft.shape[0]
        f, o, do = ft.view(N, H, D), out.view(N, H, D), dout.view(N, H, D)
        al = alpha.view(-1, H)
        delta = (do * o).sum(-1)                                         # (N,H)
        ds = al * ((do[dst] * f[src]).sum(-1) - delta[dst])              # (E,H)
        g = torch.zeros_like(f)
        g = g.index_add(0, src, al[..., None] * do[dst] + (ds / math.sqrt(D))[..., None] * f[dst])
        g = g.index_add(0, dst, (ds / math.sqrt(D))[..., None] * f[src])
        dft.copy_(g.view(N, H * D))

    def neighbor_mean(self, plan, x, out, scale_by_neighbor):
        src, dst = self._edges(plan)
        deg = (plan.indptr[1:] - plan.indptr[:-1]).to(x.dtype)
        sc = 1.0 / (deg[src] if scale_by_neighbor else deg[dst])
        out.copy_(torch.zeros_like(x).index_add(0, dst, x[src] * sc[:, None]))

    def charge_encoding(self, q, dim, lo, hi, out, col0):
        v = torch.clamp(q, lo, hi)
        s = (v + hi) / (hi - lo)
        half = dim // 2
        f = torch.exp(torch.arange(half, dtype=torch.float32, device=q.device) * (-math.log(10000.0) / half))
        out[:, col0:col0 + dim:2] = torch.sin(s[:, None] * f)
        out[:, col0 + 1:col0 + dim:2] = torch.cos(s[:, None] * f)

    # ------------------------------------------------------------------ tuples
    def tuple_gather_fwd(self, a, idx, s, pe, x):
        T, W = idx.shape[0], x.shape[1]
        if T == 0:
            return
        rows = a[idx.long().t().reshape(-1), :W].clone()           # row = pos*T + t
        if pe is not None:
            rows[:, W - 1] = pe.repeat_interleave(T)
        x.copy_(rows)

    def tuple_gather_bwd(self, inv_ptr, inv_rows, dx, da, has_pe, accumulate=False):
        N, W = da.shape[0], dx.shape[1]
        cnt = (inv_ptr[1:] - inv_ptr[:-1]).long()
        atom = torch.repeat_interleave(torch.arange(N, device=da.device), cnt)
        g = torch.zeros((N, W), dtype=dx.dtype, device=dx.device)
        if inv_rows.numel():
            g = g.index_add(0, atom, dx[inv_rows.long()])
        if has_pe:
            g[:, W - 1] = 0
        da[:, :W] = (da[:, :W] + g) if accumulate else g

    @staticmethod
    def _attn_parts(qkv, s, T, nheads):
        Fd = qkv.shape[1] // 3
        dh = Fd // nheads
        q, k, v = [t.reshape(s, T, nheads, dh) for t in qkv.split(Fd, dim=1)]
        return q, k, v, dh

    def seqattn_fwd(self, qkv, s, T, nheads, out):
        if T == 0:
            return
        q, k, v, dh = self._attn_parts(qkv, s, T, nheads)
        sc = torch.einsum("ithd,jthd->thij", q, k) / math.sqrt(dh)
        p = torch.softmax(sc, dim=-1)
        out.copy_(torch.einsum("thij,jthd->ithd", p, v).reshape(s * T, -1))

    def seqattn_bwd(self, qkv, dout, s, T, nheads, dqkv):
        if T == 0:
            return
        q, k, v, dh = self._attn_parts(qkv, s, T, nheads)
        go = dout.reshape(s, T, nheads, dh)
        scale = 1.0 / math.sqrt(dh)
        p = torch.softmax(torch.einsum("ithd,jthd->thij", q, k) * scale, dim=-1)
        dp = torch.einsum("ithd,jthd->thij", go, v)
        ds = p * (dp - (p * dp).sum(-1, keepdim=True)) * scale
        dq = torch.einsum("thij,jthd->ithd", ds, k)
        dk = torch.einsum("thij,ithd->jthd", ds, q)
        dv = torch.einsum("thij,ithd->jthd", p, go)
        dqkv.copy_(torch.cat([t.reshape(s * T, -1) for t in (dq, dk, dv)], dim=1))

    def writer_layer(self, x, s, T, nheads, drop_p, seed1, seed2, n1_w, n1_b, w_in, b_in, w_o, b_o, nf_w, nf_b, w1, b1, w2, b2, rnd=None):
        """one transformer layer of a writer head, restated from /root/reference/src/grappa/models/network_utils.py:112-133
        (DottedAttWithMLP.forward: norm1 -> nn.MultiheadAttention -> dropout -> + normed input -> ff) and :44-54 (FeedForwardLayer.forward:
        norm1 -> linear1 -> ELU -> linear2 -> dropout -> + normed input), on a token table x (s*T, F), row = pos*T + t.  The checker of the
        fused kernel grappa_writer_head_fwd (include/grappa_hip.h).  rnd: the rounding a storage configuration applies to every tensor the
        unfused kernels store (identity for fp32; `lambda t: t.bfloat16().float()` for the bf16 storage configuration).
        -> dict(x1, mean1, rstd1, qkv, att, x2, meanf, rstdf, x3, u, out)"""
        rnd = rnd or (lambda t: t)
        M, Fd = x.shape
        new = lambda *sh: torch.empty(sh, dtype=x.dtype, device=x.device)      # noqa: E731
        r = {}
        x1, mean1, rstd1 = new(M, Fd), new(M), new(M)
        self.layernorm_fwd(x, n1_w, n1_b, x1, mean1, rstd1)
        x1 = rnd(x1)
        qkv = new(M, 3 * Fd)
        self.gemm(x1, w_in, qkv, M=M, N=3 * Fd, K=Fd, bias=b_in)
        qkv = rnd(qkv)
        att = new(M, Fd)
        self.seqattn_fwd(qkv, s, T, nheads, att)
        att = rnd(att)
        x2 = new(M, Fd)
        self.gemm(att, w_o, x2, M=M, N=Fd, K=Fd, bias=b_o, drop_p=drop_p, drop_seed=seed1, res=x1)
        x2 = rnd(x2)
        x3, meanf, rstdf = new(M, Fd), new(M), new(M)
        self.layernorm_fwd(x2, nf_w, nf_b, x3, meanf, rstdf)
        x3 = rnd(x3)
        u = new(M, Fd)
        self.gemm(x3, w1, u, M=M, N=Fd, K=Fd, bias=b1, act=1)
        u = rnd(u)
        out = new(M, Fd)
        self.gemm(u, w2, out, M=M, N=Fd, K=Fd, bias=b2, drop_p=drop_p, drop_seed=seed2, res=x3)
        out = rnd(out)
        return dict(x1=x1, mean1=mean1, rstd1=rstd1, qkv=qkv, att=att, x2=x2, meanf=meanf, rstdf=rstdf, x3=x3, u=u, out=out)

    def perm_concat_fwd(self, x, s, T, perms, z):
        Fd = x.shape[1]
        xv = x.view(s, T, Fd)
        z.copy_(torch.stack([xv[list(p)] for p in perms], 0).transpose(1, 2).reshape(len(perms) * T, s * Fd))

    def perm_concat_bwd(self, dz, s, T, perms, dx):
        Fd = dx.shape[1]
        g = torch.zeros((s, T, Fd), dtype=dz.dtype, device=dz.device)
        dzv = dz.view(len(perms), T, s, Fd)
        for pi, p in enumerate(perms):
            for j, i in enumerate(p):
                g[i] += dzv[pi, :, j]
        dx.copy_(g.view(s * T, Fd))

    @staticmethod
    def _to_pos(c, mos, std, mn):
        return std * (F.elu(mos + c - 1) + 1) + mn

    @staticmethod
    def _to_pos_grad(c, mos, std):
        z = mos + c - 1
        return std * torch.where(z > 0, torch.ones_like(z), torch.exp(z))

    def param_out_fwd(self, kind, o, T, P, n_per, gated, cutoff, consts, k, eq):
        if T == 0:
            return
        c = o.view(P, T, -1).sum(0)
        if kind == 2:
            if gated:
                v = c[:, :n_per] * torch.sigmoid(c[:, n_per:2 * n_per]) * consts[:n_per]
            else:
                v = c[:, :n_per] * consts[:n_per] + consts[n_per:2 * n_per]
            if cutoff > 0:
                v = torch.where(v.abs() > cutoff, v, torch.zeros_like(v))
            k.view(T, n_per).copy_(v)
            return
        if kind == 0:
            eq.copy_(self._to_pos(c[:, 0], consts[0], consts[1], consts[2]))
        else:
            eq.copy_(consts[1] * torch.sigmoid(consts[0] * c[:, 0]))
        k.copy_(self._to_pos(c[:, 1], consts[3], consts[4], consts[5]))

    def param_out_bwd_stats(self, kind, o, T, P, n_per, gated, cutoff, consts, dk, deq, d_consts):
        """dL/d consts of the output map by autograd through the restated forward (reference models/final_layer.py)"""
        with torch.enable_grad():
            return self._param_out_bwd_stats(kind, o, T, P, n_per, gated, cutoff, consts, dk, deq, d_consts)

    def _param_out_bwd_stats(self, kind, o, T, P, n_per, gated, cutoff, consts, dk, deq, d_consts):
        cst = consts.detach().clone().requires_grad_(True)
        c = o.detach().view(P, T, -1).sum(0)
        if kind == 2:
            if gated:
                v = c[:, :n_per] * torch.sigmoid(c[:, n_per:2 * n_per]) * cst[:n_per]
            else:
                v = c[:, :n_per] * cst[:n_per] + cst[n_per:2 * n_per]
            if cutoff > 0:
                v = torch.where(v.abs() > cutoff, v, torch.zeros_like(v))
            loss = (v * dk.view(T, n_per)).sum() if dk is not None else v.sum() * 0
        else:
            eq = self._to_pos(c[:, 0], cst[0], cst[1], cst[2]) if kind == 0 else cst[1] * torch.sigmoid(cst[0] * c[:, 0])
            k = self._to_pos(c[:, 1], cst[3], cst[4], cst[5])
            loss = (eq * deq).sum() if deq is not None else eq.sum() * 0
            loss = loss + ((k * dk).sum() if dk is not None else 0)
        d_consts.copy_(torch.autograd.grad(loss, cst, allow_unused=True)[0])

    def param_out_bwd(self, kind, o, T, P, n_per, gated, cutoff, consts, dk, deq, d_o):
        if T == 0:
            return
        c = o.view(P, T, -1).sum(0)
        g = torch.zeros_like(c)
        if kind == 2:
            up = dk.view(T, n_per) if dk is not None else torch.zeros((T, n_per), dtype=o.dtype, device=o.device)
            if gated:
                sg = torch.sigmoid(c[:, n_per:2 * n_per])
                v = c[:, :n_per] * sg * consts[:n_per]
            else:
                v = c[:, :n_per] * consts[:n_per] + consts[n_per:2 * n_per]
            if cutoff > 0:
                up = torch.where(v.abs() > cutoff, up, torch.zeros_like(up))
            if gated:
                g[:, :n_per] = up * sg * consts[:n_per]
                g[:, n_per:2 * n_per] = up * c[:, :n_per] * consts[:n_per] * sg * (1 - sg)
            else:
                g[:, :n_per] = up * consts[:n_per]
        else:
            ueq = deq if deq is not None else torch.zeros(T, dtype=o.dtype, device=o.device)
            uk = dk if dk is not None else torch.zeros(T, dtype=o.dtype, device=o.device)
            if kind == 0:
                g[:, 0] = ueq * self._to_pos_grad(c[:, 0], consts[0], consts[1])
            else:
                sg = torch.sigmoid(consts[0] * c[:, 0])
                g[:, 0] = ueq * consts[1] * consts[0] * sg * (1 - sg)
            g[:, 1] = uk * self._to_pos_grad(c[:, 1], consts[3], consts[4])
        d_o.copy_(g.unsqueeze(0).expand(P, -1, -1).reshape(P * T, -1))

    # ------------------------------------------------------------------ MM energy (closed forms, as in csrc/mm_energy.hip)
    @staticmethod
    def _bond(p0, p1):
        d = p0 - p1
        r = d.norm(dim=-1)
        return r, d / r.clamp(min=1e-20)[..., None]

    @staticmethod
    def _angle(p0, p1, p2):
        u, v = p0 - p1, p2 - p1
        w = torch.cross(u, v, dim=-1)
        wl = w.norm(dim=-1)
        th = torch.atan2(wl, (u * v).sum(-1))
        iw = 1.0 / wl.clamp(min=1e-20)
        e0 = (iw / (u * u).sum(-1).clamp(min=1e-20))[..., None] * torch.cross(u, w, dim=-1)
        e2 = (iw / (v * v).sum(-1).clamp(min=1e-20))[..., None] * torch.cross(w, v, dim=-1)
        return th, e0, e2

    @staticmethod
    def _dihedral(p0, p1, p2, p3):
        a, b, c = p1 - p0, p1 - p2, p3 - p2
        n1, n2 = torch.cross(a, b, dim=-1), torch.cross(b, c, dim=-1)
        b2 = (b * b).sum(-1)
        bl = b2.sqrt()
        y = (torch.cross(n1, n2, dim=-1) * b).sum(-1) / bl.clamp(min=1e-20)
        x = (n1 * n2).sum(-1)
        phi = torch.atan2(y, x)
        d0 = (-bl / (n1 * n1).sum(-1).clamp(min=1e-20))[..., None] * n1
        d3 = (bl / (n2 * n2).sum(-1).clamp(min=1e-20))[..., None] * n2
        p = ((a * b).sum(-1) / b2.clamp(min=1e-20))[..., None]
        q = ((c * b).sum(-1) / b2.clamp(min=1e-20))[..., None]
        return phi, d0, (p - 1) * d0 - q * d3, (q - 1) * d3 - p * d0, d3

    @staticmethod
    def _levels(plan):
        from grappa_amd.constants import TUPLE_LEVELS
        return TUPLE_LEVELS

    def _geom(self, plan, xyz, l, lvl):
        idx = plan.idx32[lvl].long()
        pos = [xyz[idx[:, j]] for j in range(idx.shape[1])]        # each (T,C,3)
        if l == 0:
            x, u = self._bond(*pos)
            return x, [u, -u]
        if l == 1:
            x, e0, e2 = self._angle(*pos)
            return x, [e0, -(e0 + e2), e2]
        x, d0, d1, d2, d3 = self._dihedral(*pos)
        return x, [d0, d1, d2, d3]

    @staticmethod
    def _seg(plan, lvl, dev):
        ptr = plan.mol_ptr[lvl].long()
        return torch.repeat_interleave(torch.arange(plan.B, device=dev), ptr[1:] - ptr[:-1])

    def mm_energy_fwd(self, plan, xyz, ks, eqs, n_per, offset_torsion, energy, term_energy, tuple_e=None, tuple_x=None):
        B, Cc = plan.B, xyz.shape[1]
        total = torch.zeros((B, Cc), dtype=xyz.dtype, device=xyz.device)
        for l, lvl in enumerate(self._levels(plan)):
            T = plan.T[lvl]
            if T == 0:
                e = torch.zeros((0, Cc), dtype=xyz.dtype, device=xyz.device)
                x = e
            else:
                x, _ = self._geom(plan, xyz, l, lvl)
                if l < 2:
                    e = 0.5 * ks[l][:, None] * (x - eqs[l][:, None]) ** 2
                else:
                    k = ks[l].view(T, n_per[l])
                    n = torch.arange(1, n_per[l] + 1, device=xyz.device, dtype=xyz.dtype).view(1, -1, 1)
                    e = (k[:, :, None] * torch.cos(n * x[:, None, :])).sum(1)
                    if offset_torsion:
                        e = e + k.abs().sum(1, keepdim=True)
            contrib = torch.zeros((B, Cc), dtype=xyz.dtype, device=xyz.device).index_add(0, self._seg(plan, lvl, xyz.device), e)
            if term_energy is not None:
                term_energy[l].copy_(contrib)
            if tuple_e is not None and tuple_e[l] is not None:
                tuple_e[l].copy_(e)
            if tuple_x is not None and tuple_x[l] is not None:
                tuple_x[l].copy_(x)
            total = total + contrib
        energy.copy_(total)

    def mm_gradient_fwd(self, plan, xyz, ks, eqs, n_per, grad):
        g = torch.zeros_like(xyz)
        for l, lvl in enumerate(self._levels(plan)):
            T = plan.T[lvl]
            if T == 0:
                continue
            x, ders = self._geom(plan, xyz, l, lvl)
            if l < 2:
                coef = ks[l][:, None] * (x - eqs[l][:, None])
            else:
                k = ks[l].view(T, n_per[l])
                n = torch.arange(1, n_per[l] + 1, device=xyz.device, dtype=xyz.dtype).view(1, -1, 1)
                coef = -(n * k[:, :, None] * torch.sin(n * x[:, None, :])).sum(1)
            idx = plan.idx32[lvl].long()
            for j, dj in enumerate(ders):
                g = g.index_add(0, idx[:, j], coef[..., None] * dj)
        grad.copy_(g)

    def mm_bwd(self, plan, xyz, ks, eqs, n_per, offset_torsion, gE, gG, gks, geqs):
        Cc = xyz.shape[1]
        for l, lvl in enumerate(self._levels(plan)):
            T = plan.T[lvl]
            if T == 0:
                continue
            x, ders = self._geom(plan, xyz, l, lvl)
            idx = plan.idx32[lvl].long()
            D = torch.zeros((T, Cc), dtype=xyz.dtype, device=xyz.device)
            if gG is not None:
                for j, dj in enumerate(ders):
                    D = D + (gG[idx[:, j]] * dj).sum(-1)
            ge = gE[self._seg(plan, lvl, xyz.device)] if gE is not None else torch.zeros((T, Cc), dtype=xyz.dtype, device=xyz.device)
            if l < 2:
                dx = x - eqs[l][:, None]
                gks[l].copy_((ge * 0.5 * dx * dx + dx * D).sum(1))
                geqs[l].copy_((-ge * ks[l][:, None] * dx - ks[l][:, None] * D).sum(1))
            else:
                k = ks[l].view(T, n_per[l])
                n = torch.arange(1, n_per[l] + 1, device=xyz.device, dtype=xyz.dtype).view(1, -1, 1)
                v = ge[:, None, :] * torch.cos(n * x[:, None, :]) - n * torch.sin(n * x[:, None, :]) * D[:, None, :]
                if offset_torsion:
                    v = v + ge[:, None, :] * torch.sign(k)[:, :, None]
                gks[l].view(T, n_per[l]).copy_(v.sum(-1))

    # ------------------------------------------------------------------ loss
    def collate_gather(self, tables, B):
        """include/grappa_hip.h grappa_collate_batch, table by table and slot by slot (4-byte elements viewed as int32)"""
        for t in tables:
            w, mode = int(t["width"]), t["mode"]
            src, dst = t["src"].view(torch.int32).reshape(-1), t["dst"].view(torch.int32).reshape(-1)
            for j in range(B):
                r0, r1, s0 = int(t["dst_row"][j]), int(t["dst_row"][j + 1]), int(t["src_row"][j])
                n = r1 - r0
                if n == 0:
                    continue
                if mode == "conf":
                    c_src, c_out = int(t["p0"][j]), int(t["c0"])
                    sel = t["p1"][j * c_out:(j + 1) * c_out].long()
                    blk = src[s0:s0 + n * c_src * w].reshape(n, c_src, w)
                    dst[r0 * c_out * w:r1 * c_out * w] = blk[:, sel].reshape(-1)
                    continue
                v = src[s0 * w:(s0 + n) * w]
                if mode == "add":
                    v = v + int(t["p0"][j])
                elif mode == "inv_rows":
                    t_mol, t_off, t_batch = int(t["p0"][j]), int(t["p1"][j]), int(t["c0"])
                    pos = torch.div(v, t_mol, rounding_mode="floor")
                    v = pos * t_batch + t_off + (v - pos * t_mol)
                elif mode == "inc_code":
                    off = t["p0"][4 * j:4 * j + 4]
                    v = v + (off[((v >> 2) & 3).long()] << 4)
                dst[r0 * w:r1 * w] = v

    def eval_se(self, plan, energy, energy_ref, is_dummy, grad, grad_ref, out):
        """training/evaluation.py:53-113 per molecule (after unbatch: dummy conformations deleted, energies centred)"""
        B, dev = plan.B, out.device
        m = torch.ones_like(energy) if is_dummy is None else (is_dummy == 0).float()
        n_real = getattr(plan, "n_real_mols", None)              # a batch that ends in a padding molecule: its rows of `out` stay zero (see loss_ef)
        real = torch.arange(B, device=dev) < (B if n_real is None else int(n_real))
        m = m * real[:, None].float()
        nreal = m.sum(1)
        nreal = torch.where(real, nreal, torch.ones_like(nreal))
        me = (m * energy).sum(1, keepdim=True) / nreal[:, None]
        mr = (m * energy_ref).sum(1, keepdim=True) / nreal[:, None]
        diff = (energy - me) - (energy_ref - mr)
        out[:, 0] = (m * diff * diff).sum(1)
        out[:, 1] = nreal * real.float()
        out[:, 2:] = 0
        if grad is not None:
            ptr = plan.atom_molptr.long()
            cnt = ptr[1:] - ptr[:-1]
            seg = torch.repeat_interleave(torch.arange(B, device=dev), cnt)
            d = grad - grad_ref
            sq = (m[seg][..., None] * d * d).sum((1, 2))
            out[:, 2] = torch.zeros(B, device=dev).index_add(0, seg, sq)
            out[:, 3] = cnt.float() * nreal * real.float()

    def loss_ef(self, plan, energy, energy_ref, is_dummy, grad, grad_ref, wE, wG, inv_B, loss_mol, gE, gG):
        B = plan.B
        dev = loss_mol.device
        Cc = energy.shape[1] if energy is not None else grad.shape[1]
        m = torch.ones((B, Cc), device=dev) if is_dummy is None else (is_dummy == 0).float()
        # a batch that ends in a padding molecule (plan.n_real_mols): the product's kernels run over the real molecules only; here the
        # padding molecule is masked out (its conformations are all dummies: no real-conformation count to divide by)
        n_real = getattr(plan, "n_real_mols", None)
        real = torch.arange(B, device=dev) < (B if n_real is None else int(n_real))
        m = m * real[:, None].float()
        nreal = torch.where(real[:, None], m.sum(1, keepdim=True), torch.ones((B, 1), device=dev))
        loss = torch.zeros(B, device=dev)
        if wE != 0:
            me = (m * energy).sum(1, keepdim=True) / nreal
            mr = (m * energy_ref).sum(1, keepdim=True) / nreal
            diff = (energy - me) - (energy_ref - mr)
            loss = loss + wE * (m * diff * diff).sum(1) / nreal[:, 0]
            if gE is not None:
                gE.copy_(inv_B * wE * 2.0 / nreal * m * (diff - (m * diff).sum(1, keepdim=True) / nreal))
        elif gE is not None:
            gE.zero_()
        if wG != 0:
            ptr = plan.atom_molptr.long()
            cnt = ptr[1:] - ptr[:-1]
            seg = torch.repeat_interleave(torch.arange(B, device=dev), cnt)
            ma = m[seg]                                                  # (N,C)
            denom = (cnt.float() * nreal[:, 0] * 3.0)
            diff = grad - grad_ref
            sq = (ma[..., None] * diff * diff).sum((1, 2))
            loss = loss + wG * torch.zeros(B, device=dev).index_add(0, seg, sq) / denom
            if gG is not None:
                gG.copy_(inv_B * wG * 2.0 * ma[..., None] * diff / denom[seg][:, None, None])
        elif gG is not None:
            gG.zero_()
        loss_mol.copy_(loss)

    def loss_param(self, plan, params, refs, fac, reg, pw, inv_B, loss_mol, gps):
        lv = ["n2", "n2", "n3", "n3", "n4", "n4_improper"]
        B = plan.B
        dev = loss_mol.device
        den = torch.zeros(B, device=dev)
        for l in range(6):
            if params[l] is not None and refs[l] is not None:
                ptr = plan.mol_ptr[lv[l]].long()
                T = plan.T[lv[l]]
                w = params[l].numel() // T if T else 1
                den = den + (ptr[1:] - ptr[:-1]).float() * w
        pwv = pw if pw is not None else torch.zeros(B, device=dev)
        n_real = getattr(plan, "n_real_mols", None)
        real = (torch.arange(B, device=dev) < (B if n_real is None else int(n_real))).float()      # (see loss_ef)
        pwv = pwv * real
        num = torch.zeros(B, device=dev)
        regsum = torch.zeros(B, device=dev)
        for l in range(6):
            p = params[l]
            if p is None:
                continue
            T = plan.T[lv[l]]
            ptr = plan.mol_ptr[lv[l]].long()
            cnt = ptr[1:] - ptr[:-1]
            seg = torch.repeat_interleave(torch.arange(B, device=dev), cnt)
            w = p.numel() // T if T else 1
            pv = p.reshape(T, w)
            g = torch.zeros_like(pv)
            if refs[l] is not None and T:
                rw = refs[l].numel() // T
                r = refs[l].reshape(T, rw)
                if rw < w:
                    r = torch.cat([r, torch.zeros((T, w - rw), device=dev)], 1)
                r = r[:, :w]
                ok = ~torch.isnan(r)
                diff = torch.where(ok, pv - torch.where(ok, r, torch.zeros_like(r)), torch.zeros_like(pv))
                active = (pwv != 0)[seg][:, None]
                num = num + torch.zeros(B, device=dev).index_add(0, seg, (fac[l] ** 2 * diff * diff * active).sum(1))
                g = g + torch.where(active, pwv[seg][:, None] * fac[l] ** 2 * 2.0 * diff / den[seg][:, None], torch.zeros_like(g))
            if reg[l] > 0 and T:
                c = (cnt.float() * w)
                regc = torch.where(cnt > 0, reg[l] / c.clamp(min=1), torch.zeros_like(c)) * real
                regsum = regsum + regc * torch.zeros(B, device=dev).index_add(0, seg, (pv * pv).sum(1))
                g = g + regc[seg][:, None] * 2.0 * pv
            if gps[l] is not None:
                gps[l].copy_((inv_B * g).reshape(gps[l].shape))
        loss = regsum + torch.where((pwv != 0) & (den > 0), pwv * num / den.clamp(min=1), torch.zeros(B, device=dev))
        loss_mol.add_(loss)

    # ------------------------------------------------------------------ optimiser
    def sumsq(self, x, out, accumulate=False):
        s = (x.double() ** 2).sum().float()
        out.copy_(out + s if accumulate else s.reshape(out.shape))

    def adam_step(self, p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale, sumsq, max_norm):
        clip = 1.0
        if sumsq is not None:
            norm = torch.sqrt(sumsq.reshape(())) * grad_scale
            clip = torch.clamp(max_norm / (norm + 1e-6), max=1.0)
        gi = g * grad_scale * clip
        if weight_decay != 0:
            gi = gi + weight_decay * p
        m.copy_(beta1 * m + (1 - beta1) * gi)
        v.copy_(beta2 * v + (1 - beta2) * gi * gi)
        bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
        p.copy_(p - (lr / bc1) * (m / (v.sqrt() / math.sqrt(bc2) + eps)))
