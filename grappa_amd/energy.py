"""`Energy`: bonded MM energy and its gradient w.r.t. the coordinates, written into the graph.

Drop-in for the reference's models/energy.py:74-171 (`Energy(terms, suffix, offset_torsion, write_suffix,
gradients)`), computed by the fused HIP kernels behind `ops.MMEnergyFn` instead of ~40 small autograd ops:
  g.nodes['g'].data['energy'+ws]            (B, C)
  g.nodes['g'].data['energy_<term>'+ws]     (B, C), detached
  g.nodes[term].data['energy'+ws], ['x']    (T, C) per-tuple energies / internal coordinates (detached)
  g.nodes['n1'].data['gradient'+ws]         (N, C, 3) = + dE/dxyz, differentiable w.r.t. k, eq
Deviations from the reference: no random dihedral noise (Q1); angles use the intended
last-axis cross product (Q2); xyz.grad is not populated (Q10).
"""
from typing import List

import torch

from . import ops
from .constants import TUPLE_LEVELS


class Energy(torch.nn.Module):
    def __init__(self, terms: list = ["n2", "n3", "n4", "n4_improper"], suffix: str = "", offset_torsion: bool = False,
                 write_suffix=None, gradients: bool = True):
        super().__init__()
        if not isinstance(terms, list):
            raise ValueError("terms must be a list")
        for t in terms:
            if t not in TUPLE_LEVELS:
                raise ValueError(f"term {t} not in {TUPLE_LEVELS}")
        self.offset_torsion = offset_torsion
        self.suffix = suffix
        self.write_suffix = write_suffix if write_suffix is not None else suffix
        self.terms = terms
        self.gradients = gradients

    def forward(self, g):
        n1 = g.nodes["n1"].data
        if "xyz" not in n1:
            raise ValueError("xyz coordinates must be stored in g.nodes['n1'].data['xyz']")
        xyz = n1["xyz"].detach().float()
        plan = g.plan()
        dev = xyz.device
        Cc = xyz.shape[1]
        ks, eqs, n_per = [], [], [0, 0, 1, 1]
        for l, term in enumerate(TUPLE_LEVELS):
            T = plan.T[term]
            d = g.nodes[term].data
            if term in self.terms:
                if term not in g.ntypes:
                    raise ValueError(f"term {term} not in g.ntypes")
                if "k" + self.suffix not in d:
                    raise RuntimeError(f"{term} has no k{self.suffix} attribute")
                k = d["k" + self.suffix].float()
                if l < 2 and k.dim() != 1:
                    raise ValueError(f"k must be a 1d tensor, but has shape {k.shape}")
                eq = d["eq" + self.suffix].float() if l < 2 else None
            else:       # term switched off: zero force constants
                k = torch.zeros((T,) if l < 2 else (T, 1), dtype=torch.float32, device=dev)
                eq = torch.zeros((T,), dtype=torch.float32, device=dev) if l < 2 else None
            if l >= 2:
                k = k.reshape(T, -1) if T else k.reshape(0, max(k.shape[-1] if k.dim() == 2 else 1, 1))
                n_per[l] = max(int(k.shape[1]), 1)
            ks.append(k)
            eqs.append(eq)
        te = [torch.empty((plan.T[t], Cc), dtype=torch.float32, device=dev) for t in TUPLE_LEVELS]
        tx = [torch.empty((plan.T[t], Cc), dtype=torch.float32, device=dev) for t in TUPLE_LEVELS]
        energy, terms, grad = ops.MMEnergyFn.apply(xyz, plan, n_per, bool(self.offset_torsion), bool(self.gradients), (te, tx),
                                                   ks[0], eqs[0], ks[1], eqs[1], ks[2], ks[3])
        ws = self.write_suffix
        gd = g.nodes["g"].data
        gd["energy" + ws] = energy
        for l, term in enumerate(TUPLE_LEVELS):
            if term in self.terms:
                gd["energy_" + term + ws] = terms[l]
                g.nodes[term].data["energy" + ws] = te[l]
                g.nodes[term].data["x"] = tx[l]
        if self.gradients:
            n1["gradient" + ws] = grad
        return g
