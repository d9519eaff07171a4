"""`GrappaModel`: drop-in for the reference's `grappa.models.GrappaModel` (models/grappa.py:7-132) whose
forward runs on the hand-written HIP kernels of libgrappa_hip.so.

The module tree only HOLDS parameters, with the reference's attribute names so that state dicts are
interchangeable key for key (`gnn.pre_dense.0.weight`, `gnn.att_blocks.{i}.graph_module.fc.weight`, the
aliased `gnn.blocks.{i}.*`, `parameter_writer.bond_writer.bond_model.grappa_transformer.transformer.{l}.attn.in_proj_weight`,
`...symmetriser.mlp.{j}.linear1.weight`, buffers `to_k.mean_over_std`, `k_std`, ... -- SURVEY.md section 8(b)).
torch.nn.Linear / LayerNorm / MultiheadAttention instances are used as parameter containers (same
initialisation as the reference); their forward() is never called.  All arithmetic goes through
grappa_amd.ops (block-level autograd nodes -> C ABI).

Constructor options as in the reference (models/grappa.py:51): layer_norm=False / self_interaction=False drop the same sub-modules the
reference drops, learnable_statistics=True turns the same statistics into parameters (same state-dict keys either way).
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Union

import torch
import torch.nn as nn

from . import ops
from .backend import get_backend
from .constants import CHARGE_ENCODING_DIM, DEFAULT_FEAT_DIMS, get_default_statistics


# ------------------------------------------------------------------------------------------------ GNN
class _GraphFC(nn.Module):
    """holds DotGatConv's single bias-free projection as `fc.weight` (DGL naming)."""

    def __init__(self, in_feats, out_feats, num_heads):
        super().__init__()
        self.fc = nn.Linear(in_feats, out_feats * num_heads, bias=False)


class _SageParams(nn.Module):
    """DGL SAGEConv('mean') parameter names: fc_self.weight, fc_neigh.weight, bias."""

    def __init__(self, in_feats, out_feats):
        super().__init__()
        self.fc_neigh = nn.Linear(in_feats, out_feats, bias=False)
        self.fc_self = nn.Linear(in_feats, out_feats, bias=False)
        self.bias = nn.Parameter(torch.zeros(out_feats))
        gain = nn.init.calculate_gain("relu")
        nn.init.xavier_uniform_(self.fc_self.weight, gain=gain)
        nn.init.xavier_uniform_(self.fc_neigh.weight, gain=gain)


def _wb(norm):
    """(weight, bias) of an optional LayerNorm"""
    return (None, None) if norm is None else (norm.weight, norm.bias)


class ResidualAttentionBlock(nn.Module):
    """reference models/graph_attention.py:226-310; the same sub-modules exist under the same names for the same options
    (layer_norm=False: no layer_norm / interaction_norm; self_interaction=False: no self_interaction, no interaction_norm)"""

    def __init__(self, in_feats, num_heads, dropout, layer_norm=True, self_interaction=True):
        super().__init__()
        assert in_feats % num_heads == 0
        self.num_heads, self.p = num_heads, float(dropout)
        self.graph_module = _GraphFC(in_feats, in_feats // num_heads, num_heads)
        if layer_norm:
            self.layer_norm = nn.LayerNorm(in_feats)
        self.head_reducer = nn.Linear(in_feats, in_feats)
        if self_interaction:
            if layer_norm:
                self.interaction_norm = nn.LayerNorm(in_feats)
            self.self_interaction = nn.Sequential(nn.Linear(in_feats, 4 * in_feats), nn.ELU(), nn.Linear(4 * in_feats, in_feats), nn.ELU())
        else:
            self.self_interaction = None

    def forward(self, plan, h):
        p = self.p if self.training else 0.0
        s1, s2 = (ops.next_seed(), ops.next_seed()) if p > 0 else (0, 0)
        si = self.self_interaction
        si_params = (None,) * 4 if si is None else (si[0].weight, si[0].bias, si[2].weight, si[2].bias)
        ops.mark_mode()
        return ops.AttBlockFn.apply(h, plan, self.num_heads, p, s1, s2, *_wb(getattr(self, "layer_norm", None)),
                                    self.graph_module.fc.weight, self.head_reducer.weight, self.head_reducer.bias,
                                    *_wb(getattr(self, "interaction_norm", None)), *si_params)


class ResidualConvBlock(nn.Module):
    """reference models/graph_attention.py:343-415"""

    def __init__(self, in_feats, dropout, layer_norm=True, self_interaction=True):
        super().__init__()
        self.p = float(dropout)
        self.graph_module = _SageParams(in_feats, in_feats)
        if layer_norm:
            self.layer_norm = nn.LayerNorm(in_feats)
        if self_interaction:
            self.self_interaction = nn.Sequential(nn.Linear(in_feats, in_feats), nn.ELU())
            if layer_norm:
                self.interaction_norm = nn.LayerNorm(in_feats)
        else:
            self.self_interaction = None

    def forward(self, plan, h):
        p = self.p if self.training else 0.0
        s1, s2 = (ops.next_seed(), ops.next_seed()) if p > 0 else (0, 0)
        gm = self.graph_module
        si = self.self_interaction
        si_params = (None, None) if si is None else (si[0].weight, si[0].bias)
        return ops.ConvBlockFn.apply(h, plan, p, s1, s2, *_wb(getattr(self, "layer_norm", None)), gm.fc_self.weight,
                                     gm.fc_neigh.weight, gm.bias, *_wb(getattr(self, "interaction_norm", None)), *si_params)


class GrappaGNN(nn.Module):
    """reference models/graph_attention.py:11-183"""

    def __init__(self, out_feats=512, in_feats=None, node_feats=None, n_conv=3, n_att=3, n_heads=8,
                 in_feat_name=("atomic_number", "ring_encoding", "partial_charge"), in_feat_dims={}, conv_dropout=0.,
                 attention_dropout=0., final_dropout=0., initial_dropout=0., layer_norm=True, self_interaction=True, charge_encoding=True):
        super().__init__()
        if not isinstance(in_feat_name, (list, tuple)):
            in_feat_name = [in_feat_name]
        self.in_feat_name = list(in_feat_name)
        dims = dict(DEFAULT_FEAT_DIMS)
        dims.update(in_feat_dims)
        if in_feats is None:
            in_feats = sum(dims[f] for f in self.in_feat_name)
        if node_feats is None:
            node_feats = out_feats
        self.charge_encoding = charge_encoding
        self.in_feats = in_feats + (CHARGE_ENCODING_DIM if charge_encoding else 0)
        self.p_initial, self.p_final = float(initial_dropout), float(final_dropout)
        self.pre_dense = nn.Sequential(nn.Linear(self.in_feats, node_feats), nn.ELU())
        self.no_convs = (n_conv + n_att) == 0
        if not self.no_convs:
            self.conv_blocks = nn.ModuleList([ResidualConvBlock(node_feats, conv_dropout, layer_norm, self_interaction) for _ in range(n_conv)])
            self.att_blocks = nn.ModuleList([ResidualAttentionBlock(node_feats, n_heads, attention_dropout, layer_norm, self_interaction)
                                             for _ in range(n_att)])
        self.post_dense = nn.Sequential(nn.Linear(node_feats, out_feats))
        if not self.no_convs:
            self.blocks = self.conv_blocks + self.att_blocks       # same aliasing as the reference (state-dict keys twice)

    def input_features(self, g) -> torch.Tensor:
        d = g.nodes["n1"].data
        cols = [d[f].float() if d[f].dim() >= 2 else d[f].unsqueeze(-1).float() for f in self.in_feat_name]
        n_plain = sum(c.shape[1] for c in cols)
        N = cols[0].shape[0]
        # rows of a multiple of four floats (the weight-gradient product of pre_dense reads them in 16-byte pieces: ops._padded_cols)
        x = torch.zeros((N, (self.in_feats + 3) // 4 * 4), dtype=torch.float32, device=cols[0].device)[:, :self.in_feats]
        if n_plain + (CHARGE_ENCODING_DIM if self.charge_encoding else 0) != self.in_feats:
            raise AssertionError(f"the input features must have {self.in_feats} columns in total, got {n_plain}")
        torch.cat(cols, dim=-1, out=x[:, :n_plain])
        if self.charge_encoding:
            get_backend().charge_encoding(d["partial_charge"].float().contiguous(), CHARGE_ENCODING_DIM, -2.0, 2.0, x, n_plain)
        return x

    def forward(self, g):
        plan = g.plan()
        x = self.input_features(g)
        p0 = self.p_initial if self.training else 0.0
        h = ops.LinearFn.apply(x, self.pre_dense[0].weight, self.pre_dense[0].bias, ops.ELU, p0, ops.next_seed() if p0 > 0 else 0, ops.act_dtype())
        if not self.no_convs:
            for blk in self.blocks:
                h = blk(plan, h)
        p1 = self.p_final if self.training else 0.0
        h = ops.LinearFn.apply(h, self.post_dense[0].weight, self.post_dense[0].bias, 0, p1, ops.next_seed() if p1 > 0 else 0, torch.float32)
        g.nodes["n1"].data["h"] = h
        return g


# ------------------------------------------------------------------------------------------------ writers
class FeedForwardLayer(nn.Module):
    """parameter holder: linear1, linear2, norm1 (reference models/network_utils.py:5-54)"""

    def __init__(self, in_feats, hidden_feats, out_feats, layer_norm=True):
        super().__init__()
        self.linear1 = nn.Linear(in_feats, hidden_feats)
        self.linear2 = nn.Linear(hidden_feats, out_feats)
        if layer_norm:
            self.norm1 = nn.LayerNorm(in_feats)

    def params(self):
        return (*_wb(getattr(self, "norm1", None)), self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias)


class DottedAttWithMLP(nn.Module):
    def __init__(self, n_feats, num_heads, hidden_feats, dropout, layer_norm=True):
        super().__init__()
        assert n_feats % num_heads == 0, f"Number of features ({n_feats}) must be divisible by the number of heads ({num_heads})."
        self.num_heads, self.p = num_heads, float(dropout)
        if layer_norm:
            self.norm1 = nn.LayerNorm(n_feats)
        self.attn = nn.MultiheadAttention(n_feats, num_heads, dropout=0)
        self.ff = FeedForwardLayer(n_feats, hidden_feats, n_feats, layer_norm)

    def forward(self, x, s, T):
        p = self.p if self.training else 0.0
        s1, s2 = (ops.next_seed(), ops.next_seed()) if p > 0 else (0, 0)
        ops.mark_mode()
        return ops.TransformerLayerFn.apply(x, s, T, self.num_heads, p, s1, s2, *_wb(getattr(self, "norm1", None)),
                                            self.attn.in_proj_weight, self.attn.in_proj_bias, self.attn.out_proj.weight,
                                            self.attn.out_proj.bias, *self.ff.params())


class GrappaTransformer(nn.Module):
    def __init__(self, n_feats, n_heads, hidden_feats, n_layers, positional_encoding: Optional[torch.Tensor], dropout, layer_norm=True):
        super().__init__()
        if positional_encoding is not None:
            self.register_buffer("positional_encoding", positional_encoding.float())
            n_feats = n_feats + positional_encoding.shape[1]
        else:
            self.positional_encoding = None
        if n_feats % n_heads:
            raise ValueError(f"The number of input features cannot be divided by the number of heads: {n_feats} / {n_heads}")
        self.n_feats = n_feats
        self.transformer = nn.Sequential(*[DottedAttWithMLP(n_feats, n_heads, hidden_feats, dropout, layer_norm) for _ in range(n_layers)])


class Symmetriser(nn.Module):
    def __init__(self, in_feats, out_feats, permutations: torch.Tensor, hidden_feats, n_layers, layer_norm=True):
        super().__init__()
        assert n_layers >= 1, "n_layers must be >= 1"
        P, s = permutations.shape
        assert torch.all(permutations[0].int() == torch.arange(s).int()), "permutations must include the identity permutation at the zeroth entry."
        self.register_buffer("permutation_prefactors", torch.ones(P, dtype=torch.float32).view(P, 1, 1))
        self.register_buffer("permutations", permutations.int())
        self.n_seq, self.out_feats = s, out_feats
        layers = [FeedForwardLayer(in_feats * s, hidden_feats, hidden_feats if n_layers > 1 else out_feats, layer_norm)]
        for i in range(1, n_layers):
            layers.append(FeedForwardLayer(hidden_feats, hidden_feats, out_feats if i == n_layers - 1 else hidden_feats, layer_norm))
        self.mlp = nn.Sequential(*layers)
        self._perm_list = [tuple(int(v) for v in p) for p in permutations.tolist()]


class SymmetrisedTransformer(nn.Module):
    def __init__(self, n_feats, n_heads, hidden_feats, n_layers, out_feats, permutations, dropout, symmetriser_layers,
                 symmetriser_hidden_feats, positional_encoding, layer_norm=True):
        super().__init__()
        if n_layers > 0:
            self.grappa_transformer = GrappaTransformer(n_feats, n_heads, hidden_feats, n_layers, positional_encoding, dropout, layer_norm)
            width = self.grappa_transformer.n_feats
        else:
            self.grappa_transformer = None
            width = n_feats
        self.symmetriser = Symmetriser(width, out_feats, permutations, symmetriser_hidden_feats, symmetriser_layers, layer_norm)

    def forward(self, x, s, T, first_layer_done=False):
        """x: (s*T, F) token table (row = pos*T + t) -> (P*T, out_feats), one row per permuted copy.  first_layer_done: x is already
        the output of the first transformer layer (ops.ProjFirstLayerFn)."""
        if self.grappa_transformer is not None:
            for li, layer in enumerate(self.grappa_transformer.transformer):
                if li == 0 and first_layer_done:
                    continue
                x = layer(x, s, T)
        sym = self.symmetriser
        flat = [t for ff in sym.mlp for t in ff.params()]
        ops.mark_mode()
        return ops.SymmetriserFn.apply(x, s, T, sym._perm_list, len(sym.mlp), *flat)


class RepProjector(nn.Module):
    def __init__(self, dim_tupel, in_feats, out_feats, improper=False):
        super().__init__()
        self.dim_tupel, self.improper = dim_tupel, improper
        self.mlp = nn.Sequential(nn.Linear(in_feats, out_feats), nn.ELU())


class ToPositive(nn.Module):
    """parameter holder (reference models/final_layer.py:11-51): learnable_statistics turns mean_over_std and std into parameters"""

    def __init__(self, mean, std, min_=0., learnable_statistics=False):
        super().__init__()
        if learnable_statistics:
            self.mean_over_std = nn.Parameter(torch.tensor(float(mean / std)).float())
            self.std = nn.Parameter(torch.tensor(float(std)).float())
        else:
            self.register_buffer("mean_over_std", torch.tensor(float(mean / std)))
            self.register_buffer("std", torch.tensor(float(std)))
        self.register_buffer("min_", torch.tensor(float(min_)))


class ToRange(nn.Module):
    """reference models/final_layer.py:53-96: learnable_statistics turns std_over_max into a parameter"""

    def __init__(self, max_, std, learnable_statistics=False):
        super().__init__()
        if learnable_statistics:
            self.std_over_max = nn.Parameter(torch.tensor(float(std / max_)).float())
        else:
            self.register_buffer("std_over_max", torch.tensor(float(std / max_)).float())
        self.register_buffer("max", torch.tensor(float(max_)).float())


def _pos_enc(s, positional_encoding, wrong_symmetry=False):
    if not positional_encoding or s == 2:
        return None
    if s == 3:
        return torch.tensor([[0.], [1.], [0.]])
    if wrong_symmetry:
        return torch.tensor([[0.], [0.], [1.], [0.]])
    return torch.tensor([[0.], [1.], [1.], [0.]])


class _WriterBase(nn.Module):
    level = "n2"
    s = 2
    kind = 0

    def _model(self) -> SymmetrisedTransformer:
        raise NotImplementedError

    def _consts(self) -> torch.Tensor:
        raise NotImplementedError

    def _tokens(self, g):
        """atom embeddings -> tokens, through the first transformer layer where that runs on (atom, position) rows:
        (x (s*T, F), T, transformer layers still to run)"""
        plan = g.plan()
        lvl = self.level
        h = g.nodes["n1"].data["h"]
        model = self._model()
        T, N = plan.T[lvl], plan.N
        pe = None
        layers = []
        if model.grappa_transformer is not None:
            layers = list(model.grappa_transformer.transformer)
            if model.grappa_transformer.positional_encoding is not None:
                pe = model.grappa_transformer.positional_encoding.reshape(-1).contiguous()
        lin = self.rep_projector.mlp[0]
        if ops.FIRST_LAYER_ON_ATOM_ROWS and layers and T > 0 and 4 * N <= 3 * T:
            l0 = layers[0]
            p = l0.p if l0.training else 0.0
            s1, s2 = (ops.next_seed(), ops.next_seed()) if p > 0 else (0, 0)
            ops.mark_mode()
            x = ops.ProjFirstLayerFn.apply(h, lin.weight, lin.bias, plan.position_tables(lvl), self.s, T, pe, ops.act_dtype(), l0.num_heads, p, s1, s2,
                                           *_wb(getattr(l0, "norm1", None)), l0.attn.in_proj_weight, l0.attn.in_proj_bias,
                                           l0.attn.out_proj.weight, l0.attn.out_proj.bias, *l0.ff.params())
            return x, T, layers[1:]
        x = ops.ProjGatherFn.apply(h, lin.weight, lin.bias, plan.idx32[lvl], plan.inv_ptr[lvl], plan.inv_rows[lvl], self.s, pe, ops.act_dtype())
        return x, T, layers

    def _write(self, g, o, T):
        """the output map of the symmetrised features o -> k (, eq) in the graph (the tail of forward)"""
        raise NotImplementedError

    def _symmetrised(self, g):
        """atom embeddings -> tokens -> transformer -> symmetriser: (o (P*T, out_feats), T)"""
        plan = g.plan()
        lvl = self.level
        h = g.nodes["n1"].data["h"]
        model = self._model()
        T, N = plan.T[lvl], plan.N
        pe = None
        layers = []
        if model.grappa_transformer is not None:
            layers = list(model.grappa_transformer.transformer)
            if model.grappa_transformer.positional_encoding is not None:
                pe = model.grappa_transformer.positional_encoding.reshape(-1).contiguous()
        lin = self.rep_projector.mlp[0]
        # far fewer (atom, position) pairs than tokens (propers, angles): the first layer's LayerNorm and QKV product run on those
        if ops.FIRST_LAYER_ON_ATOM_ROWS and layers and T > 0 and 4 * N <= 3 * T:
            l0 = layers[0]
            p = l0.p if l0.training else 0.0
            s1, s2 = (ops.next_seed(), ops.next_seed()) if p > 0 else (0, 0)
            ops.mark_mode()
            x = ops.ProjFirstLayerFn.apply(h, lin.weight, lin.bias, plan.position_tables(lvl), self.s, T, pe, ops.act_dtype(), l0.num_heads, p, s1, s2,
                                           *_wb(getattr(l0, "norm1", None)), l0.attn.in_proj_weight, l0.attn.in_proj_bias,
                                           l0.attn.out_proj.weight, l0.attn.out_proj.bias, *l0.ff.params())
            return model(x, self.s, T, first_layer_done=True), T
        x = ops.ProjGatherFn.apply(h, lin.weight, lin.bias, plan.idx32[lvl], plan.inv_ptr[lvl], plan.inv_rows[lvl], self.s, pe, ops.act_dtype())
        return model(x, self.s, T), T


class WriteBondParameters(_WriterBase):
    level, s, kind = "n2", 2, 0

    def __init__(self, rep_feats, between_feats, suffix, stats, n_att, n_heads, dense_layers, dropout, symmetriser_feats, gate, layer_norm=True, learnable_statistics=False):
        super().__init__()
        eps = 1e-6
        self.suffix, self.gate = suffix, gate
        self.rep_projector = RepProjector(2, rep_feats, between_feats)
        self.bond_model = SymmetrisedTransformer(between_feats, n_heads, between_feats, n_att, 2 + int(gate),
                                                 torch.tensor([[0, 1], [1, 0]], dtype=torch.int32), dropout, dense_layers,
                                                 symmetriser_feats, None, layer_norm)
        self.to_k = ToPositive(stats["mean"]["n2_k"].item(), stats["std"]["n2_k"].item() + eps, 0., learnable_statistics)
        self.to_eq = ToPositive(stats["mean"]["n2_eq"].item(), stats["std"]["n2_eq"].item() + eps, 0., learnable_statistics)

    def _model(self):
        return self.bond_model

    def _consts(self):
        return torch.stack([self.to_eq.mean_over_std, self.to_eq.std, self.to_eq.min_, self.to_k.mean_over_std, self.to_k.std,
                            self.to_k.min_]).float().contiguous()

    def _write(self, g, o, T):
        k, eq = ops.ParamOutFn.apply(o, 0, T, 2, 0, False, 0.0, self._consts())
        g.nodes["n2"].data["eq" + self.suffix] = eq
        g.nodes["n2"].data["k" + self.suffix] = k           # harmonic_gate has no effect on the outputs (reference quirk Q3)
        return g

    def forward(self, g):
        o, T = self._symmetrised(g)
        return self._write(g, o, T)


class WriteAngleParameters(_WriterBase):
    level, s, kind = "n3", 3, 1

    def __init__(self, rep_feats, between_feats, suffix, stats, n_att, n_heads, dense_layers, dropout, symmetriser_feats,
                 positional_encoding, gate, layer_norm=True, learnable_statistics=False):
        super().__init__()
        eps = 1e-6
        self.suffix, self.gate = suffix, gate
        proj = between_feats - 1 if positional_encoding else between_feats
        self.rep_projector = RepProjector(3, rep_feats, proj)
        self.angle_model = SymmetrisedTransformer(proj, n_heads, between_feats, n_att, 2 + int(gate),
                                                  torch.tensor([[0, 1, 2], [2, 1, 0]], dtype=torch.int32), dropout, dense_layers,
                                                  symmetriser_feats, _pos_enc(3, positional_encoding), layer_norm)
        self.to_k = ToPositive(stats["mean"]["n3_k"].item(), stats["std"]["n3_k"].item() + eps, 0., learnable_statistics)
        self.to_eq = ToRange(math.pi, stats["std"]["n3_eq"].item() + eps, learnable_statistics)

    def _model(self):
        return self.angle_model

    def _consts(self):
        z = torch.zeros((), dtype=torch.float32, device=self.to_k.std.device)
        return torch.stack([self.to_eq.std_over_max, self.to_eq.max, z, self.to_k.mean_over_std, self.to_k.std, self.to_k.min_]).float().contiguous()

    def _write(self, g, o, T):
        k, eq = ops.ParamOutFn.apply(o, 1, T, 2, 0, False, 0.0, self._consts())
        g.nodes["n3"].data["eq" + self.suffix] = eq
        g.nodes["n3"].data["k" + self.suffix] = k
        return g

    def forward(self, g):
        if "n3" not in g.ntypes:
            return g
        o, T = self._symmetrised(g)
        return self._write(g, o, T)


class WriteTorsionParameters(_WriterBase):
    s, kind = 4, 2

    def __init__(self, rep_feats, between_feats, suffix, n_periodicity, improper, n_att, n_heads, dense_layers, dropout,
                 symmetriser_feats, stats, positional_encoding, gated, wrong_symmetry, cutoff, layer_norm=True, learnable_statistics=False):
        super().__init__()
        eps = 1e-1 if gated else 1e-2
        self.gated, self.improper, self.suffix, self.cutoff_value = gated, improper, suffix, float(cutoff)
        self.level = "n4_improper" if improper else "n4"
        self.register_buffer("n_periodicity", torch.tensor(n_periodicity).long())
        self._n_per = int(n_periodicity)
        if not improper:
            km, ks = stats["mean"]["n4_k"], stats["std"]["n4_k"] + eps
        elif "n4_improper_k" not in stats["mean"]:
            km, ks = torch.zeros(n_periodicity), torch.ones(n_periodicity)
        else:
            km, ks = stats["mean"]["n4_improper_k"], stats["std"]["n4_improper_k"] + eps
            if len(km) < n_periodicity or len(ks) < n_periodicity:
                raise ValueError(f"n_periodicity is {n_periodicity} but the param_statistics contains {len(km)} values for the improper torsion parameters.")
        if learnable_statistics:                    # reference interaction_parameters.py:465-470
            self.k_mean = nn.Parameter(km[:n_periodicity].clone().float().unsqueeze(0))
            self.k_std = nn.Parameter(ks[:n_periodicity].clone().float().unsqueeze(0))
        else:
            self.register_buffer("k_mean", km[:n_periodicity].clone().float().unsqueeze(0))
            self.register_buffer("k_std", ks[:n_periodicity].clone().float().unsqueeze(0))
        proj = between_feats - 1 if positional_encoding else between_feats
        self.rep_projector = RepProjector(4, rep_feats, proj, improper=improper)
        pe = _pos_enc(4, positional_encoding)
        if not improper:
            perms = [[0, 1, 2, 3], [3, 2, 1, 0]]
        elif wrong_symmetry:
            perms = [[0, 1, 2, 3], [3, 1, 2, 0], [1, 3, 2, 0], [0, 3, 2, 1], [3, 0, 2, 1], [1, 0, 2, 3]]
            pe = _pos_enc(4, True, True)
        else:
            perms = [[0, 1, 2, 3], [3, 1, 2, 0]]
        self._P = len(perms)
        n_out = 2 * n_periodicity if gated else n_periodicity
        self.torsion_model = SymmetrisedTransformer(proj, n_heads, between_feats, n_att, n_out, torch.tensor(perms, dtype=torch.int32),
                                                    dropout, dense_layers, symmetriser_feats, pe, layer_norm)

    def _model(self):
        return self.torsion_model

    def _consts(self):
        return torch.cat([self.k_std.reshape(-1), self.k_mean.reshape(-1)]).float().contiguous()

    def _write(self, g, o, T):
        k = ops.ParamOutFn.apply(o, 2, T, self._P, self._n_per, self.gated, self.cutoff_value, self._consts())
        g.nodes[self.level].data["k" + self.suffix] = k
        return g

    def forward(self, g):
        lvl = self.level
        if lvl not in g.ntypes:
            return g
        o, T = self._symmetrised(g)
        return self._write(g, o, T)


class WriteParameters(nn.Module):
    """reference models/interaction_parameters.py:10-135"""

    def __init__(self, cfg: Dict, stats: Dict, suffix=""):
        super().__init__()
        rep, drop, pos = cfg["graph_node_features"], cfg["parameter_dropout"], cfg["positional_encoding"]
        gate = cfg["harmonic_gate"]
        self.bond_writer = WriteBondParameters(rep, cfg["bond_transformer_width"], suffix, stats, cfg["bond_transformer_depth"],
                                               cfg["bond_n_heads"], cfg["bond_symmetriser_depth"], drop, cfg["bond_symmetriser_width"], gate,
                                               layer_norm=cfg.get("layer_norm", True), learnable_statistics=cfg.get("learnable_statistics", False))
        self.angle_writer = WriteAngleParameters(rep, cfg["angle_transformer_width"], suffix, stats, cfg["angle_transformer_depth"],
                                                 cfg["angle_n_heads"], cfg["angle_symmetriser_depth"], drop,
                                                 cfg["angle_symmetriser_width"], pos, gate, layer_norm=cfg.get("layer_norm", True), learnable_statistics=cfg.get("learnable_statistics", False))
        self.proper_writer = WriteTorsionParameters(rep, cfg["proper_transformer_width"], suffix, cfg["n_periodicity_proper"], False,
                                                    cfg["proper_transformer_depth"], cfg["proper_n_heads"], cfg["proper_symmetriser_depth"],
                                                    drop, cfg["proper_symmetriser_width"], stats, pos, cfg["gated_torsion"], False,
                                                    cfg["torsion_cutoff"], layer_norm=cfg.get("layer_norm", True), learnable_statistics=cfg.get("learnable_statistics", False))
        self.improper_writer = WriteTorsionParameters(rep, cfg["improper_transformer_width"], suffix, cfg["n_periodicity_improper"], True,
                                                      cfg["improper_transformer_depth"], cfg["improper_n_heads"],
                                                      cfg["improper_symmetriser_depth"], drop, cfg["improper_symmetriser_width"], stats, pos,
                                                      cfg["gated_torsion"], cfg["wrong_symmetry"], cfg["torsion_cutoff"],
                                                      layer_norm=cfg.get("layer_norm", True), learnable_statistics=cfg.get("learnable_statistics", False))

        # The four writers read the same atom embedding and write disjoint tuple levels: on the GPU each runs on its own HIP stream
        # (largest first), so that the tail rounds and launch gaps of one head's kernels are filled by another head's; autograd replays
        # every backward node on the stream of its forward, which gives the same overlap in the backward pass (-1.7 .. -2.0 ms of a 38 ms
        # C2 step).  Default since round 3 (GRAPPA_HEAD_STREAMS=1 puts everything back on one stream).  History: a
        # since-removed LayerNorm kernel computed deviating rows beside MFMA wavefronts of another queue; the trigger -- packed fp32
        # instructions -- is compiled out of this library, and the gradients of h are joined by the library's own kernel
        # (ops.SplitHeadsFn), so no torch arithmetic kernel runs on a side stream.
        self.head_streams = int(os.environ.get("GRAPPA_HEAD_STREAMS", "4"))
        self._streams = None
        # round 4: the heads LAYER-LOCKED on one stream -- every product of a transformer / symmetriser layer is ONE grouped launch over the
        # heads (ops.MultiTransformerLayerFn, backend.gemm_group): what the four streams approximated, without their launch count.
        # GRAPPA_MERGED_HEADS: "0" (default) never, "1" always, "auto" = when a head's tokens would not fill the chip by themselves (fewer
        # than `merged_heads_max_tokens` tokens in the largest head).  Measured (round 4): the layer-locked heads cut
        # the launches per batch-32 step from 681 to 487 and the products' time on one queue by 6 %, but lose to four streams on the wall
        # clock in every regime tried -- C2 37.2 against 35.8 ms, batch 32 eager 18.4 against 16.4 ms (the Python that builds the grouped
        # calls costs more than the launches it saves), recorded 9.5 against 9.0 ms, recorded predict 3.5 against 3.0 ms -- hence opt-in
        self.merged_heads = os.environ.get("GRAPPA_MERGED_HEADS", "0")
        self.merged_heads_max_tokens = int(os.environ.get("GRAPPA_MERGED_HEADS_MAX_TOKENS", "40000"))

    def _writers_largest_first(self):
        return [self.proper_writer, self.angle_writer, self.improper_writer, self.bond_writer]

    def forward(self, g):
        h = g.nodes["n1"].data["h"]
        writers = self._writers_largest_first()
        # every head reads an alias of h of its own, so that its gradient of h arrives alone at the node that adds the four
        aliases = ops.SplitHeadsFn.apply(h, len(writers)) if (torch.is_grad_enabled() and h.requires_grad) else (h,) * len(writers)
        try:
            if self._use_merged(g, h):
                return self._forward_merged(g, writers, aliases)
            if self.head_streams <= 1 or not h.is_cuda:
                for w, a in zip(writers, aliases):          # same host order (hence dropout seeds) as the multi-stream path
                    g.nodes["n1"].data["h"] = a
                    g = w(g)
                return g
            main = torch.cuda.current_stream(h.device)
            if self._streams is None or self._streams[0].device != h.device:
                self._streams = [torch.cuda.Stream(device=h.device) for _ in range(min(self.head_streams, 4) - 1)]
            lanes = [main] + self._streams                     # the largest head stays on the caller's stream
            for s in self._streams:
                s.wait_stream(main)
            for i, (w, a) in enumerate(zip(writers, aliases)):
                with torch.cuda.stream(lanes[i % len(lanes)]):
                    g.nodes["n1"].data["h"] = a
                    g = w(g)
            for s in self._streams:
                main.wait_stream(s)
            # the parameters were allocated on the side streams and are consumed on the caller's: tell the caching allocator
            for lvl in ("n2", "n3", "n4", "n4_improper"):
                for key, t in g.nodes[lvl].data.items():
                    if key.startswith(("k", "eq")) and torch.is_tensor(t) and t.is_cuda:
                        t.record_stream(main)
            return g
        finally:
            g.nodes["n1"].data["h"] = h


def _use_merged(self, g, h) -> bool:
    if self.merged_heads in ("0", "") or ops.act_dtype() is not None:
        return False
    if self.merged_heads == "1":
        return True
    plan = g.plan()
    return h.is_cuda and max(w.s * plan.T[w.level] for w in self._writers_largest_first()) < self.merged_heads_max_tokens


def _forward_merged(self, g, writers, aliases):
    """the heads layer-locked: tokens head by head, then transformer layer li of every head that still has one as ONE node, the symmetrisers
    as one node, the output maps head by head.  Dropout seeds are drawn in the order of the head-by-head path (same masks)."""
    plan = g.plan()
    st = []
    for w, a in zip(writers, aliases):
        if w.level not in g.ntypes:
            continue
        if plan.T[w.level] == 0:                                # an empty level: the head-by-head path (zero-row tensors, no launches)
            g.nodes["n1"].data["h"] = a
            g = w(g)
            continue
        g.nodes["n1"].data["h"] = a
        x, T, left = w._tokens(g)
        model = w._model()
        nl = len(model.grappa_transformer.transformer) if model.grappa_transformer is not None else 0
        seeds = []
        for layer in left:
            p = layer.p if layer.training else 0.0
            seeds.append((p,) + ((ops.next_seed(), ops.next_seed()) if p > 0 else (0, 0)))
        st.append(dict(w=w, x=x, T=T, left=left, first=nl - len(left), seeds=seeds, model=model))
    depth = max((s_["first"] + len(s_["left"]) for s_ in st), default=0)
    for li in range(depth):
        act = [s_ for s_ in st if s_["first"] <= li < s_["first"] + len(s_["left"])]
        if not act:
            continue
        cfgs, flat = [], []
        for s_ in act:
            layer = s_["left"][li - s_["first"]]
            p, s1, s2 = s_["seeds"][li - s_["first"]]
            cfgs.append((s_["w"].s, s_["T"], layer.num_heads, p, s1, s2))
            flat += [s_["x"], *_wb(getattr(layer, "norm1", None)), layer.attn.in_proj_weight, layer.attn.in_proj_bias, layer.attn.out_proj.weight,
                     layer.attn.out_proj.bias, *layer.ff.params()]
        ops.mark_mode()
        outs = ops.MultiTransformerLayerFn.apply(tuple(cfgs), *flat)
        for s_, o in zip(act, outs):
            s_["x"] = o
    if st:
        cfgs, flat = [], []
        for s_ in st:
            sym = s_["model"].symmetriser
            cfgs.append((s_["w"].s, s_["T"], sym._perm_list, len(sym.mlp)))
            flat += [s_["x"]] + [t for ff in sym.mlp for t in ff.params()]
        ops.mark_mode()
        outs = ops.MultiSymmetriserFn.apply(tuple(cfgs), *flat)
        for s_, o in zip(st, outs):
            g = s_["w"]._write(g, o, s_["T"])
    return g


WriteParameters._use_merged = _use_merged
WriteParameters._forward_merged = _forward_merged


class GrappaModel(nn.Module):
    """Same constructor keywords, defaults and `forward(g) -> g` contract as the reference (models/grappa.py:51, :111-132)."""

    def __init__(self, graph_node_features: int = 512, in_feats: int = None,
                 in_feat_name: Union[str, List[str]] = ["atomic_number", "ring_encoding", "partial_charge"],
                 in_feat_dims: Dict[str, int] = {}, gnn_width: int = None, gnn_attentional_layers: int = 3, gnn_convolutions: int = 3,
                 gnn_attention_heads: int = 8, gnn_dropout_attention: float = 0., gnn_dropout_initial: float = 0.,
                 gnn_dropout_conv: float = 0., gnn_dropout_final: float = 0., parameter_dropout: float = 0.,
                 bond_transformer_depth=2, bond_n_heads=8, bond_transformer_width=512, bond_symmetriser_depth=2, bond_symmetriser_width=256,
                 angle_transformer_depth=2, angle_n_heads=8, angle_transformer_width=512, angle_symmetriser_depth=2, angle_symmetriser_width=256,
                 proper_transformer_depth=2, proper_n_heads=8, proper_transformer_width=512, proper_symmetriser_depth=2, proper_symmetriser_width=256,
                 improper_transformer_depth=2, improper_n_heads=8, improper_transformer_width=512, improper_symmetriser_depth=2,
                 improper_symmetriser_width=256, n_periodicity_proper=6, n_periodicity_improper=3, gated_torsion: bool = False,
                 wrong_symmetry=False, positional_encoding=True, layer_norm=True, self_interaction=True, learnable_statistics: bool = False,
                 param_statistics: dict = None, torsion_cutoff=1.e-4, harmonic_gate: bool = False):
        super().__init__()
        if param_statistics is None:
            param_statistics = get_default_statistics()
        cfg = dict(locals())
        for k in ("self", "__class__", "param_statistics"):
            cfg.pop(k, None)
        self.model_config = cfg
        self.gnn = GrappaGNN(out_feats=graph_node_features, in_feats=in_feats, node_feats=gnn_width, n_conv=gnn_convolutions,
                             n_att=gnn_attentional_layers, n_heads=gnn_attention_heads, in_feat_name=in_feat_name,
                             in_feat_dims=in_feat_dims, conv_dropout=gnn_dropout_conv, attention_dropout=gnn_dropout_attention,
                             final_dropout=gnn_dropout_final, initial_dropout=gnn_dropout_initial, layer_norm=layer_norm,
                             self_interaction=self_interaction)
        self.parameter_writer = WriteParameters(cfg, param_statistics)
        self.field_of_view = gnn_attentional_layers + gnn_convolutions + 3
        self.on_heads_backward_done = None      # optional callback (dist.BucketedGradReducer): overlap of the gradient all-reduce

    def forward(self, g):
        # tuple-index consistency (reference grappa.py:122-128) is validated once per batch, on the host, when the plan is built
        plan = g.plan()
        be = get_backend()
        if hasattr(be, "set_tail_launches"):
            # ONE plan setting for every product of the call, the GNN's included, chosen before the first of them (flipping it
            # in front of the heads planned the first call's GNN differently from every later call's).  Several streams busy: a product's
            # partial last round overlaps with another head's kernels, its split-K tail launch is a loss
            multi = self.parameter_writer.head_streams > 1 and plan.device.type == "cuda"
            # (opt-in GRAPPA_GNN_TAILS=1: the GNN runs alone on the chip -- forward and backward -- so ITS products keep their tail launches;
            # the setting is per PHASE and the same in every call: the backward pass switches at ops.SplitHeadsFn)
            be.set_tail_launches(True if (multi and getattr(be, "gnn_tails", False)) else not multi)
        g = self.gnn(g)
        if hasattr(be, "set_tail_launches") and getattr(be, "gnn_tails", False):
            be.set_tail_launches(not (self.parameter_writer.head_streams > 1 and plan.device.type == "cuda"))
        h = g.nodes["n1"].data["h"]
        if self.on_heads_backward_done is not None and h.requires_grad:
            # fires when the gradient of the atom embedding is complete = every writer head has finished its backward pass
            cb = self.on_heads_backward_done
            h.register_hook(lambda grad: (cb(), None)[1])
        g = self.parameter_writer(g)
        return g
