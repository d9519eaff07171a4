"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product package `grappa_amd`.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

CPU restatement (plain PyTorch fp32 ops + autograd, no DGL) of the Grappa hot path:
GrappaModel.forward -> Energy -> MolwiseLoss.  It is the parity target of the HIP engine and
the "port" timed as `cpu_baseline` on the GPU box (the reference itself needs DGL, which is not
installable there, and reference code must not travel).

PARITY PIN: this restatement is checked in tests/test_oracle_goldens.py against fixtures under
tests/golden/ that were produced by running the *reference's own modules* (imported from
/root/reference/src with the pure-torch DGL shim in oracle/dgl_shim) in the build container --
see oracle/make_goldens.py.  DGL's arithmetic (DotGatConv / SAGEConv / readout_nodes / batch) is
third-party and absent from /root/reference (unpinned `pip install dgl`, DGL 1.1-2.1 era;
installation_openmm.sh:35-63); it is restated from DGL's published semantics, so parity of the
graph-attention step is pinned to that restatement, not to a DGL binary ("parity unpinned" at
the DGL boundary).

Reference lines followed (all under /root/reference/src/grappa/):
  models/graph_attention.py:142-183 (GrappaGNN.forward), :276-310 (ResidualAttentionBlock),
  :383-415 (ResidualConvBlock), :418-444 (PositionalEncoding);
  models/interaction_parameters.py:155-180 (RepProjector), :244-266 (bond), :337-362 (angle),
  :519-562 (torsion); models/perm_equiv_transformer.py:127-151, :239-276;
  models/network_utils.py:44-54, :112-133, :144-145; models/final_layer.py:52, :91-97;
  models/internal_coordinates.py:41-122, :150-210; models/energy.py:8-71, :99-145;
  training/loss.py:45-167; utils/graph_utils.py:35-86.
Deliberate deviations: no dihedral noise (Q1), cross product over the last
axis (Q2), the improper regulariser is skipped for molecules without impropers (Q4).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

TUPLE_LEVELS = ["n2", "n3", "n4", "n4_improper"]
MAX_ELEMENT = 53
CHARGE_MODELS = ["am1BCC", "amber99"]
DEFAULT_DIMS = {"atomic_number": MAX_ELEMENT, "ring_encoding": 7, "partial_charge": 1, "sp_hybridization": 6,
                "mass": 2, "degree": 6, "is_radical": 1, "laplacian_positional_encoding": 5,
                "charge_model": len(CHARGE_MODELS)}


def default_statistics():
    return {
        "mean": {"n2_k": torch.tensor([763.2819]), "n2_eq": torch.tensor([1.2353]), "n3_k": torch.tensor([105.6576]),
                 "n3_eq": torch.tensor([1.9750]),
                 "n4_k": torch.tensor([1.5617e-01, -5.8312e-01, 7.0820e-02, -6.3840e-04, 4.7139e-04, -4.1655e-04]),
                 "n4_improper_k": torch.tensor([0.0000, -2.3933, 0.0000])},
        "std": {"n2_k": torch.tensor([161.2278]), "n2_eq": torch.tensor([0.1953]), "n3_k": torch.tensor([26.5965]),
                "n3_eq": torch.tensor([0.0917]), "n4_k": torch.tensor([0.4977, 1.2465, 0.1466, 0.0192, 0.0075, 0.0066]),
                "n4_improper_k": torch.tensor([0.0000, 4.0571, 0.0000])}}


def n1_edges(g):
    """(src, dst) of the atom graph for a MolBatch or a dgl-shim graph."""
    try:
        return g.edges()
    except Exception:
        return g.edges(etype="n1_edge")


def mol_counts(g, ntype) -> torch.Tensor:
    return g.batch_num_nodes(ntype).long().cpu()


# ----------------------------------------------------------------------------- GNN
def charge_encoding(q: torch.Tensor, dim: int = 16, lo: float = -2.0, hi: float = 2.0) -> torch.Tensor:
    v = torch.clamp(q, lo, hi)
    s = (v + hi) / (hi - lo)
    freq = torch.exp(torch.arange(0, dim // 2, dtype=torch.float32, device=q.device)
                     * -torch.log(torch.tensor(10000.0, device=q.device)) / (dim // 2))
    enc = torch.zeros(len(q), dim, device=q.device)
    enc[:, 0::2] = torch.sin(s.unsqueeze(1) * freq)
    enc[:, 1::2] = torch.cos(s.unsqueeze(1) * freq)
    return enc


def dot_gat(ft: torch.Tensor, src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """ft (N,H,D) -> (N,H,D): softmax over the in-edges of every destination of <ft_u,ft_v>/sqrt(D)."""
    N, H, D = ft.shape
    a = (ft[src] * ft[dst]).sum(-1) / D ** 0.5                                   # (E,H)
    mx = torch.full((N, H), float("-inf"), dtype=ft.dtype, device=ft.device).index_reduce(0, dst, a, "amax")
    ex = torch.exp(a - mx[dst])
    den = torch.zeros((N, H), dtype=ft.dtype, device=ft.device).index_add(0, dst, ex)
    alpha = ex / den[dst]
    return torch.zeros_like(ft).index_add(0, dst, ft[src] * alpha.unsqueeze(-1))


class _GraphFC(nn.Module):
    """parameter holder named like DGL's DotGatConv (`fc.weight`, no bias)."""

    def __init__(self, in_feats, out_feats, num_heads):
        super().__init__()
        self.fc = nn.Linear(in_feats, out_feats * num_heads, bias=False)
        self.num_heads, self.out_feats = num_heads, out_feats


class _SageParams(nn.Module):
    def __init__(self, in_feats, out_feats):
        super().__init__()
        self.fc_neigh = nn.Linear(in_feats, out_feats, bias=False)
        self.fc_self = nn.Linear(in_feats, out_feats, bias=False)
        self.bias = nn.Parameter(torch.zeros(out_feats))
        gain = nn.init.calculate_gain("relu")
        nn.init.xavier_uniform_(self.fc_self.weight, gain=gain)
        nn.init.xavier_uniform_(self.fc_neigh.weight, gain=gain)


class AttBlock(nn.Module):
    def __init__(self, feats, heads, dropout):
        super().__init__()
        self.graph_module = _GraphFC(feats, feats // heads, heads)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.layer_norm = nn.LayerNorm(feats)
        self.head_reducer = nn.Linear(feats, feats)
        self.interaction_norm = nn.LayerNorm(feats)
        self.self_interaction = nn.Sequential(nn.Linear(feats, 4 * feats), nn.ELU(), nn.Linear(4 * feats, feats), nn.ELU())

    def forward(self, src, dst, h):
        h = self.layer_norm(h)
        skip = h
        gm = self.graph_module
        ft = gm.fc(h).view(-1, gm.num_heads, gm.out_feats)
        h = dot_gat(ft, src, dst).flatten(-2, -1)
        h = self.dropout1(self.head_reducer(h)) + skip
        h = self.interaction_norm(h)
        skip = h
        return self.dropout2(self.self_interaction(h)) + skip


class ConvBlock(nn.Module):
    def __init__(self, feats, dropout):
        super().__init__()
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.graph_module = _SageParams(feats, feats)
        self.layer_norm = nn.LayerNorm(feats)
        self.self_interaction = nn.Sequential(nn.Linear(feats, feats), nn.ELU())
        self.interaction_norm = nn.LayerNorm(feats)

    def forward(self, src, dst, h):
        h = self.layer_norm(h)
        skip = h
        N = h.shape[0]
        deg = torch.bincount(dst, minlength=N).clamp(min=1).to(h.dtype)
        mean = torch.zeros_like(h).index_add(0, dst, h[src]) / deg.unsqueeze(-1)
        gm = self.graph_module
        h = F.elu(gm.fc_self(h) + gm.fc_neigh(mean) + gm.bias)
        h = self.dropout1(h) + skip
        h = self.interaction_norm(h)
        skip = h
        return self.dropout2(self.self_interaction(h)) + skip


class RefGNN(nn.Module):
    def __init__(self, out_feats, in_feat_name, in_feat_dims, node_feats, n_conv, n_att, n_heads,
                 conv_dropout, attention_dropout, final_dropout, initial_dropout):
        super().__init__()
        dims = dict(DEFAULT_DIMS)
        dims.update(in_feat_dims)
        self.in_feat_name = list(in_feat_name)
        self.in_feats = sum(dims[f] for f in self.in_feat_name) + 16
        self.initial_dropout = nn.Dropout(initial_dropout)
        self.final_dropout = nn.Dropout(final_dropout)
        self.pre_dense = nn.Sequential(nn.Linear(self.in_feats, node_feats), nn.ELU())
        self.conv_blocks = nn.ModuleList([ConvBlock(node_feats, conv_dropout) for _ in range(n_conv)])
        self.att_blocks = nn.ModuleList([AttBlock(node_feats, n_heads, attention_dropout) for _ in range(n_att)])
        self.post_dense = nn.Sequential(nn.Linear(node_feats, out_feats))
        self.blocks = self.conv_blocks + self.att_blocks

    def forward(self, g):
        d = g.nodes["n1"].data
        dt = self.pre_dense[0].weight.dtype          # fp32; float64 when the model was .double()d (tests: the oracle as ground truth)
        x = torch.cat([d[f].to(dt) if d[f].dim() >= 2 else d[f].unsqueeze(-1).to(dt) for f in self.in_feat_name], dim=-1)
        x = torch.cat([x, charge_encoding(d["partial_charge"]).to(dt)], dim=-1)
        h = self.initial_dropout(self.pre_dense(x))
        src, dst = n1_edges(g)
        src, dst = src.long(), dst.long()
        for blk in self.blocks:
            h = blk(src, dst, h)
        h = self.final_dropout(self.post_dense(h))
        d["h"] = h
        return g


# ----------------------------------------------------------------------------- writers
class FeedForward(nn.Module):
    def __init__(self, in_feats, hidden, out_feats, dropout, skip):
        super().__init__()
        self.linear1 = nn.Linear(in_feats, hidden)
        self.linear2 = nn.Linear(hidden, out_feats)
        self.dropout = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(in_feats)
        self.skip = skip

    def forward(self, x):
        x = self.norm1(x)
        y = self.dropout(self.linear2(F.elu(self.linear1(x))))
        return y + x if self.skip else y


class TransformerLayer(nn.Module):
    def __init__(self, feats, heads, hidden, dropout):
        super().__init__()
        self.norm1 = nn.LayerNorm(feats)
        self.attn = nn.MultiheadAttention(feats, heads, dropout=0)
        self.dropout = nn.Dropout(dropout)
        self.ff = FeedForward(feats, hidden, feats, dropout, skip=True)

    def forward(self, x):
        x = self.norm1(x)
        a, _ = self.attn(x, x, x, need_weights=False)
        return self.ff(self.dropout(a) + x)


class _Transformer(nn.Module):
    def __init__(self, n_feats, heads, hidden, n_layers, dropout, pos_enc: Optional[torch.Tensor]):
        super().__init__()
        if pos_enc is not None:
            self.register_buffer("positional_encoding", pos_enc.float())
            n_feats = n_feats + pos_enc.shape[1]
        else:
            self.positional_encoding = None
        self.n_feats = n_feats
        self.transformer = nn.Sequential(*[TransformerLayer(n_feats, heads, hidden, dropout) for _ in range(n_layers)])

    def forward(self, x):
        if self.positional_encoding is not None:
            x = torch.cat([x, self.positional_encoding.unsqueeze(1).repeat(1, x.shape[1], 1)], dim=-1)
        return self.transformer(x)


class _Symmetriser(nn.Module):
    def __init__(self, in_feats, out_feats, perms: torch.Tensor, hidden, n_layers):
        super().__init__()
        self.register_buffer("permutation_prefactors", torch.ones(perms.shape[0]).view(-1, 1, 1))
        self.register_buffer("permutations", perms.int())
        self.n_seq = perms.shape[1]
        self.out_feats = out_feats
        layers = [FeedForward(in_feats * self.n_seq, hidden, hidden if n_layers > 1 else out_feats, 0.0, skip=False)]
        for i in range(1, n_layers):
            last = i == n_layers - 1
            layers.append(FeedForward(hidden, hidden, out_feats if last else hidden, 0.0, skip=not last))
        self.mlp = nn.Sequential(*layers)

    def forward(self, x):                                   # (s, T, F)
        P, T = self.permutations.shape[0], x.shape[1]
        xp = torch.stack([x[p.long()] for p in self.permutations], dim=0).transpose(1, 2).contiguous().view(P * T, -1)
        y = self.mlp(xp).view(P, T, self.out_feats) * self.permutation_prefactors
        return y.sum(dim=0)


class _SymTransformer(nn.Module):
    def __init__(self, n_feats, heads, hidden, n_layers, out_feats, perms, dropout, sym_layers, sym_hidden, pos_enc):
        super().__init__()
        if n_layers > 0:
            self.grappa_transformer = _Transformer(n_feats, heads, hidden, n_layers, dropout, pos_enc)
            n_feats = self.grappa_transformer.n_feats
        else:
            self.grappa_transformer = None
        self.symmetriser = _Symmetriser(n_feats, out_feats, perms, sym_hidden, sym_layers)

    def forward(self, x):
        if self.grappa_transformer is not None:
            x = self.grappa_transformer(x)
        return self.symmetriser(x)


class _RepProjector(nn.Module):
    def __init__(self, in_feats, out_feats):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(in_feats, out_feats), nn.ELU())

    def forward(self, h, idxs):
        a = self.mlp(h)
        if len(idxs) == 0:
            return torch.zeros((idxs.shape[1], 0, a.shape[-1]), dtype=a.dtype, device=a.device)
        return a[idxs.long()].transpose(0, 1).contiguous()


class _ToPositive(nn.Module):
    def __init__(self, mean, std, min_=0.0):
        super().__init__()
        self.register_buffer("mean_over_std", torch.tensor(float(mean / std)))
        self.register_buffer("std", torch.tensor(float(std)))
        self.register_buffer("min_", torch.tensor(float(min_)))

    def forward(self, x):
        return self.std * (F.elu(self.mean_over_std + x - 1) + 1) + self.min_


class _ToRange(nn.Module):
    def __init__(self, max_, std):
        super().__init__()
        self.register_buffer("std_over_max", torch.tensor(float(std / max_)).float())
        self.register_buffer("max", torch.tensor(float(max_)).float())

    def forward(self, x):
        return self.max * torch.sigmoid(self.std_over_max * x)


def _pos_enc(s, positional_encoding, wrong_symmetry=False):
    if not positional_encoding or s == 2:
        return None
    if s == 3:
        return torch.tensor([[0.], [1.], [0.]])
    if wrong_symmetry:
        return torch.tensor([[0.], [0.], [1.], [0.]])
    return torch.tensor([[0.], [1.], [1.], [0.]])


class BondWriter(nn.Module):
    def __init__(self, rep, width, stats, n_att, heads, sym_depth, dropout, sym_width, gate):
        super().__init__()
        eps = 1e-6
        self.rep_projector = _RepProjector(rep, width)
        self.gate = gate
        self.bond_model = _SymTransformer(width, heads, width, n_att, 2 + int(gate), torch.tensor([[0, 1], [1, 0]]),
                                          dropout, sym_depth, sym_width, None)
        self.to_k = _ToPositive(stats["mean"]["n2_k"].item(), stats["std"]["n2_k"].item() + eps)
        self.to_eq = _ToPositive(stats["mean"]["n2_eq"].item(), stats["std"]["n2_eq"].item() + eps)

    def forward(self, g):
        c = self.bond_model(self.rep_projector(g.nodes["n1"].data["h"], g.nodes["n2"].data["idxs"]))
        g.nodes["n2"].data["eq"] = self.to_eq(c[:, 0])
        g.nodes["n2"].data["k"] = self.to_k(c[:, 1])        # the harmonic gate is a no-op in the reference (Q3)
        return g


class AngleWriter(nn.Module):
    def __init__(self, rep, width, stats, n_att, heads, sym_depth, dropout, sym_width, pos, gate):
        super().__init__()
        eps = 1e-6
        proj = width - 1 if pos else width
        self.rep_projector = _RepProjector(rep, proj)
        self.angle_model = _SymTransformer(proj, heads, width, n_att, 2 + int(gate), torch.tensor([[0, 1, 2], [2, 1, 0]]),
                                           dropout, sym_depth, sym_width, _pos_enc(3, pos))
        self.to_k = _ToPositive(stats["mean"]["n3_k"].item(), stats["std"]["n3_k"].item() + eps)
        self.to_eq = _ToRange(math.pi, stats["std"]["n3_eq"].item() + eps)

    def forward(self, g):
        x = self.rep_projector(g.nodes["n1"].data["h"], g.nodes["n3"].data["idxs"])
        if x.shape[1] == 0:
            # no angle in the whole batch (diatomics only).  The reference guards only the torsion writers
            # (interaction_parameters.py:531-532) and has no such molecules; the same empty result is the natural extension.
            g.nodes["n3"].data["eq"] = torch.zeros(0, dtype=x.dtype, device=x.device)
            g.nodes["n3"].data["k"] = torch.zeros(0, dtype=x.dtype, device=x.device)
            return g
        c = self.angle_model(x)
        g.nodes["n3"].data["eq"] = self.to_eq(c[:, 0])
        g.nodes["n3"].data["k"] = self.to_k(c[:, 1])
        return g


class TorsionWriter(nn.Module):
    def __init__(self, rep, width, n_periodicity, improper, n_att, heads, sym_depth, dropout, sym_width, stats, pos,
                 gated, wrong_symmetry, cutoff):
        super().__init__()
        eps = 1e-1 if gated else 1e-2
        self.gated, self.improper, self.cutoff = gated, improper, cutoff
        self.register_buffer("n_periodicity", torch.tensor(n_periodicity).long())
        if not improper:
            km, ks = stats["mean"]["n4_k"], stats["std"]["n4_k"] + eps
        elif "n4_improper_k" not in stats["mean"]:
            km, ks = torch.zeros(n_periodicity), torch.ones(n_periodicity)
        else:
            km, ks = stats["mean"]["n4_improper_k"], stats["std"]["n4_improper_k"] + eps
        self.register_buffer("k_mean", km[:n_periodicity].unsqueeze(0).clone())
        self.register_buffer("k_std", ks[:n_periodicity].unsqueeze(0).clone())
        proj = width - 1 if pos else width
        self.rep_projector = _RepProjector(rep, proj)
        if not improper:
            perms = torch.tensor([[0, 1, 2, 3], [3, 2, 1, 0]])
        elif wrong_symmetry:
            perms = torch.tensor([[0, 1, 2, 3], [3, 1, 2, 0], [1, 3, 2, 0], [0, 3, 2, 1], [3, 0, 2, 1], [1, 0, 2, 3]])
        else:
            perms = torch.tensor([[0, 1, 2, 3], [3, 1, 2, 0]])
        n_out = 2 * n_periodicity if gated else n_periodicity
        self.torsion_model = _SymTransformer(proj, heads, width, n_att, n_out, perms, dropout, sym_depth, sym_width,
                                             _pos_enc(4, pos, improper and wrong_symmetry))

    def forward(self, g):
        lvl = "n4_improper" if self.improper else "n4"
        n = int(self.n_periodicity)
        x = self.rep_projector(g.nodes["n1"].data["h"], g.nodes[lvl].data["idxs"])
        if x.shape[1] == 0:
            g.nodes[lvl].data["k"] = torch.zeros((0, n), dtype=x.dtype, device=x.device)
            return g
        c = self.torsion_model(x)
        if self.gated:
            c = c[:, :n] * torch.sigmoid(c[:, n:]) * self.k_std
        else:
            c = c * self.k_std + self.k_mean
        if self.cutoff > 0:
            c = torch.where(torch.abs(c) > self.cutoff, c, torch.zeros_like(c))
        g.nodes[lvl].data["k"] = c
        return g


class _Writers(nn.Module):
    def __init__(self, cfg, stats):
        super().__init__()
        rep, drop, pos = cfg["graph_node_features"], cfg["parameter_dropout"], cfg["positional_encoding"]
        gate = cfg.get("harmonic_gate", False)
        self.bond_writer = BondWriter(rep, cfg["bond_transformer_width"], stats, cfg["bond_transformer_depth"], cfg["bond_n_heads"],
                                      cfg["bond_symmetriser_depth"], drop, cfg["bond_symmetriser_width"], gate)
        self.angle_writer = AngleWriter(rep, cfg["angle_transformer_width"], stats, cfg["angle_transformer_depth"], cfg["angle_n_heads"],
                                        cfg["angle_symmetriser_depth"], drop, cfg["angle_symmetriser_width"], pos, gate)
        self.proper_writer = TorsionWriter(rep, cfg["proper_transformer_width"], cfg["n_periodicity_proper"], False,
                                           cfg["proper_transformer_depth"], cfg["proper_n_heads"], cfg["proper_symmetriser_depth"], drop,
                                           cfg["proper_symmetriser_width"], stats, pos, cfg["gated_torsion"], False, cfg["torsion_cutoff"])
        self.improper_writer = TorsionWriter(rep, cfg["improper_transformer_width"], cfg["n_periodicity_improper"], True,
                                             cfg["improper_transformer_depth"], cfg["improper_n_heads"], cfg["improper_symmetriser_depth"], drop,
                                             cfg["improper_symmetriser_width"], stats, pos, cfg["gated_torsion"], cfg["wrong_symmetry"],
                                             cfg["torsion_cutoff"])

    def forward(self, g):
        return self.improper_writer(self.proper_writer(self.angle_writer(self.bond_writer(g))))


_CFG_DEFAULTS = dict(
    graph_node_features=512, in_feats=None, in_feat_name=["atomic_number", "ring_encoding", "partial_charge"], in_feat_dims={},
    gnn_width=None, gnn_attentional_layers=3, gnn_convolutions=3, gnn_attention_heads=8, gnn_dropout_attention=0.,
    gnn_dropout_initial=0., gnn_dropout_conv=0., gnn_dropout_final=0., parameter_dropout=0.,
    bond_transformer_depth=2, bond_n_heads=8, bond_transformer_width=512, bond_symmetriser_depth=2, bond_symmetriser_width=256,
    angle_transformer_depth=2, angle_n_heads=8, angle_transformer_width=512, angle_symmetriser_depth=2, angle_symmetriser_width=256,
    proper_transformer_depth=2, proper_n_heads=8, proper_transformer_width=512, proper_symmetriser_depth=2, proper_symmetriser_width=256,
    improper_transformer_depth=2, improper_n_heads=8, improper_transformer_width=512, improper_symmetriser_depth=2,
    improper_symmetriser_width=256, n_periodicity_proper=6, n_periodicity_improper=3, gated_torsion=False, wrong_symmetry=False,
    positional_encoding=True, layer_norm=True, self_interaction=True, learnable_statistics=False, torsion_cutoff=1e-4, harmonic_gate=False)


class RefGrappaModel(nn.Module):
    """State-dict compatible with the reference's GrappaModel (grappa.py:51) for layer_norm=True,
    self_interaction=True, learnable_statistics=False (the only values the shipped configs use)."""

    def __init__(self, param_statistics=None, **model_config):
        super().__init__()
        cfg = dict(_CFG_DEFAULTS)
        cfg.update(model_config)
        assert cfg["layer_norm"] and cfg["self_interaction"] and not cfg["learnable_statistics"]
        stats = param_statistics if param_statistics is not None else default_statistics()
        width = cfg["gnn_width"] if cfg["gnn_width"] is not None else cfg["graph_node_features"]
        names = cfg["in_feat_name"] if isinstance(cfg["in_feat_name"], list) else [cfg["in_feat_name"]]
        self.gnn = RefGNN(cfg["graph_node_features"], names, cfg["in_feat_dims"], width, cfg["gnn_convolutions"],
                          cfg["gnn_attentional_layers"], cfg["gnn_attention_heads"], cfg["gnn_dropout_conv"],
                          cfg["gnn_dropout_attention"], cfg["gnn_dropout_final"], cfg["gnn_dropout_initial"])
        self.parameter_writer = _Writers(cfg, stats)
        self.field_of_view = cfg["gnn_attentional_layers"] + cfg["gnn_convolutions"] + 3

    def forward(self, g):
        return self.parameter_writer(self.gnn(g))


# ----------------------------------------------------------------------------- energy
def bond_length(x0, x1):
    return torch.norm(x0 - x1, p=2, dim=-1)


def bond_angle(x0, x1, x2):
    r0, r1 = x1 - x0, x1 - x2
    return torch.atan2(torch.norm(torch.cross(r0, r1, dim=-1), p=2, dim=-1), (r0 * r1).sum(-1))


def dihedral(x0, x1, x2, x3):
    r01, r21, r23 = x1 - x0, x1 - x2, x3 - x2
    n1 = torch.cross(r01, r21, dim=-1)
    n2 = torch.cross(r21, r23, dim=-1)
    rn = r21 / torch.norm(r21, dim=-1, keepdim=True)
    y = (torch.cross(n1, n2, dim=-1) * rn).sum(-1)
    x = (n1 * n2).sum(-1)
    return torch.atan2(y, x)


def _segment_sum(x, counts):
    B = len(counts)
    seg = torch.repeat_interleave(torch.arange(B, device=x.device), counts.to(x.device))
    return torch.zeros((B,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device).index_add(0, seg, x)


class RefEnergy(nn.Module):
    """E (B,C) and dE/dxyz (N,C,3) with create_graph=True, as the reference's Energy module."""

    def __init__(self, terms=("n2", "n3", "n4", "n4_improper"), suffix="", offset_torsion=False, write_suffix=None, gradients=True):
        super().__init__()
        self.terms, self.suffix, self.offset_torsion, self.gradients = list(terms), suffix, offset_torsion, gradients
        self.write_suffix = suffix if write_suffix is None else write_suffix

    def forward(self, g):
        with torch.enable_grad():
            xyz = g.nodes["n1"].data["xyz"]
            if self.gradients:
                xyz = xyz.detach().requires_grad_(True)
            C = xyz.shape[1]
            B = len(mol_counts(g, "g"))
            energy = torch.zeros((B, C), device=xyz.device)
            for term in self.terms:
                idx = g.nodes[term].data["idxs"].long()
                k = g.nodes[term].data["k" + self.suffix]
                if len(idx) == 0:
                    x = torch.zeros((0, C), device=xyz.device)
                else:
                    p = xyz[idx]
                    if term == "n2":
                        x = bond_length(p[:, 0], p[:, 1])
                    elif term == "n3":
                        x = bond_angle(p[:, 0], p[:, 1], p[:, 2])
                    else:
                        x = dihedral(p[:, 0], p[:, 1], p[:, 2], p[:, 3])
                g.nodes[term].data["x"] = x
                if term in ("n2", "n3"):
                    eq = g.nodes[term].data["eq" + self.suffix]
                    e = 0.5 * k.unsqueeze(-1) * torch.square(x - eq.unsqueeze(-1))
                else:
                    n = torch.arange(1, k.shape[1] + 1, device=k.device, dtype=torch.float32).view(1, -1, 1)
                    e = k.unsqueeze(-1) * torch.cos(n * x.unsqueeze(1))
                    if self.offset_torsion:
                        e = e + torch.abs(k).unsqueeze(-1)
                    e = e.sum(dim=1)
                contrib = _segment_sum(e, mol_counts(g, term))
                energy = energy + contrib
                g.nodes["g"].data["energy_" + term + self.write_suffix] = contrib.detach()
                g.nodes[term].data["energy" + self.write_suffix] = e
            g.nodes["g"].data["energy" + self.write_suffix] = energy
            if self.gradients:
                grad = torch.autograd.grad(energy.sum(), xyz, retain_graph=True, create_graph=True, allow_unused=True)[0]
                g.nodes["n1"].data["gradient" + self.write_suffix] = grad
        return g


# ----------------------------------------------------------------------------- loss
class RefMolwiseLoss(nn.Module):
    """Vectorised restatement of training/loss.py:45-167 (mean over molecules of per-molecule MSEs)."""

    def __init__(self, gradient_weight=0.8, energy_weight=1.0, param_weight=1e-3, tuplewise_weight=0,
                 weights={"n2_k": 1e-3, "n3_k": 1e-2, "n4_k": 1e-4}, skip_params_if_not_present=True,
                 proper_regularisation=0., improper_regularisation=0., param_weights_by_dataset={}):
        super().__init__()
        assert tuplewise_weight == 0
        self.gw, self.ew, self.pw = gradient_weight, energy_weight, param_weight
        self.weights = dict(weights)
        self.skip = skip_params_if_not_present
        self.reg_p, self.reg_i = proper_regularisation, improper_regularisation
        self.by_ds = dict(param_weights_by_dataset)

    def forward(self, g, dsnames: List[str] = None):
        assert not (self.gw == 0 and self.ew == 0 and self.pw == 0)
        B = len(mol_counts(g, "g"))
        gd = g.nodes["g"].data
        dev = g.nodes["n1"].data["partial_charge"].device
        loss = torch.zeros(B, device=dev)
        if "is_dummy" in gd:
            real = (gd["is_dummy"] == 0).float()
        else:
            real = None
        if self.ew != 0.:
            e, er = gd["energy"], gd["energy_ref"]
            if torch.isnan(e).any() or torch.isnan(er).any():
                raise RuntimeError("energies are nan")
            m = torch.ones_like(e) if real is None else real
            nc = m.sum(1, keepdim=True)
            ec = (e - (e * m).sum(1, keepdim=True) / nc) - (er - (er * m).sum(1, keepdim=True) / nc)
            loss = loss + self.ew * (ec * ec * m).sum(1) / nc[:, 0]
        if self.gw != 0.:
            gr, grr = g.nodes["n1"].data["gradient"], g.nodes["n1"].data["gradient_ref"]
            if torch.isnan(gr).any() or torch.isnan(grr).any():
                raise RuntimeError("gradients are nan")
            cnt = mol_counts(g, "n1").to(dev)
            d2 = torch.square(gr - grr).sum(-1)                              # (N,C)
            per_mol = _segment_sum(d2, cnt)                                  # (B,C)
            m = torch.ones_like(per_mol) if real is None else real
            loss = loss + self.gw * (per_mol * m).sum(1) / (cnt.float() * m.sum(1) * 3.0)
        pw = torch.full((B,), float(self.pw), device=dev)
        if dsnames is not None:
            for i, n in enumerate(dsnames):
                if n in self.by_ds:
                    pw[i] = self.by_ds[n]
        if bool((pw != 0).any()):
            num = torch.zeros(B, device=dev)
            den = torch.zeros(B, device=dev)
            # get_parameters(graph, suffix="_ref") needs all six reference tensors (incl. the improper k_ref),
            # otherwise the whole parameter term is skipped (loss.py:80-86, graph_utils.py:8-32)
            all_levels = [("n2", "k"), ("n2", "eq"), ("n3", "k"), ("n3", "eq"), ("n4", "k"), ("n4_improper", "k")]
            found = all((name + "_ref") in g.nodes[lvl].data for lvl, name in all_levels)
            if not found and not self.skip:
                raise KeyError("reference parameters missing")
            for lvl, name in (all_levels[:5] if found else []):
                key = f"{lvl}_{name}"
                d = g.nodes[lvl].data
                fac = self.weights.get(key, 1.0)
                p, pr = d[name], d[name + "_ref"]
                if key == "n4_k":
                    if pr.shape[1] < p.shape[1]:
                        pr = torch.cat([pr, torch.zeros_like(pr[:, :(p.shape[1] - pr.shape[1])])], dim=1)
                    elif pr.shape[1] > p.shape[1]:
                        pr = pr[:, :p.shape[1]]
                nanmask = torch.isnan(pr)
                diff = torch.where(nanmask, torch.zeros_like(p), p - torch.where(nanmask, torch.zeros_like(pr), pr)) * fac
                sq = torch.square(diff).reshape(len(diff), -1)
                cnt = mol_counts(g, lvl).to(dev)
                num = num + _segment_sum(sq, cnt).sum(-1)
                den = den + cnt.float() * sq.shape[1]
            if found:
                loss = loss + pw * torch.where(den > 0, num / den.clamp(min=1), torch.full_like(num, float("nan")))
        if self.reg_p > 0.:
            k = g.nodes["n4"].data["k"]
            cnt = mol_counts(g, "n4").to(dev)
            s = _segment_sum(torch.square(k), cnt).sum(-1)
            loss = loss + torch.where(cnt > 0, self.reg_p * s / (cnt.float() * k.shape[1]).clamp(min=1), torch.zeros_like(s))
        if self.reg_i > 0.:
            k = g.nodes["n4_improper"].data["k"]
            cnt = mol_counts(g, "n4_improper").to(dev)
            s = _segment_sum(torch.square(k), cnt).sum(-1)
            # the reference adds this term twice (loss.py:128-132); molecules without impropers are skipped (Q4)
            loss = loss + torch.where(cnt > 0, 2.0 * self.reg_i * s / (cnt.float() * k.shape[1]).clamp(min=1), torch.zeros_like(s))
        self.last_per_molecule = loss.detach()
        return loss.sum() / B
