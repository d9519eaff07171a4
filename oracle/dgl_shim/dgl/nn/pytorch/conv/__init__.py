"""ORACLE / TEST INFRASTRUCTURE ONLY -- restatement of the two DGL conv layers the reference
instantiates (models/graph_attention.py:249 DotGatConv, :360-363 SAGEConv(mean)) from DGL's
published source/docs (DGL 1.1 - 2.1; not pinned by the reference).  GATConv/GATv2Conv exist
only so that the `assert attention_layer in [...]` at graph_attention.py:246 evaluates.
"""
import torch
from torch import nn


def _edge_softmax(dst, scores, num_dst):
    """softmax over the incoming edges of every destination node. scores: (E, H, 1)."""
    mx = torch.full((num_dst,) + tuple(scores.shape[1:]), float("-inf"), dtype=scores.dtype, device=scores.device)
    mx = mx.index_reduce(0, dst, scores, "amax", include_self=True)
    ex = torch.exp(scores - mx[dst])
    den = torch.zeros_like(mx).index_add(0, dst, ex)
    return ex / den[dst]


class DotGatConv(nn.Module):
    """ft = fc(h).view(N,H,D) used as source AND destination feature (one bias-free Linear);
    a_uv = <ft_u, ft_v>; alpha = edge_softmax(a / sqrt(D)) over the in-edges of v;
    rst_v = sum_u alpha_uv * ft_u.  Raises for zero-in-degree nodes like DGL does."""

    def __init__(self, in_feats, out_feats, num_heads, allow_zero_in_degree=False):
        super().__init__()
        self._in_feats = in_feats
        self._out_feats = out_feats
        self._num_heads = num_heads
        self._allow_zero_in_degree = allow_zero_in_degree
        self.fc = nn.Linear(in_feats, out_feats * num_heads, bias=False)

    def forward(self, graph, feat, get_attention=False):
        src, dst = graph.edges()
        N = graph.num_nodes()
        if not self._allow_zero_in_degree:
            if (torch.bincount(dst, minlength=N) == 0).any():
                raise RuntimeError("There are 0-in-degree nodes in the graph")
        ft = self.fc(feat).view(-1, self._num_heads, self._out_feats)
        a = (ft[src] * ft[dst]).sum(dim=-1, keepdim=True)  # u_dot_v -> (E,H,1)
        sa = _edge_softmax(dst, a / self._out_feats ** 0.5, N)
        rst = torch.zeros_like(ft).index_add(0, dst, ft[src] * sa)
        if get_attention:
            return rst, sa
        return rst


class SAGEConv(nn.Module):
    """aggregator 'mean' only: rst = fc_self(h_v) + fc_neigh(mean_{u->v} h_u) + bias
    (DGL >= 0.8 layout: bias-free fc_self/fc_neigh + separate bias parameter)."""

    def __init__(self, in_feats, out_feats, aggregator_type, feat_drop=0.0, bias=True, norm=None, activation=None):
        super().__init__()
        assert aggregator_type == "mean", "shim: only the 'mean' aggregator is used by the reference"
        self._in_src_feats = self._in_dst_feats = in_feats
        self._out_feats = out_feats
        self._aggre_type = aggregator_type
        self.fc_neigh = nn.Linear(in_feats, out_feats, bias=False)
        self.fc_self = nn.Linear(in_feats, out_feats, bias=False)
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_feats))
        else:
            self.register_buffer("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        gain = nn.init.calculate_gain("relu")
        nn.init.xavier_uniform_(self.fc_self.weight, gain=gain)
        nn.init.xavier_uniform_(self.fc_neigh.weight, gain=gain)

    def forward(self, graph, feat, edge_weight=None):
        src, dst = graph.edges()
        N = graph.num_nodes()
        deg = torch.bincount(dst, minlength=N).clamp(min=1).to(feat.dtype)
        lin_before_mp = self._in_src_feats > self._out_feats
        msg = self.fc_neigh(feat) if lin_before_mp else feat
        h_neigh = torch.zeros((N, msg.shape[1]), dtype=feat.dtype, device=feat.device).index_add(0, dst, msg[src])
        h_neigh = h_neigh / deg.unsqueeze(-1)
        if not lin_before_mp:
            h_neigh = self.fc_neigh(h_neigh)
        rst = self.fc_self(feat) + h_neigh
        if self.bias is not None:
            rst = rst + self.bias
        return rst


class GATConv(nn.Module):
    def __init__(self, *a, **k):
        raise NotImplementedError("dgl shim: GATConv is not used by the reference's configs")


class GATv2Conv(nn.Module):
    def __init__(self, *a, **k):
        raise NotImplementedError("dgl shim: GATv2Conv is not used by the reference's configs")
