"""ORACLE / TEST INFRASTRUCTURE ONLY -- pure-torch shim of the DGL surface used by the
reference (see heterograph.py for provenance and the semantics restated)."""
from typing import Dict, List

import torch

from .heterograph import DGLGraph, DGLHeteroGraph  # noqa: F401  (submodule import must precede the function below)
from . import nn  # noqa: F401

__version__ = "0.0-shim"


def heterograph(data_dict, num_nodes_dict=None, idtype=None, device=None):  # noqa: F811 (shadows the submodule name, as in real DGL)
    """dgl.heterograph: {(src_t, e_t, dst_t): (src, dst)} -> graph (reference call site: data/Molecule.py:497)."""
    edges = {}
    nn_ = {} if num_nodes_dict is None else dict(num_nodes_dict)
    for (st, et, dt), (src, dst) in data_dict.items():
        src = torch.as_tensor(src).long()
        dst = torch.as_tensor(dst).long()
        edges[(st, et, dt)] = (src, dst)
        if num_nodes_dict is None:
            nn_[st] = max(nn_.get(st, 0), int(src.max()) + 1 if len(src) else 0)
            nn_[dt] = max(nn_.get(dt, 0), int(dst.max()) + 1 if len(dst) else 0)
    return DGLGraph(edges, nn_)


def node_type_subgraph(g, ntypes):
    return g.node_type_subgraph(ntypes)


def to_homogeneous(g, ndata=None, edata=None, store_type=True, return_count=False):
    """Only the single-node-type / single-edge-type case the reference uses (graph_attention.py:170)."""
    assert len(g.ntypes) == 1 and len(g.canonical_etypes) == 1
    nt = g.ntypes[0]
    k = g.canonical_etypes[0]
    h = DGLGraph({("_N", "_E", "_N"): g._edges[k]}, {"_N": g._num_nodes[nt]})
    if g._batch_num_nodes is not None:
        h._batch_num_nodes = {"_N": g._batch_num_nodes[nt]}
        h._batch_num_edges = {("_N", "_E", "_N"): g._batch_num_edges[k]}
    return h


def batch(graphs: List[DGLGraph], ndata=None, edata=None):
    """dgl.batch (reference call site: utils/dgl_utils.py:60)."""
    assert len(graphs) > 0
    g0 = graphs[0]
    ntypes = g0.ntypes
    offsets = {nt: [0] for nt in ntypes}
    for g in graphs:
        assert g.ntypes == ntypes
        for nt in ntypes:
            offsets[nt].append(offsets[nt][-1] + g.num_nodes(nt))
    edges = {}
    for k in g0.canonical_etypes:
        srcs, dsts = [], []
        for i, g in enumerate(graphs):
            s, d = g._edges[k]
            srcs.append(s + offsets[k[0]][i])
            dsts.append(d + offsets[k[2]][i])
        edges[k] = (torch.cat(srcs), torch.cat(dsts))
    out = DGLGraph(edges, {nt: offsets[nt][-1] for nt in ntypes})
    for nt in ntypes:
        feats = list(g0._ndata[nt].keys())
        for f in feats:
            out._ndata[nt][f] = torch.cat([g._ndata[nt][f] for g in graphs], dim=0)
    out._batch_num_nodes = {nt: torch.tensor([g.num_nodes(nt) for g in graphs], dtype=torch.long) for nt in ntypes}
    out._batch_num_edges = {k: torch.tensor([g.num_edges(k) for g in graphs], dtype=torch.long) for k in g0.canonical_etypes}
    return out


def unbatch(g: DGLGraph, node_split=None, edge_split=None):
    """dgl.unbatch (reference call site: utils/dgl_utils.py:69)."""
    B = g.batch_size
    bnn = {nt: g.batch_num_nodes(nt).tolist() for nt in g.ntypes}
    bne = {k: g.batch_num_edges(k).tolist() for k in g.canonical_etypes}
    noff = {nt: [0] for nt in g.ntypes}
    for nt in g.ntypes:
        for n in bnn[nt]:
            noff[nt].append(noff[nt][-1] + n)
    eoff = {k: [0] for k in g.canonical_etypes}
    for k in g.canonical_etypes:
        for n in bne[k]:
            eoff[k].append(eoff[k][-1] + n)
    out = []
    for i in range(B):
        edges = {}
        for k in g.canonical_etypes:
            s, d = g._edges[k]
            sl = slice(eoff[k][i], eoff[k][i + 1])
            edges[k] = (s[sl] - noff[k[0]][i], d[sl] - noff[k[2]][i])
        sub = DGLGraph(edges, {nt: bnn[nt][i] for nt in g.ntypes})
        for nt in g.ntypes:
            for f, t in g._ndata[nt].items():
                sub._ndata[nt][f] = t[noff[nt][i]:noff[nt][i + 1]]
        out.append(sub)
    return out


def readout_nodes(graph, feat, weight=None, *, op="sum", ntype=None):
    """dgl.readout_nodes: segment reduction over batch_num_nodes(ntype) -> (B, ...).
    Reference call sites: models/energy.py:69, utils/graph_utils.py:164."""
    assert weight is None
    x = graph.nodes[ntype].data[feat]
    counts = graph.batch_num_nodes(ntype).to(x.device)
    B = len(counts)
    seg = torch.repeat_interleave(torch.arange(B, device=x.device), counts)
    out = torch.zeros((B,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    if op == "sum":
        out = out.index_add(0, seg, x)
    elif op == "mean":
        out = out.index_add(0, seg, x) / counts.clamp(min=1).view(-1, *([1] * (x.dim() - 1))).to(x.dtype)
    else:
        raise NotImplementedError(op)
    return out


def save_graphs(*a, **k):
    raise NotImplementedError("dgl shim: save_graphs is outside the hot path")


def load_graphs(*a, **k):
    raise NotImplementedError("dgl shim: load_graphs is outside the hot path")
