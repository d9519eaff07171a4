"""ORACLE / TEST INFRASTRUCTURE ONLY -- never imported by grappa_amd.

Pure-torch stand-in for the *subset* of DGL's heterograph container that the
reference (hits-mbm-dev/grappa) touches.  DGL itself is a
third-party dependency of the reference that is neither vendored under
/root/reference nor pinned (installation_openmm.sh:35-63: whatever
`pip install dgl -f https://data.dgl.ai/wheels/...` resolved for torch 2.0.1/2.2.0,
i.e. DGL 1.1 - 2.1).  The semantics restated here follow DGL's public
documentation for those releases:

  * heterograph(data_dict): node count of a type = max node id + 1 over the relations
    it takes part in; node types / edge types sorted alphabetically.
  * batch(graphs): node features concatenated per type in input order, edges
    relabelled by cumulative node offsets, batch_num_nodes / batch_num_edges recorded.
  * unbatch(g): inverse of the above.
  * readout_nodes(g, feat, op='sum', ntype): segment reduction over batch_num_nodes(ntype).
  * to_homogeneous(node_type_subgraph(g, ['n1'])): a plain graph with the n1 edges.

With this module on sys.path the reference's grappa.models / grappa.data /
grappa.training.loss import and run unmodified in the build container.
"""
import copy as _copy
from typing import Dict, List, Tuple

import torch


class _NodeView:
    def __init__(self, graph, ntype):
        self._g = graph
        self._nt = ntype

    @property
    def data(self):
        return self._g._ndata[self._nt]


class _NodesAccessor:
    def __init__(self, graph):
        self._g = graph

    def __getitem__(self, ntype):
        if ntype not in self._g._ndata:
            raise KeyError(f"node type {ntype} not in graph")
        return _NodeView(self._g, ntype)


class _NDataView:
    """g.ndata[feat] -> {ntype: tensor} (heterograph form) or tensor (single node type)."""

    def __init__(self, graph):
        self._g = graph

    def __getitem__(self, feat):
        if len(self._g.ntypes) == 1:
            return self._g._ndata[self._g.ntypes[0]][feat]
        return {nt: d[feat] for nt, d in self._g._ndata.items() if feat in d}

    def __setitem__(self, feat, value):
        if len(self._g.ntypes) == 1:
            self._g._ndata[self._g.ntypes[0]][feat] = value
        else:
            for nt, v in value.items():
                self._g._ndata[nt][feat] = v

    def keys(self):
        ks = []
        for d in self._g._ndata.values():
            for k in d.keys():
                if k not in ks:
                    ks.append(k)
        return ks


class DGLGraph:
    """Dict-of-tensors heterograph.  Edges are kept per canonical edge type as (src, dst)."""

    def __init__(self, edges: Dict[Tuple[str, str, str], Tuple[torch.Tensor, torch.Tensor]],
                 num_nodes: Dict[str, int]):
        self._edges = {k: (v[0].long(), v[1].long()) for k, v in edges.items()}
        self._num_nodes = dict(num_nodes)
        self.ntypes = sorted(self._num_nodes.keys())
        self.canonical_etypes = sorted(self._edges.keys(), key=lambda k: k[1])
        self.etypes = [k[1] for k in self.canonical_etypes]
        self._ndata = {nt: {} for nt in self.ntypes}
        self._batch_num_nodes = None  # ntype -> LongTensor
        self._batch_num_edges = None  # canonical etype -> LongTensor

    # --- structure queries -------------------------------------------------------------
    @property
    def nodes(self):
        return _NodesAccessor(self)

    @property
    def ndata(self):
        return _NDataView(self)

    def num_nodes(self, ntype=None):
        if ntype is None:
            return sum(self._num_nodes.values())
        return self._num_nodes[ntype]

    number_of_nodes = num_nodes

    def _canon(self, etype):
        if etype is None:
            assert len(self.canonical_etypes) == 1
            return self.canonical_etypes[0]
        if isinstance(etype, tuple):
            return etype
        for k in self.canonical_etypes:
            if k[1] == etype:
                return k
        raise KeyError(etype)

    def num_edges(self, etype=None):
        if etype is None and len(self.canonical_etypes) != 1:
            return sum(len(v[0]) for v in self._edges.values())
        return len(self._edges[self._canon(etype)][0])

    number_of_edges = num_edges

    def edges(self, etype=None):
        return self._edges[self._canon(etype)]

    def in_degrees(self, etype=None):
        k = self._canon(etype)
        _, dst = self._edges[k]
        return torch.bincount(dst, minlength=self._num_nodes[k[2]])

    def batch_num_nodes(self, ntype=None):
        if ntype is None:
            assert len(self.ntypes) == 1
            ntype = self.ntypes[0]
        if self._batch_num_nodes is None:
            return torch.tensor([self._num_nodes[ntype]], dtype=torch.long, device=self.device)
        return self._batch_num_nodes[ntype]

    def batch_num_edges(self, etype=None):
        k = self._canon(etype)
        if self._batch_num_edges is None:
            return torch.tensor([len(self._edges[k][0])], dtype=torch.long, device=self.device)
        return self._batch_num_edges[k]

    @property
    def batch_size(self):
        if self._batch_num_nodes is None:
            return 1
        return len(next(iter(self._batch_num_nodes.values())))

    @property
    def device(self):
        for v in self._edges.values():
            return v[0].device
        return torch.device("cpu")

    def to(self, device):
        g = DGLGraph({k: (v[0].to(device), v[1].to(device)) for k, v in self._edges.items()},
                     self._num_nodes)
        for nt in self.ntypes:
            for f, t in self._ndata[nt].items():
                g._ndata[nt][f] = t.to(device)
        if self._batch_num_nodes is not None:
            g._batch_num_nodes = {k: v.to(device) for k, v in self._batch_num_nodes.items()}
            g._batch_num_edges = {k: v.to(device) for k, v in self._batch_num_edges.items()}
        return g

    def local_scope(self):
        import contextlib
        return contextlib.nullcontext()

    def node_type_subgraph(self, ntypes: List[str]):
        edges = {k: v for k, v in self._edges.items() if k[0] in ntypes and k[2] in ntypes}
        g = DGLGraph(edges, {nt: self._num_nodes[nt] for nt in ntypes})
        for nt in ntypes:
            g._ndata[nt] = dict(self._ndata[nt])
        if self._batch_num_nodes is not None:
            g._batch_num_nodes = {nt: self._batch_num_nodes[nt] for nt in ntypes}
            g._batch_num_edges = {k: self._batch_num_edges[k] for k in edges}
        return g

    # --- single-node-type helpers used by the reference's water guard (utils/dgl_utils.py:210-234) -------------------
    def to_networkx(self):
        """DGLGraph.to_networkx(): a networkx.MultiDiGraph over node ids 0..n-1 with one edge per stored edge."""
        import networkx as nx
        assert len(self.canonical_etypes) == 1
        src, dst = self._edges[self.canonical_etypes[0]]
        h = nx.MultiDiGraph()
        h.add_nodes_from(range(self.num_nodes()))
        h.add_edges_from(zip(src.tolist(), dst.tolist()))
        return h

    def subgraph(self, nodes):
        """node-induced subgraph of a single-node-type graph; node features are carried over (DGL's DGLGraph.subgraph)."""
        assert len(self.ntypes) == 1 and len(self.canonical_etypes) == 1
        nt, et = self.ntypes[0], self.canonical_etypes[0]
        nodes = torch.as_tensor(nodes).long()
        new_id = torch.full((self._num_nodes[nt],), -1, dtype=torch.long)
        new_id[nodes] = torch.arange(len(nodes))
        src, dst = self._edges[et]
        keep = (new_id[src] >= 0) & (new_id[dst] >= 0)
        g = DGLGraph({et: (new_id[src[keep]], new_id[dst[keep]])}, {nt: len(nodes)})
        for f, t in self._ndata[nt].items():
            g._ndata[nt][f] = t[nodes]
        return g

    def __deepcopy__(self, memo):
        g = DGLGraph({k: (v[0].clone(), v[1].clone()) for k, v in self._edges.items()},
                     self._num_nodes)
        for nt in self.ntypes:
            for f, t in self._ndata[nt].items():
                g._ndata[nt][f] = t.detach().clone() if not t.requires_grad else _copy.deepcopy(t, memo)
        if self._batch_num_nodes is not None:
            g._batch_num_nodes = {k: v.clone() for k, v in self._batch_num_nodes.items()}
            g._batch_num_edges = {k: v.clone() for k, v in self._batch_num_edges.items()}
        return g


DGLHeteroGraph = DGLGraph
