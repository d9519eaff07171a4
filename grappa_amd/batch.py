"""`MolBatch`: the DGL-free batched molecular graph of the Grappa hot path.

Replaces the reference's DGL heterograph (data/Molecule.py:429-537 `to_dgl`,
utils/dgl_utils.py:11-60 `batch`, :63-82 `unbatch`, :85-120 `delete_dummy_confs`,
:132-171 `set_number_confs`) with one flat container laid out for the HIP kernels:

  * atoms (`n1`): a CSR of the bonded neighbours, destination-major (`indptr[N+1]`,
    `indices[E]`, int32) plus `rev[E]`, the slot of the reverse edge -- the graph is undirected
    (every bond is stored as two directed edges, Molecule.py:465-472), which lets the GAT backward
    gather instead of scatter;
  * tuples (`n2,n3,n4,n4_improper`): dense index tables `idxs (T_s, s)` (int64 in `.data` for
    API parity with the reference, int32 copies for the kernels), molecule segment pointers
    `mol_ptr` (= cumulative `batch_num_nodes`), and the inverse incidence (atom -> token rows)
    used by the atomic-free backward of the tuple gather and by the force kernel;
  * molecule level (`g`): one row per molecule (energy_ref, is_dummy, ...).

It duck-types the DGL subset the reference's callers touch: `g.ntypes`, `g.nodes[nt].data`,
`g.num_nodes(nt)`, `g.batch_num_nodes(nt)`, `g.to(device)`.
"""
from __future__ import annotations

import copy
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .constants import LEVEL_ARITY, TUPLE_LEVELS

NTYPES = ["g", "n1", "n2", "n3", "n4", "n4_improper"]


class _NodeView:
    __slots__ = ("data",)

    def __init__(self, data):
        self.data = data


class _Nodes:
    def __init__(self, g):
        self._g = g

    def __getitem__(self, nt) -> _NodeView:
        if nt not in self._g._data:
            raise KeyError(f"node type {nt} not in graph")
        return _NodeView(self._g._data[nt])


class BatchPlan:
    """Device-resident int32 index structures derived from a MolBatch (built once per batch)."""

    def __init__(self, g: "MolBatch", device):
        dev = torch.device(device)
        N = g.num_nodes("n1")
        src, dst = g._src.cpu().numpy().astype(np.int64), g._dst.cpu().numpy().astype(np.int64)
        E = len(src)
        idx_np = []
        for lvl in TUPLE_LEVELS:
            s = LEVEL_ARITY[lvl]
            idx = g._data[lvl]["idxs"].detach().cpu().numpy().astype(np.int64).reshape(-1, s)
            if idx.shape[0] and (idx.min() < 0 or idx.max() >= N):
                raise AssertionError(
                    f"Encountered idxs up to {idx.max()} at the level g.nodes[{lvl}].data[\"idxs\"], "
                    f"but there are only {N} atom-level-nodes in the graph")
            idx_np.append(idx.astype(np.int32))
        arrays = _plan_arrays_native(N, src, dst, idx_np) if _NATIVE_PLAN else None
        if arrays is None:
            arrays = _plan_arrays_numpy(N, src, dst, idx_np)
        self.N, self.E = N, E
        self.max_degree = arrays.pop("max_degree")
        self.B = g.num_nodes("g")
        arrays["atom_molptr"] = _cum(g._bnn["n1"])
        for lvl, idx in zip(TUPLE_LEVELS, idx_np):
            arrays[f"idx32.{lvl}"] = idx
            arrays[f"mol_ptr.{lvl}"] = _cum(g._bnn[lvl])
        # ONE transfer: every table is a 16-byte aligned view of one flat int32 buffer (a 40-atom molecule's plan was 25 copies of a few
        # hundred bytes each)
        off, layout = 0, {}
        for name, a in arrays.items():
            layout[name] = (off, a.size, a.shape)
            off += (a.size + 3) // 4 * 4
        flat = np.zeros(max(off, 4), dtype=np.int32)
        for name, a in arrays.items():
            o, n, _ = layout[name]
            flat[o:o + n] = a.reshape(-1)
        flat_t = torch.from_numpy(flat).to(dev)
        view = {name: flat_t[o:o + n].view(shape) for name, (o, n, shape) in layout.items()}
        self.indptr, self.indices, self.rev = view["indptr"], view["indices"], view["rev"]
        self.atom_molptr = view["atom_molptr"]
        self.T: Dict[str, int] = {lvl: int(idx.shape[0]) for lvl, idx in zip(TUPLE_LEVELS, idx_np)}
        self.idx32 = {lvl: view[f"idx32.{lvl}"] for lvl in TUPLE_LEVELS}
        self.mol_ptr = {lvl: view[f"mol_ptr.{lvl}"] for lvl in TUPLE_LEVELS}
        self.inv_ptr = {lvl: view[f"inv_ptr.{lvl}"] for lvl in TUPLE_LEVELS}
        self.inv_rows = {lvl: view[f"inv_rows.{lvl}"] for lvl in TUPLE_LEVELS}
        self.inc_ptr, self.inc_code = view["inc_ptr"], view["inc_code"]
        self.device = dev
        self._idx_host = dict(zip(TUPLE_LEVELS, idx_np))      # (position_tables builds its tables from these on the host: no device sync)


import os as _os

_NATIVE_PLAN = _os.environ.get("GRAPPA_HOST_PLAN", "1") not in ("0", "")


def _plan_arrays_native(N, src, dst, idx_np):
    """the index structures from libgrappa_host.so (include/grappa_host.h grappa_plan_build); None if the library is not built"""
    try:
        from . import _hostlib
        out = _hostlib.plan_build(N, src, dst, idx_np)
    except (OSError, RuntimeError) as e:
        if "0-in-degree" in str(e) or "grappa_plan_build failed" in str(e):
            raise
        return None
    arrays = {"indptr": out["indptr"], "indices": out["indices"], "rev": out["rev"], "max_degree": out["max_degree"]}
    for li, lvl in enumerate(TUPLE_LEVELS):
        arrays[f"inv_ptr.{lvl}"] = out["inv_ptr"][li]
        arrays[f"inv_rows.{lvl}"] = out["inv_rows"][li]
    arrays["inc_ptr"], arrays["inc_code"] = out["inc_ptr"], out["inc_code"]
    return arrays


def _plan_arrays_numpy(N, src, dst, idx_np):
    """the same in numpy (GRAPPA_HOST_PLAN=0, or the host library missing): the definition the native builder is tested against"""
    E = len(src)
    # CSR by destination, neighbours ascending by source id (deterministic)
    order = np.lexsort((src, dst))
    s_sorted, d_sorted = src[order], dst[order]
    indptr = np.zeros(N + 1, dtype=np.int64)
    np.add.at(indptr, d_sorted + 1, 1)
    indptr = np.cumsum(indptr)
    # reverse edge slot: edge (u -> v) stored in v's list; its reverse (v -> u) is in u's list
    key = d_sorted * max(N, 1) + s_sorted          # sorted ascending by construction
    rkey = s_sorted * max(N, 1) + d_sorted
    rev = np.searchsorted(key, rkey)
    if E and not (np.all(rev < E) and np.all(key[np.minimum(rev, E - 1)] == rkey)):
        raise ValueError("MolBatch: the n1 graph must contain both directions of every bond")
    if N and np.any(np.diff(indptr) == 0):
        raise RuntimeError("There are 0-in-degree nodes in the graph (every atom must be bonded)")
    arrays = {"indptr": indptr.astype(np.int32), "indices": s_sorted.astype(np.int32), "rev": rev.astype(np.int32),
              "max_degree": int(np.max(np.diff(indptr))) if N else 0}
    inc_atom, inc_code = [], []
    for li, (lvl, idx) in enumerate(zip(TUPLE_LEVELS, idx_np)):
        s = LEVEL_ARITY[lvl]
        idx = idx.astype(np.int64)
        T = idx.shape[0]
        # inverse incidence: atom -> rows (pos*T + t) of the (s, T, F) token table
        atoms = idx.T.reshape(-1)                      # row r = pos*T + t  <->  atoms[r]
        rows = np.argsort(atoms, kind="stable")
        ptr = np.zeros(N + 1, dtype=np.int64)
        np.add.at(ptr, atoms + 1, 1)
        arrays[f"inv_ptr.{lvl}"] = np.cumsum(ptr).astype(np.int32)
        arrays[f"inv_rows.{lvl}"] = rows.astype(np.int32)
        # packed incidence for the force kernel: code = (tuple << 4) | (level << 2) | pos
        pos = np.repeat(np.arange(s, dtype=np.int64), T)
        t = np.tile(np.arange(T, dtype=np.int64), s)
        inc_atom.append(atoms)
        inc_code.append((t << 4) | (li << 2) | pos)
    inc_atom = np.concatenate(inc_atom) if inc_atom else np.zeros(0, np.int64)
    inc_code = np.concatenate(inc_code) if inc_code else np.zeros(0, np.int64)
    o = np.argsort(inc_atom, kind="stable")
    ptr = np.zeros(N + 1, dtype=np.int64)
    np.add.at(ptr, inc_atom + 1, 1)
    if len(inc_code) and inc_code.max() >= 2 ** 31:
        raise ValueError("too many tuples for the packed int32 incidence code")
    arrays["inc_ptr"] = np.cumsum(ptr).astype(np.int32)
    arrays["inc_code"] = inc_code[o].astype(np.int32)
    return arrays


def _plan_position_tables(self, lvl: str):
    cache = self.__dict__.setdefault("_pos_tables", {})
    if lvl not in cache:
        cache[lvl] = _position_tables(self, lvl)
    return cache[lvl]


BatchPlan.position_tables = _plan_position_tables


_POS_CONST: Dict[tuple, tuple] = {}


def _position_tables(plan: "BatchPlan", lvl: str):
    """index tables of the (atom, position) formulation of a writer's first layer (ops.ProjFirstLayerFn), built on the plan's device
    from idx32 and cached: table row pos*N + n holds atom n at position pos.
      idx_id (N, s): idx_id[n, pos] = n                       -- "gather" that lays the table out from the atom rows
      invid_ptr / invid_rows: atom n <- table rows pos*N + n   -- its inverse
      idx_tab (T, s): idx_tab[t, pos] = pos*N + idx[t, pos]    -- tokens from table rows
      invtab_ptr / invtab_rows: table row <- token rows pos*T + t (ascending) -- its inverse"""
    s, N, T = LEVEL_ARITY[lvl], plan.N, plan.T[lvl]
    dev = plan.idx32[lvl].device
    host = getattr(plan, "_idx_host", None)
    if host is not None and _os.environ.get("GRAPPA_HOST_PLAN", "native") != "numpy":
        # natively (libgrappa_host.so grappa_position_tables): 5 us instead of the 40 us of the numpy expressions below, per level
        try:
            from . import _hostlib
            flat, parts = _hostlib.position_tables(N, host[lvl].reshape(T, s))
        except (OSError, RuntimeError, AttributeError):
            flat = None
        if flat is not None:
            ft = torch.from_numpy(flat).to(dev)
            v = [ft[o:o + n] for o, n in parts]
            return v[0].view(N, s), v[1], v[2], v[3].view(T, s), v[4], v[5]
    if host is not None:
        # the plan was built on the host: six numpy expressions and ONE transfer instead of a dozen torch kernels (and bincount's sync)
        idx = host[lvl].astype(np.int64)
        pos = np.arange(s, dtype=np.int64)
        idx_tab = idx + pos.reshape(1, s) * N
        key = idx_tab.T.reshape(-1)
        parts = [np.repeat(np.arange(N, dtype=np.int64), s),                                  # idx_id (N, s)
                 np.arange(N + 1, dtype=np.int64) * s,                                          # invid_ptr
                 (pos.reshape(1, s) * N + np.arange(N, dtype=np.int64).reshape(N, 1)).reshape(-1),   # invid_rows
                 idx_tab.reshape(-1),                                                           # idx_tab (T, s)
                 np.concatenate(([0], np.cumsum(np.bincount(key, minlength=s * N)))),           # invtab_ptr
                 np.argsort(key, kind="stable")]                                                # invtab_rows
        sizes = [p.size for p in parts]
        offs = np.concatenate(([0], np.cumsum([(n + 3) // 4 * 4 for n in sizes])))
        flat = np.zeros(max(int(offs[-1]), 4), dtype=np.int32)
        for p, o in zip(parts, offs):
            flat[o:o + p.size] = p
        ft = torch.from_numpy(flat).to(dev)
        v = [ft[o:o + n] for o, n in zip(offs, sizes)]
        return v[0].view(N, s), v[1], v[2], v[3].view(T, s), v[4], v[5]
    i32 = dict(dtype=torch.int32, device=dev)
    pos = torch.arange(s, **i32)
    # the three tables that depend on (N, s) alone: built once per atom count and device (batches padded to a few shapes ask for the same ones
    # every step: a dozen launches per level saved); read-only everywhere
    ck = (N, s, str(dev))
    const = _POS_CONST.get(ck)
    if const is None:
        if len(_POS_CONST) >= 64:
            _POS_CONST.pop(next(iter(_POS_CONST)))
        const = _POS_CONST[ck] = (torch.arange(N, **i32).view(N, 1).expand(N, s).contiguous(), (torch.arange(N + 1, **i32) * s).contiguous(),
                                  (pos.view(1, s) * N + torch.arange(N, **i32).view(N, 1)).reshape(-1).contiguous())
    idx_id, invid_ptr, invid_rows = const
    idx_tab = (plan.idx32[lvl] + pos.view(1, s) * N).contiguous()
    key = idx_tab.t().reshape(-1).long()                           # token row r = pos*T + t -> its table row
    invtab_rows = torch.argsort(key, stable=True).to(torch.int32).contiguous()
    # (index_add_, not bincount: bincount reads the largest key back to the host -- a synchronisation per level and batch)
    counts = torch.zeros(s * N, dtype=torch.int32, device=dev).index_add_(0, key, torch.ones(key.shape[0], dtype=torch.int32, device=dev))
    invtab_ptr = torch.zeros(s * N + 1, **i32)
    invtab_ptr[1:] = torch.cumsum(counts, 0, dtype=torch.int32)
    return idx_id, invid_ptr, invid_rows, idx_tab, invtab_ptr, invtab_rows


def _cum(counts: np.ndarray) -> np.ndarray:
    out = np.zeros(len(counts) + 1, dtype=np.int32)
    out[1:] = np.cumsum(counts)
    return out


class MolBatch:
    def __init__(self, src: torch.Tensor, dst: torch.Tensor, data: Dict[str, Dict[str, torch.Tensor]],
                 batch_num_nodes: Dict[str, np.ndarray]):
        self._src = src.long()
        self._dst = dst.long()
        self._data = {nt: dict(data.get(nt, {})) for nt in NTYPES}
        self._bnn = {nt: np.asarray(batch_num_nodes[nt], dtype=np.int64) for nt in NTYPES}
        self.ntypes = list(NTYPES)
        self._plan: Optional[BatchPlan] = None

    # ---- DGL-like surface ------------------------------------------------------------------
    @property
    def nodes(self) -> _Nodes:
        return _Nodes(self)

    def num_nodes(self, ntype: Optional[str] = None) -> int:
        if ntype is None:
            return int(sum(v.sum() for v in self._bnn.values()))
        return int(self._bnn[ntype].sum())

    number_of_nodes = num_nodes

    def num_edges(self, etype=None) -> int:
        return int(self._src.shape[0])

    def batch_num_nodes(self, ntype: str) -> torch.Tensor:
        return torch.from_numpy(self._bnn[ntype].copy()).to(self.device)

    @property
    def batch_size(self) -> int:
        return len(self._bnn["g"])

    @property
    def device(self):
        return self._src.device

    def edges(self):
        return self._src, self._dst

    def to(self, device) -> "MolBatch":
        g = MolBatch(self._src.to(device), self._dst.to(device),
                     {nt: {k: v.to(device) for k, v in d.items()} for nt, d in self._data.items()},
                     self._bnn)
        if self._plan is not None and self._plan.device == torch.device(device):
            g._plan = self._plan
        return g

    def cuda(self, device=None) -> "MolBatch":
        return self.to("cuda" if device is None else device)

    def cpu(self) -> "MolBatch":
        return self.to("cpu")

    def plan(self) -> BatchPlan:
        """Index structures on the graph's device (cached)."""
        if self._plan is None or self._plan.device != self.device:
            self._plan = BatchPlan(self, self.device)
        return self._plan

    def __deepcopy__(self, memo):
        return MolBatch(self._src.clone(), self._dst.clone(),
                        {nt: {k: v.detach().clone() for k, v in d.items()} for nt, d in self._data.items()},
                        {k: v.copy() for k, v in self._bnn.items()})

    def __repr__(self):
        return (f"MolBatch(B={self.batch_size}, atoms={self.num_nodes('n1')}, "
                + ", ".join(f"{l}={self.num_nodes(l)}" for l in TUPLE_LEVELS) + f", device={self.device})")


# ---------------------------------------------------------------------------------------------
def single_graph(n_atoms: int, bonds: np.ndarray, idxs: Dict[str, np.ndarray],
                 n1_data: Dict[str, torch.Tensor], ids: Optional[np.ndarray] = None) -> MolBatch:
    """One molecule.  bonds: (n_bonds, 2) atom *indices*; both directions are stored,
    src = cat(b0, b1), dst = cat(b1, b0) as the reference does (Molecule.py:465-472)."""
    b = torch.as_tensor(np.asarray(bonds, dtype=np.int64).reshape(-1, 2))
    if len(b) and int(b.max()) >= n_atoms:
        raise AssertionError(f"Maximal atom index in bonds ({int(b.max())}) must be smaller than the number of atoms ({n_atoms})")
    if len(torch.unique(b.flatten())) != n_atoms:
        missing = np.setdiff1d(np.arange(n_atoms), np.unique(b.numpy()))
        raise AssertionError(f"Every atom must be part of a bond but {len(torch.unique(b.flatten()))} of {n_atoms} atoms "
                             f"are in the bonds. Atoms {missing.tolist()} are not part of a bond.")
    src = torch.cat((b[:, 0], b[:, 1]))
    dst = torch.cat((b[:, 1], b[:, 0]))
    data = {"n1": dict(n1_data), "g": {}}
    if ids is not None:
        data["n1"]["ids"] = torch.as_tensor(np.asarray(ids), dtype=torch.int64)
    bnn = {"g": np.array([1]), "n1": np.array([n_atoms])}
    for lvl in TUPLE_LEVELS:
        t = torch.as_tensor(np.asarray(idxs[lvl], dtype=np.int64).reshape(-1, LEVEL_ARITY[lvl]))
        data[lvl] = {"idxs": t}
        bnn[lvl] = np.array([t.shape[0]])
    return MolBatch(src, dst, data, bnn)


def batch(graphs: Sequence[MolBatch], deep_copies_of_same_n_atoms: bool = False) -> MolBatch:
    """Concatenate molecules; `idxs` are shifted by the cumulative atom offset
    (reference utils/dgl_utils.py:11-60).  The inputs are not modified."""
    graphs = list(graphs)
    assert len(graphs) > 0
    num_confs = None
    if "xyz" in graphs[0]._data["n1"]:
        num_confs = graphs[0]._data["n1"]["xyz"].shape[1]
    offsets = np.zeros(len(graphs) + 1, dtype=np.int64)
    for i, g in enumerate(graphs):
        offsets[i + 1] = offsets[i] + g.num_nodes("n1")
        if num_confs is not None and g._data["n1"]["xyz"].shape[1] != num_confs:
            raise ValueError(f"All graphs must have the same number of conformations but found {num_confs} "
                             f"and {g._data['n1']['xyz'].shape[1]}")
    src = torch.cat([g._src + int(off) for g, off in zip(graphs, offsets)])
    dst = torch.cat([g._dst + int(off) for g, off in zip(graphs, offsets)])
    data: Dict[str, Dict[str, torch.Tensor]] = {}
    for nt in NTYPES:
        data[nt] = {}
        for k in graphs[0]._data[nt].keys():
            parts = []
            for g, off in zip(graphs, offsets):
                if k not in g._data[nt]:
                    raise KeyError(f"feature {k} of node type {nt} is missing in one of the graphs")
                v = g._data[nt][k]
                if k == "idxs" and nt in TUPLE_LEVELS:
                    v = v + int(off)
                parts.append(v)
            data[nt][k] = torch.cat(parts, dim=0)
    bnn = {nt: np.concatenate([g._bnn[nt] for g in graphs]) for nt in NTYPES}
    return MolBatch(src, dst, data, bnn)


def unbatch(g: MolBatch) -> List[MolBatch]:
    """Inverse of `batch` incl. removal of dummy conformations (reference utils/dgl_utils.py:63-82)."""
    B = g.batch_size
    ptr = {nt: _cum(g._bnn[nt]) for nt in NTYPES}
    # edges are stored molecule by molecule only if built by batch(); split by the molecule of the destination
    mol_of_atom = np.repeat(np.arange(B), g._bnn["n1"])
    emol = mol_of_atom[g._dst.cpu().numpy()] if g.num_edges() else np.zeros(0, np.int64)
    out = []
    for i in range(B):
        a0 = int(ptr["n1"][i])
        sel = torch.from_numpy(np.nonzero(emol == i)[0]).to(g._src.device)
        data = {}
        for nt in NTYPES:
            sl = slice(int(ptr[nt][i]), int(ptr[nt][i + 1]))
            data[nt] = {}
            for k, v in g._data[nt].items():
                v = v[sl]
                if k == "idxs" and nt in TUPLE_LEVELS:
                    v = v - a0
                data[nt][k] = v
        sub = MolBatch(g._src[sel] - a0, g._dst[sel] - a0, data, {nt: g._bnn[nt][i:i + 1] for nt in NTYPES})
        out.append(delete_dummy_confs(sub))
    return out


def delete_dummy_confs(g: MolBatch) -> MolBatch:
    """Drop conformations with is_dummy == 1 from xyz / *energy* / *gradient* features of a
    single-molecule graph (reference utils/dgl_utils.py:85-120)."""
    if "is_dummy" not in g._data["g"]:
        return g
    mask = g._data["g"]["is_dummy"][0] == 0
    if bool(mask.all()):
        return g
    g._data["g"]["is_dummy"] = g._data["g"]["is_dummy"][:, mask]
    g._data["n1"]["xyz"] = g._data["n1"]["xyz"][:, mask, :]
    if torch.isnan(g._data["n1"]["xyz"]).any():
        raise RuntimeError("Found nan in xyz after unbatching")
    for k in list(g._data["g"].keys()):
        if "energy" in k:
            assert g._data["g"][k].shape[0] == 1, "Internal error while unbatching."
            g._data["g"][k] = g._data["g"][k][:, mask]
    for k in list(g._data["n1"].keys()):
        if "gradient" in k:
            g._data["n1"][k] = g._data["n1"][k][:, mask, :]
    for lvl in TUPLE_LEVELS:
        for k in list(g._data[lvl].keys()):
            if "energy" in k:
                g._data[lvl][k] = g._data[lvl][k][:, mask]
    return g


def set_number_confs(g: MolBatch, num_confs: int, seed: Optional[int] = None) -> MolBatch:
    """Sub-sample (randperm) or pad (repeat the last conformation, flagged in `is_dummy`) the
    conformations of a single-molecule graph (reference utils/dgl_utils.py:132-171)."""
    if "xyz" not in g._data["n1"]:
        return g
    present = g._data["n1"]["xyz"].shape[1]
    if seed is not None:
        torch.manual_seed(seed)
    if present == num_confs:
        g._data["g"]["is_dummy"] = torch.zeros((1, num_confs), dtype=torch.float32)
        return g
    if present > num_confs:
        g._data["g"]["is_dummy"] = torch.zeros((1, num_confs), dtype=torch.float32)
        conf_idxs = torch.randperm(present)[:num_confs]
    else:
        g._data["g"]["is_dummy"] = torch.cat((torch.zeros((1, present)), torch.ones((1, num_confs - present))), dim=-1)
        conf_idxs = torch.cat((torch.arange(present), torch.full((num_confs - present,), present - 1, dtype=torch.long)))
    g._data["n1"]["xyz"] = g._data["n1"]["xyz"][:, conf_idxs]
    for k in list(g._data["g"].keys()):
        if "energy" in k:
            g._data["g"][k] = g._data["g"][k][:, conf_idxs]
            if torch.isnan(g._data["g"][k]).any():
                raise RuntimeError(f"Found nan in {k} after setting number of conformations to {num_confs}")
    for k in list(g._data["n1"].keys()):
        if "gradient" in k:
            g._data["n1"][k] = g._data["n1"][k][:, conf_idxs]
    return g


def check_disconnected_graphs(g: MolBatch, print_information: bool = True, reference_water_guard: bool = False) -> None:
    """Water guard of `Grappa.predict` (reference utils/dgl_utils.py:210-236): raise if a
    connected component has exactly three atoms with elements {H, O}.
    reference_water_guard=True: the reference's check LITERALLY -- it compares argmax(one-hot atomic number) = Z - 1 with {1, 8}
    (utils/dgl_utils.py:231-234 with data/Molecule.py:521), i.e. looks for {He, F}, which no three-atom component of a real system has:
    it never fires on water, and with this switch neither does this one (results identical to the reference's on the same inputs)."""
    n = g.num_nodes("n1")
    src, dst = g._src.cpu().numpy(), g._dst.cpu().numpy()
    try:                                              # connected components natively: the Python union-find was 0.2 - 0.45 ms of a 3 ms predict
        from . import _hostlib
        roots = _hostlib.components(n, src, dst)
    except (OSError, RuntimeError):
        parent = np.arange(n)

        def find(a):
            while parent[a] != a:
                parent[a] = parent[parent[a]]
                a = parent[a]
            return a

        for a, b in zip(src.tolist(), dst.tolist()):
            ra, rb = find(a), find(b)
            if ra != rb:
                parent[ra] = rb
        roots = np.array([find(a) for a in range(n)])
    comps, counts = np.unique(roots, return_counts=True)
    if print_information and len(comps) > 1:
        print(f"Found {len(comps)} disconnected subgraphs of lengths {counts[:3].tolist()}...")
    if "atomic_number" not in g._data["n1"]:
        return
    z = torch.argmax(g._data["n1"]["atomic_number"], dim=-1).cpu().numpy()
    for c, cnt in zip(comps, counts):
        if cnt == 3:
            els = set(z[roots == c].tolist())
            if els == ({1, 8} if reference_water_guard else {0, 7}):
                raise ValueError("Found a water molecule in the graph. Grappa can currently not parametrize water "
                                 "molecules. Strip the water, parametrize and solvate then.")
