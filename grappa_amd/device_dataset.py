"""`DeviceDataset`: a training set resident in HBM, batches assembled on the device.

The reference collates every batch on the host (data/GraphDataLoader.py:23-73 `collate_fn`: per-graph deep copies,
`set_number_confs`, `dgl.batch`) and copies the result to the GPU.  At the throughput of the HIP path (~5,000 molecules/s on one
MI355X) that host work -- 90 ms per 256 molecules for collate + index plan in this package's own host path -- is longer than
the train step it feeds.  An MI355X holds 288 GB: a whole Espaloma-sized dataset (10^5 molecules x 10^2 conformations ~ 10 GB)
fits many times over.  So every per-molecule table is packed ONCE into flat device arrays -- features, conformations, tuple
tables AND the per-molecule pieces of the index plan (CSR, reverse-edge slots, inverse incidences: the batch graph is block
diagonal, so the plan of a batch is the concatenation of the molecules' plans with shifted indices) -- and a batch is

    host:   pick ids, prefix sums of the per-molecule row counts (numpy, microseconds), conformation selection (the same
            seeded `torch.randperm` calls as `set_number_confs`, so a batch is bit-identical to the host collate's)
    device: ONE launch of `grappa_collate_batch` (grid: slots x tables) + six prefix sums for the pointer arrays.

`collate(ids, conf_strategy)` returns `(MolBatch, names)` like the reference's collate_fn, with the `BatchPlan` already attached.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from .backend import get_backend
from .batch import NTYPES, BatchPlan, MolBatch, delete_dummy_confs
from .constants import LEVEL_ARITY, TUPLE_LEVELS
from .dataloader import _shallow_copy


def _as_words(t: torch.Tensor) -> Tuple[torch.Tensor, int]:
    """(rows, ...) tensor -> (flat int32 view, 4-byte words per row)"""
    if t.dtype not in (torch.float32, torch.int32, torch.int64):
        raise TypeError(f"DeviceDataset packs float32 / int32 / int64 features, got {t.dtype}")
    t = t.contiguous()
    rows = t.shape[0]
    per_row = int(np.prod(t.shape[1:])) if t.dim() > 1 else 1
    width = per_row * (2 if t.dtype == torch.int64 else 1)
    words = t.reshape(-1).view(torch.int32) if t.numel() else torch.zeros(rows * width, dtype=torch.int32)
    return words, width


class _Table:
    """one packed table: `data` (device, int32 words), per-molecule first row and row count (host), words per row"""

    def __init__(self, parts: List[torch.Tensor], device, like: torch.Tensor, conf: bool = False):
        self.dtype, self.trailing = like.dtype, tuple(like.shape[2:] if conf else like.shape[1:])
        words, rows, starts, acc = [], [], [], 0
        self.width = None
        for p in parts:
            if p.dtype != self.dtype or tuple(p.shape[2:] if conf else p.shape[1:]) != self.trailing:
                raise ValueError("every molecule must carry the feature with the same dtype and trailing shape")
            if conf:                                   # (rows, C_mol, ...): one source row = C_mol * k words, element-addressed
                k = int(np.prod(self.trailing)) if self.trailing else 1
                k *= 2 if p.dtype == torch.int64 else 1
                w = p.contiguous().reshape(-1).view(torch.int32) if p.numel() else torch.zeros(0, dtype=torch.int32)
                words.append(w)
                starts.append(acc)
                acc += w.numel()
                rows.append(p.shape[0])
                self.width = k
            else:
                w, width = _as_words(p)
                words.append(w)
                starts.append(acc)
                acc += p.shape[0]
                rows.append(p.shape[0])
                self.width = width if self.width is None else self.width
                if width != self.width:
                    raise ValueError("inconsistent feature width")
        self.data = (torch.cat(words) if words else torch.zeros(0, dtype=torch.int32)).to(device)
        if self.data.numel() == 0:
            self.data = torch.zeros(1, dtype=torch.int32, device=device)       # a valid pointer for empty tables
        self.start = np.asarray(starts, dtype=np.int64)       # first row (conf tables: first WORD) of every molecule
        self.rows = np.asarray(rows, dtype=np.int64)
        self.conf = conf
        self.used = acc                                       # rows (conf tables: words) the molecules occupy
        self.pad_cap = 0

    def reserve(self, rows_cap: int) -> None:
        """room for ONE more molecule of up to rows_cap rows behind the packed ones (entry len(start) - 1 afterwards): the padding molecule of
        `DeviceDataset.collate(pad_to=...)`, rewritten for every batch.  Conformational tables hold it with one conformation."""
        words_cap = max(int(rows_cap), 1) * self.width
        base_words = self.used if self.conf else self.used * self.width
        data = torch.zeros(base_words + words_cap, dtype=torch.int32, device=self.data.device)
        n = min(base_words, self.data.numel())
        data[:n] = self.data[:n]
        self.data = data
        if self.pad_cap == 0:
            self.start = np.concatenate([self.start, [self.used]])
            self.rows = np.concatenate([self.rows, [0]])
        self.pad_cap = max(int(rows_cap), 1)
        self.pad_words0 = base_words

    def write_pad(self, part: torch.Tensor, todo: list) -> None:
        """the padding molecule's rows of this table (host tensor, conformational tables: (rows, 1, ...)) for the reserved region: appended to
        `todo` as (destination view, host words) -- DeviceDataset._write_pad sends all tables in ONE transfer"""
        rows = int(part.shape[0])
        if rows > self.pad_cap:
            raise ValueError(f"padding molecule: {rows} rows, {self.pad_cap} reserved")
        self.rows[-1] = rows
        if rows == 0 or part.numel() == 0:
            return
        w = part.contiguous().reshape(-1).view(torch.int32)
        if w.numel() != rows * self.width:
            raise ValueError("padding molecule: a table of another width than the dataset's")
        todo.append((self.data[self.pad_words0:self.pad_words0 + w.numel()], w))


class DeviceDataset:
    def __init__(self, items: Sequence[Tuple[MolBatch, str]], device="cuda"):
        self.device = torch.device(device)
        graphs = [delete_dummy_confs(_shallow_copy(g)) for g, _ in items]
        self.names = [n for _, n in items]
        M = len(graphs)
        if M == 0:
            raise ValueError("empty dataset")
        for g in graphs:
            if g.batch_size != 1:
                raise ValueError("DeviceDataset takes single-molecule graphs")
        self.has_confs = "xyz" in graphs[0]._data["n1"]
        self.n_confs = np.array([g._data["n1"]["xyz"].shape[1] if self.has_confs else 0 for g in graphs], dtype=np.int64)
        # ---- per-molecule index plans (host, once): the batch plan is their concatenation with shifted indices
        plans = [BatchPlan(g, "cpu") for g in graphs]
        self.count = {nt: np.array([g.num_nodes(nt) for g in graphs], dtype=np.int64) for nt in NTYPES}
        self.n_edges = np.array([p.E for p in plans], dtype=np.int64)
        self.max_degree = np.array([p.max_degree for p in plans], dtype=np.int64)
        dev = self.device
        T = {}
        T["deg"] = _Table([p.indptr[1:] - p.indptr[:-1] for p in plans], dev, plans[0].indptr)
        T["indices"] = _Table([p.indices for p in plans], dev, plans[0].indices)
        T["rev"] = _Table([p.rev for p in plans], dev, plans[0].rev)
        for lvl in TUPLE_LEVELS:
            T[f"idx/{lvl}"] = _Table([p.idx32[lvl] for p in plans], dev, plans[0].idx32[lvl])
            T[f"inv_cnt/{lvl}"] = _Table([p.inv_ptr[lvl][1:] - p.inv_ptr[lvl][:-1] for p in plans], dev, plans[0].indptr)
            T[f"inv_rows/{lvl}"] = _Table([p.inv_rows[lvl] for p in plans], dev, plans[0].indices)
        T["inc_cnt"] = _Table([p.inc_ptr[1:] - p.inc_ptr[:-1] for p in plans], dev, plans[0].indptr)
        T["inc_code"] = _Table([p.inc_code for p in plans], dev, plans[0].indices)
        self.plan_tables = T
        # ---- features: plain row tables and conformational tables (same key rules as batch() / set_number_confs)
        self.feat: Dict[Tuple[str, str], _Table] = {}
        for nt in NTYPES:
            for k, v0 in graphs[0]._data[nt].items():
                if (nt in TUPLE_LEVELS and k == "idxs") or (nt == "g" and k == "is_dummy"):
                    continue
                parts = []
                for g in graphs:
                    if k not in g._data[nt]:
                        raise KeyError(f"feature {k} of node type {nt} is missing in one of the graphs")
                    parts.append(g._data[nt][k])
                conf = self.has_confs and ((nt == "n1" and (k == "xyz" or "gradient" in k)) or (nt == "g" and "energy" in k))
                if nt in TUPLE_LEVELS and "energy" in k:
                    raise NotImplementedError("per-tuple energies are outputs of the model, not dataset features")
                self.feat[(nt, k)] = _Table(parts, dev, v0, conf=conf)
        self._i64 = self._i32 = None
        self.pad_caps: Optional[Dict[str, int]] = None        # enable_padding(): the largest padding molecule the tables have room for
        # every bond is two directed edges and one n2 tuple (Molecule.py:465-472): the padding molecule keeps that, so a padded batch has E = 2 T_n2
        self.bonds_are_n2 = bool(np.all(self.n_edges == 2 * self.count["n2"]))

    def __len__(self) -> int:
        return len(self.names)

    # ------------------------------------------------------------------------------------------------------------------
    def _n_confs_of(self, counts: np.ndarray, conf_strategy: Union[str, int]) -> int:
        if isinstance(conf_strategy, int):
            return int(min(conf_strategy, counts.max()))
        if conf_strategy == "min":
            return int(counts.min())
        if conf_strategy in ("max", "all"):
            return int(counts.max())
        if conf_strategy == "mean":
            return int(np.mean(counts))
        raise ValueError(f"Unknown conf_strategy: {conf_strategy}")

    # ---- batches of a fixed shape: a padding molecule behind the real ones ---------------------------------------------------------
    PAD_DIMS = ("n1", "n2", "n3", "n4", "n4_improper")

    def totals(self, ids) -> Dict[str, int]:
        """rows of every level the batch of molecules `ids` has (host arithmetic on the per-molecule counts)"""
        ids = np.asarray(ids, dtype=np.int64)
        return {nt: int(self.count[nt][ids].sum()) for nt in self.PAD_DIMS}

    @staticmethod
    def pad_sizes(totals: Dict[str, int], caps: Dict[str, int]) -> Optional[Dict[str, int]]:
        """rows the padding molecule needs at every level to bring a batch of `totals` to `caps`; None if no valid molecule does: it needs
        at least four atoms (its tuples are runs of consecutive atoms), every atom in a bond (the index plan refuses unbonded atoms: at least
        ceil(n / 2) bonds) and not more bonds than atom pairs"""
        p = {nt: int(caps[nt]) - int(totals[nt]) for nt in DeviceDataset.PAD_DIMS}
        n, b = p["n1"], p["n2"]
        if n < 4 or min(p.values()) < 0 or b < (n + 1) // 2 or b > n * (n - 1) // 2:
            return None
        return p

    def enable_padding(self, max_pad: Dict[str, int]) -> None:
        """room for a padding molecule of up to max_pad[level] rows in every packed table (entry len(self) of the tables: not a molecule of the
        dataset -- `names`, `len()` and the samplers do not see it)"""
        if not self.bonds_are_n2:
            raise ValueError("padded batches need a dataset whose bonds are its n2 tuples (two directed edges per n2 row)")
        caps = {nt: max(int(max_pad.get(nt, 0)), 4 if nt == "n1" else 0) for nt in self.PAD_DIMS}
        if self.pad_caps is not None and all(caps[k] <= self.pad_caps[k] for k in caps):
            return
        if self.pad_caps is not None:
            caps = {k: max(caps[k], self.pad_caps[k]) for k in caps}
        first = self.pad_caps is None
        N, E = caps["n1"], 2 * caps["n2"]
        P = self.plan_tables
        for name in ("deg", "inc_cnt"):
            P[name].reserve(N)
        P["indices"].reserve(E)
        P["rev"].reserve(E)
        inc = 0
        for lvl in TUPLE_LEVELS:
            P[f"idx/{lvl}"].reserve(caps[lvl])
            P[f"inv_cnt/{lvl}"].reserve(N)
            P[f"inv_rows/{lvl}"].reserve(caps[lvl] * LEVEL_ARITY[lvl])
            inc += caps[lvl] * LEVEL_ARITY[lvl]
        P["inc_code"].reserve(inc)
        for (nt, k), table in self.feat.items():
            table.reserve(1 if nt == "g" else caps[nt])
        if first:
            for nt in NTYPES:
                self.count[nt] = np.concatenate([self.count[nt], [1 if nt == "g" else 0]])
            self.n_edges = np.concatenate([self.n_edges, [0]])
            self.max_degree = np.concatenate([self.max_degree, [0]])
            self.n_confs = np.concatenate([self.n_confs, [1 if self.has_confs else 0]])
        self.pad_caps = caps

    @staticmethod
    def _pad_topology(p: Dict[str, int]):
        """the padding molecule: n atoms on a zigzag (no two coincide, no three in a line, no four in a plane: every internal coordinate and its
        derivative is finite), bonds = the first b of [pairs covering every atom, the rest of the chain, chords (i, i + d) of growing span d],
        tuples = runs of consecutive atoms, repeated cyclically"""
        n, b = p["n1"], p["n2"]
        i = np.arange(n - 1, dtype=np.int64)
        # first the bonds that put every atom into one -- (0,1), (2,3), ... and (n-2, n-1) for an odd n --, then the rest of the chain, then chords
        first = i[::2] if n % 2 == 0 else np.concatenate([i[:-1:2], [n - 2]])
        rest = np.setdiff1d(i, first)
        i = np.concatenate([first, rest])
        bonds = [np.stack([i, i + 1], axis=1)]
        have, d = n - 1, 2
        while have < b:
            j = np.arange(min(n - d, b - have), dtype=np.int64)
            bonds.append(np.stack([j, j + d], axis=1))
            have += len(j)
            d += 1
        bonds = np.concatenate(bonds)[:b]
        idxs = {"n2": bonds}
        for lvl in ("n3", "n4", "n4_improper"):
            s = LEVEL_ARITY[lvl]
            first = np.arange(p[lvl], dtype=np.int64) % (n - s + 1)
            idxs[lvl] = first.reshape(-1, 1) + np.arange(s, dtype=np.int64).reshape(1, s)
        a = np.arange(n, dtype=np.float64)
        # (the plain zigzag has every four consecutive atoms in one plane: the two incommensurate wobbles take them out of it)
        xyz = np.stack([1.3 * a, 0.9 * (np.arange(n) % 2) + 0.23 * np.sin(1.7 * a + 0.3), 0.7 * ((np.arange(n) // 2) % 2) + 0.31 * np.cos(2.9 * a + 0.1)],
                       axis=1).astype(np.float32)
        return bonds, idxs, xyz

    def _write_pad(self, p: Dict[str, int]) -> None:
        """build the padding molecule of p[level] rows on the host (graph, index plan: the same BatchPlan code as for a molecule of the dataset)
        and write it into the reserved region of every table; features are zeros except the conformation (one, the zigzag)"""
        if self.pad_caps is None or any(p[k] > self.pad_caps[k] for k in self.PAD_DIMS):
            raise ValueError(f"padding molecule {p} exceeds the reserved room {self.pad_caps}: call enable_padding with larger sizes")
        bonds, idxs, xyz = self._pad_topology(p)
        n = p["n1"]
        src = torch.from_numpy(np.concatenate([bonds[:, 0], bonds[:, 1]]))
        dst = torch.from_numpy(np.concatenate([bonds[:, 1], bonds[:, 0]]))
        data = {nt: {} for nt in NTYPES}
        for lvl in TUPLE_LEVELS:
            data[lvl]["idxs"] = torch.from_numpy(idxs[lvl].reshape(-1, LEVEL_ARITY[lvl]))
        bnn = {nt: np.array([1 if nt == "g" else p[nt]], dtype=np.int64) for nt in NTYPES}
        plan = BatchPlan(MolBatch(src, dst, data, bnn), "cpu")
        P = self.plan_tables
        todo: list = []
        P["deg"].write_pad(plan.indptr[1:] - plan.indptr[:-1], todo)
        P["indices"].write_pad(plan.indices, todo)
        P["rev"].write_pad(plan.rev, todo)
        for lvl in TUPLE_LEVELS:
            P[f"idx/{lvl}"].write_pad(plan.idx32[lvl], todo)
            P[f"inv_cnt/{lvl}"].write_pad(plan.inv_ptr[lvl][1:] - plan.inv_ptr[lvl][:-1], todo)
            P[f"inv_rows/{lvl}"].write_pad(plan.inv_rows[lvl], todo)
        P["inc_cnt"].write_pad(plan.inc_ptr[1:] - plan.inc_ptr[:-1], todo)
        P["inc_code"].write_pad(plan.inc_code, todo)
        for (nt, k), table in self.feat.items():
            rows = 1 if nt == "g" else p[nt]
            if nt == "n1" and k == "xyz" and table.conf:
                part = torch.from_numpy(xyz).view(n, 1, 3)
            else:
                part = torch.zeros((rows,) + ((1,) if table.conf else ()) + table.trailing, dtype=table.dtype)
            table.write_pad(part, todo)
        if todo:
            # ONE pinned buffer, ONE transfer, one multi-tensor copy into the tables' reserved regions (was: ~40 small transfers per batch)
            flat = torch.cat([w for _, w in todo])
            if self.device.type == "cuda":
                flat = flat.pin_memory()
            stage = flat.to(self.device, non_blocking=True)
            srcs, o = [], 0
            for _, w in todo:
                srcs.append(stage[o:o + w.numel()])
                o += w.numel()
            torch._foreach_copy_([d for d, _ in todo], srcs)
            self._pad_stage = (flat, stage)
        for nt in self.PAD_DIMS:
            self.count[nt][-1] = p[nt]
        self.n_edges[-1] = plan.E
        self.max_degree[-1] = plan.max_degree

    def collate(self, ids: Sequence[int], conf_strategy: Union[str, int] = "min", pad_to: Optional[Dict[str, int]] = None) -> Tuple[MolBatch, Tuple[str, ...]]:
        """the batch of molecules `ids` (in that order), assembled on the device; same result as
        `get_collate_fn(conf_strategy)([dataset[i] for i in ids])` followed by `.to(device)` and `.plan()`.

        pad_to = {level: rows} (n1, n2, n3, n4, n4_improper): a batch of EXACTLY these sizes -- the molecules `ids` followed by ONE padding
        molecule (`_write_pad`) that owns the missing rows of every level; all its conformations are dummies, `plan.n_real_mols = len(ids)` and
        the loss runs over the real molecules only, so the padding rows get exactly zero gradient.  The conformation selection draws the same
        random numbers as without padding.  Raises ValueError if no valid padding molecule fills the gap (`pad_sizes`)."""
        ids = np.asarray(ids, dtype=np.int64)
        n_real = len(ids)
        if pad_to is not None:
            p = self.pad_sizes(self.totals(ids), pad_to)
            if p is None:
                raise ValueError(f"collate: the batch ({self.totals(ids)}) cannot be padded to {dict(pad_to)}")
            self._write_pad(p)
            ids = np.concatenate([ids, [len(self.names)]])
        B, dev = len(ids), self.device
        cnt = {nt: self.count[nt][ids] for nt in NTYPES}
        off = {nt: np.concatenate([[0], np.cumsum(cnt[nt])]) for nt in NTYPES}
        n_e = self.n_edges[ids]
        e_off = np.concatenate([[0], np.cumsum(n_e)])
        N, E = int(off["n1"][-1]), int(e_off[-1])
        # ---- conformation selection, the same torch calls and order as set_number_confs (utils/dgl_utils.py:132-171)
        n_out, sel, is_dummy = 0, None, None
        if self.has_confs:
            present = self.n_confs[ids[:n_real]]
            n_out = self._n_confs_of(present, conf_strategy)
            sel = np.zeros((B, n_out), dtype=np.int32)
            is_dummy = np.zeros((B, n_out), dtype=np.float32)
            is_dummy[n_real:] = 1.0                    # (the padding molecule: its one conformation repeated, all dummies)
            ar = np.arange(n_out, dtype=np.int32)
            for j, c in enumerate(present.tolist()):
                if c == n_out:
                    sel[j] = ar
                elif c > n_out:
                    sel[j] = torch.randperm(c)[:n_out].numpy()
                else:
                    sel[j, :c] = ar[:c]
                    sel[j, c:] = c - 1
                    is_dummy[j, c:] = 1.0
        # ---- every small per-batch array goes up in two transfers (int64 offsets, int32 parameters)
        i64, i32 = [], []

        def put64(a):
            i64.append(np.ascontiguousarray(a, dtype=np.int64))
            return (sum(x.size for x in i64[:-1]), i64[-1].size)

        def put32(a):
            i32.append(np.ascontiguousarray(a, dtype=np.int32).reshape(-1))
            return (sum(x.size for x in i32[:-1]), i32[-1].size)

        specs = []     # (table, dst tensor, dst_row slot, src_row slot, mode, p0 slot, p1 slot, c0, width)
        atom_off32 = put32(off["n1"][:-1])
        edge_off32 = put32(e_off[:-1])
        conf_cnt32 = put32(self.n_confs[ids]) if self.has_confs else None
        sel32 = put32(sel) if self.has_confs else None
        row_ptr = {"n1": put64(off["n1"]), "g": put64(off["g"]), "edge": put64(e_off)}
        t_off32, t_cnt32 = {}, {}
        for lvl in TUPLE_LEVELS:
            row_ptr[lvl] = put64(off[lvl])
            row_ptr["inv/" + lvl] = put64(off[lvl] * LEVEL_ARITY[lvl])
            t_off32[lvl] = put32(off[lvl][:-1])
            t_cnt32[lvl] = put32(cnt[lvl])
        inc_rows = sum(cnt[lvl] * LEVEL_ARITY[lvl] for lvl in TUPLE_LEVELS)
        row_ptr["inc"] = put64(np.concatenate([[0], np.cumsum(inc_rows)]))
        t_off4 = put32(np.stack([off[lvl][:-1] for lvl in TUPLE_LEVELS], axis=1))
        # (the plan's molecule pointers and the dummy mask ride in the same pinned upload: a pageable host-to-device copy of their own would
        #  wait for everything queued on the stream -- the previous train step -- before the host may go on)
        molptr32 = {"n1": put32(off["n1"])}
        for lvl in TUPLE_LEVELS:
            molptr32[lvl] = put32(off[lvl])
        dummy32 = put32(is_dummy.view(np.int32)) if self.has_confs else None
        out: Dict[str, torch.Tensor] = {}

        def add(name, table: _Table, rows_total, dst_key, mode, p0=None, p1=None, c0=0, words_per_row=None):
            w = table.width if words_per_row is None else words_per_row
            dst = torch.empty(max(int(rows_total) * w * (n_out if table.conf else 1), 1), dtype=torch.int32, device=dev)
            out[name] = dst
            specs.append((table, dst, row_ptr[dst_key], put64(table.start[ids]), mode, p0, p1, c0, w))

        P = self.plan_tables
        add("deg", P["deg"], N, "n1", "copy")
        add("indices", P["indices"], E, "edge", "add", p0=atom_off32)
        add("rev", P["rev"], E, "edge", "add", p0=edge_off32)
        for lvl in TUPLE_LEVELS:
            Tl = int(off[lvl][-1])
            add(f"idx/{lvl}", P[f"idx/{lvl}"], Tl, lvl, "add", p0=atom_off32)
            add(f"inv_cnt/{lvl}", P[f"inv_cnt/{lvl}"], N, "n1", "copy")
            add(f"inv_rows/{lvl}", P[f"inv_rows/{lvl}"], Tl * LEVEL_ARITY[lvl], "inv/" + lvl, "inv_rows", p0=t_cnt32[lvl], p1=t_off32[lvl], c0=Tl)
        add("inc_cnt", P["inc_cnt"], N, "n1", "copy")
        add("inc_code", P["inc_code"], int(inc_rows.sum()), "inc", "inc_code", p0=t_off4)
        for (nt, k), table in self.feat.items():
            rows_total = int(off[nt][-1])
            if table.conf:
                add(f"feat/{nt}/{k}", table, rows_total, nt, "conf", p0=conf_cnt32, p1=sel32, c0=n_out)
            else:
                add(f"feat/{nt}/{k}", table, rows_total, nt, "copy")
        h64 = torch.from_numpy(np.concatenate(i64)) if i64 else torch.zeros(0, dtype=torch.int64)
        h32 = torch.from_numpy(np.concatenate(i32)) if i32 else torch.zeros(0, dtype=torch.int32)
        if dev.type == "cuda":
            h64, h32 = h64.pin_memory(), h32.pin_memory()
        d64, d32 = h64.to(dev, non_blocking=True), h32.to(dev, non_blocking=True)
        self._i64, self._i32 = (h64, d64), (h32, d32)          # alive until the next collate: the copies and the kernel are asynchronous
        tables = []
        for table, dst, dst_row, src_row, mode, p0, p1, c0, w in specs:
            t = {"src": table.data, "dst": dst, "dst_row": d64[dst_row[0]:dst_row[0] + dst_row[1]], "src_row": d64[src_row[0]:src_row[0] + src_row[1]],
                 "mode": mode, "width": w, "c0": c0}
            if p0 is not None:
                t["p0"] = d32[p0[0]:p0[0] + p0[1]]
            if p1 is not None:
                t["p1"] = d32[p1[0]:p1[0] + p1[1]]
            tables.append(t)
        get_backend().collate_gather(tables, B)

        # ---- pointer arrays = prefix sums of the gathered per-atom counts
        def ptr_of(name):
            p = torch.zeros(N + 1, dtype=torch.int32, device=dev)
            if N:
                p[1:] = torch.cumsum(out[name][:N], 0, dtype=torch.int32)
            return p

        plan = BatchPlan.__new__(BatchPlan)
        plan.N, plan.E, plan.B = N, E, B
        plan.indptr, plan.indices, plan.rev = ptr_of("deg"), out["indices"][:E], out["rev"][:E]
        plan.max_degree = int(self.max_degree[ids].max()) if B else 0
        take32 = lambda slot: d32[slot[0]:slot[0] + slot[1]]      # noqa: E731
        plan.atom_molptr = take32(molptr32["n1"])
        plan.idx32, plan.mol_ptr, plan.T, plan.inv_ptr, plan.inv_rows = {}, {}, {}, {}, {}
        for lvl in TUPLE_LEVELS:
            Tl, s = int(off[lvl][-1]), LEVEL_ARITY[lvl]
            plan.T[lvl] = Tl
            plan.idx32[lvl] = out[f"idx/{lvl}"][:Tl * s].view(Tl, s)
            plan.mol_ptr[lvl] = take32(molptr32[lvl])
            plan.inv_ptr[lvl] = ptr_of(f"inv_cnt/{lvl}")
            plan.inv_rows[lvl] = out[f"inv_rows/{lvl}"][:Tl * s]
        plan.inc_ptr, plan.inc_code = ptr_of("inc_cnt"), out["inc_code"][:int(inc_rows.sum())]
        plan.device = plan.indptr.device
        plan.n_real_mols = n_real if pad_to is not None else None
        # ---- the graph object
        data: Dict[str, Dict[str, torch.Tensor]] = {nt: {} for nt in NTYPES}
        for (nt, k), table in self.feat.items():
            rows_total = int(off[nt][-1])
            words = out[f"feat/{nt}/{k}"]
            if table.conf:
                v = words[:rows_total * n_out * table.width].view(table.dtype).view((rows_total, n_out) + table.trailing)
            else:
                v = words[:rows_total * table.width].view(table.dtype).view((rows_total,) + table.trailing)
            data[nt][k] = v
        for lvl in TUPLE_LEVELS:
            data[lvl]["idxs"] = plan.idx32[lvl].long()
        if self.has_confs:
            data["g"]["is_dummy"] = take32(dummy32).view(torch.float32).view(B, n_out)
        # (output_size: without it repeat_interleave reads the total back from the device -- the one host sync of a collate)
        dst = (torch.repeat_interleave(torch.arange(N, device=dev), (plan.indptr[1:] - plan.indptr[:-1]).long(), output_size=int(plan.E))
               if N else torch.zeros(0, dtype=torch.long, device=dev))
        g = MolBatch(plan.indices.long(), dst, data, {nt: cnt[nt] for nt in NTYPES})
        g._plan = plan
        return g, tuple(self.names[i] for i in ids[:n_real].tolist())


class ShapeBuckets:
    """A handful of batch shapes (rows per level) that every batch of an epoch is padded to, so that a recorded train step (capture.
    CapturedTrainStep: shapes are part of a hipGraph) serves real epochs: the batches of one epoch are sorted by their atom count and cut into
    `n_buckets` groups; a bucket's caps are its group's largest rows per level plus room for the padding molecule (`min_pad` atoms at least),
    rounded up, the last bucket with `margin` on top for the batches of later epochs.  `choose` returns the smallest bucket a batch fits."""

    def __init__(self, dataset: DeviceDataset, batches: Sequence[np.ndarray], n_buckets: int = 4, min_pad: int = 16, round_to: int = 32, margin: float = 0.06):
        tot = [dataset.totals(b) for b in batches]
        if not tot:
            raise ValueError("ShapeBuckets: no batches to calibrate on")
        order = np.argsort([t["n1"] for t in tot], kind="stable")
        groups = [g for g in np.array_split(order, max(1, min(int(n_buckets), len(tot)))) if len(g)]
        up = lambda v, r: int((int(v) + r - 1) // r * r)      # noqa: E731
        self.caps: List[Dict[str, int]] = []
        for gi, grp in enumerate(groups):
            m = 1.0 + (margin if gi == len(groups) - 1 else 0.0)
            c = {nt: max(tot[i][nt] for i in grp) for nt in DeviceDataset.PAD_DIMS}
            c = {nt: up(c[nt] * m + (min_pad if nt in ("n1", "n2") else 0), round_to) for nt in c}
            while any(DeviceDataset.pad_sizes(tot[i], c) is None for i in grp):      # (a batch with few atoms and many bonds: more padding atoms hold more bonds)
                c["n1"] += round_to
                c["n2"] += round_to
                if c["n1"] > 4 * max(tot[i]["n1"] for i in grp) + 1024:
                    raise ValueError("ShapeBuckets: no caps found that every batch of the group can be padded to")
            if not self.caps or any(c[k] > self.caps[-1][k] for k in c):
                self.caps.append(c)
        self.max_pad = {nt: max(c[nt] for c in self.caps) - min(t[nt] for t in tot) for nt in DeviceDataset.PAD_DIMS}
        self.max_pad = {nt: int(v * 1.25) + 64 for nt, v in self.max_pad.items()}

    def choose(self, totals: Dict[str, int]) -> Optional[Dict[str, int]]:
        for c in self.caps:
            if DeviceDataset.pad_sizes(totals, c) is not None:
                return c
        return None
