"""`Dataset`: the list of (molecule graph, subdataset name) items with their molecule ids that the training loop draws from
 -- the reference's data/Dataset.py on `MolBatch` graphs, with an own on-disk format.

Reference surface kept (same names, arguments and results; data/Dataset.py:22-295): `Dataset(graphs, mol_ids, subdataset)`, `len`, `ds[i]
-> (graph, subdataset)`, `split(train_ids, val_ids, test_ids)` (ids in no list go to the test set), `from_moldata`, `+`,
`remove_uncommon_features`, `clean`, `calc_split_ids` / `get_k_fold_split_ids` (utils/torch_utils.py:11-135, :141-345: splits BY
MOLECULE ID, per-subdataset shares, duplicated ids kept together -- the same seeded draws, so a split file written by the reference
and one computed here agree, tests/test_host_dataset.py), `slice`, `where`, `shuffle`, `subsampled`, `save` / `load`.

On disk (`save(dir)`): `mol_ids.json` and `subdataset.json` exactly as the reference writes them, and -- instead of DGL's private
`graphs.bin`, which needs DGL to read -- ONE shard file `graphs.gshard`: a JSON header (tables, dtypes, per-molecule shapes, byte
offsets) followed by every feature of every molecule as raw little-endian arrays, 64-byte aligned.  `load` memory-maps it: a molecule's
tensors are views into the map (nothing is parsed or copied until a batch is built), and `to_device()` packs the whole set into HBM
once (`DeviceDataset`).  `from_npz_dir` reads a directory of the reference's MolData `.npz` records (data/MolData.py:200-352; the
reference's dataset_creation/benchmark_datasets/to_dgl.py:5-50 does the same through DGL).
"""
from __future__ import annotations

import json
import os
import random
from collections import Counter
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from .batch import NTYPES, MolBatch

_MAGIC = b"GRAPPASH"
_VERSION = 1
_ALIGN = 64
SHARD_NAME = "graphs.gshard"


# ------------------------------------------------------------------------------------------------ splits by molecule id
def _partition_of(partition, name) -> Tuple[float, float, float]:
    """(train, val, test) shares of one subdataset: `partition` is one triple for all, or (default triple, {name: triple})"""
    if isinstance(partition[1], dict):
        p = tuple(partition[1].get(name, partition[0]))
    elif isinstance(partition, (tuple, list)):
        p = tuple(partition)
    else:
        raise ValueError(f"Unknown type for partition: {type(partition)}")
    if any(x < 0. for x in p):
        raise ValueError(f"Partition tuple for {name} contains negative values: {p}")
    if len(p) != 3 or abs(sum(p) - 1.) > 1e-10:
        raise ValueError(f"Partition tuple for {name} does not sum to 1.0: {p}")
    return p


def calc_split_ids(ids: Sequence[str], ds_names: Sequence[str], partition, seed: int = 0, duplicate_partition=(0.8, 0.1, 0.1),
                   existing_split: Optional[Dict[str, List[str]]] = None) -> Dict[str, List[str]]:
    """{'train': ids, 'val': ids, 'test': ids}: every subdataset gets approximately its partition; an id that occurs more than once
    (in one or in several subdatasets) lands in ONE set; ids of `existing_split` keep their place and only the others are drawn.
    Seeded like the reference (utils/torch_utils.py:141-345): same ids, same seed -> the same lists in the same order."""
    rng = random.Random(seed)
    if existing_split is not None:
        placed = set(existing_split["train"]) | set(existing_split["val"]) | set(existing_split["test"])
        rest = [(i, n) for i, n in zip(ids, ds_names) if i not in placed]
        if not rest:
            return existing_split
        ids, ds_names = [r[0] for r in rest], [r[1] for r in rest]
    else:
        ids, ds_names = list(ids), list(ds_names)
    occurrences = Counter(ids)
    sets = ("train", "val", "test")
    out: Dict[str, List[str]] = {s: [] for s in sets}

    # ids that occur once, per subdataset (names in sorted order: the order of the draws), and the repeated ids with their subdatasets
    singles: Dict[str, List[str]] = {name: [] for name in sorted(set(ds_names))}
    repeated: Dict[str, List[str]] = {}
    for i, name in zip(ids, ds_names):
        if occurrences[i] == 1:
            singles[name].append(i)
        else:
            repeated.setdefault(i, []).append(name)

    # a repeated id one of whose subdatasets goes wholly into one set has no choice
    free: List[str] = []
    for i, names in repeated.items():
        forced = None
        for name in names:
            p = _partition_of(partition, name)
            if any(abs(x - 1.) < 1e-10 for x in p):
                where = int(np.argmax(p))
                if forced is not None and forced != where:
                    raise ValueError(f"Internal error: Duplicate id {i} has to be in both {forced} and {where}.")
                forced = where
        if forced is None:
            free.append(i)
        else:
            out[sets[forced]].append(i)
    if isinstance(partition[1], dict):
        for name in {n for i in free for n in repeated[i]}:
            if name in partition[1]:
                assert partition[1][name] == duplicate_partition, (f"Partition for {name} ({partition[1][name]}) does not match duplicate partition "
                                                                    f"({duplicate_partition}) although dataset has duplicate ids.")
    rng.shuffle(free)
    n_tr, n_vl = int(len(free) * duplicate_partition[0]), int(len(free) * duplicate_partition[1])
    dup = {"train": free[:n_tr], "val": free[n_tr:n_tr + n_vl], "test": free[n_tr + n_vl:]}
    where_dup = {i: s for s in sets for i in dup[s]}

    # how many RECORDS of every subdataset the repeated ids already put into each set
    taken = {name: Counter() for name in singles}
    for i, name in zip(ids, ds_names):
        if i in where_dup:
            taken[name][where_dup[i]] += 1

    for name, mine in singles.items():
        p = _partition_of(partition, name)
        rng.shuffle(mine)
        total = len(mine) + sum(taken[name].values())
        add_tr = max(int(total * p[0]) - taken[name]["train"], 0)
        add_vl = max(int(total * p[1]) - taken[name]["val"], 0)
        add_te = len(mine) - add_tr - add_vl
        while add_te < 0:                                  # more asked for than there is: give back from train, then from val
            if add_tr > 0:
                add_tr -= 1
            elif add_vl > 0:
                add_vl -= 1
            else:
                raise ValueError("Not enough samples to fill test set")
            add_te += 1
        out["train"] += mine[:add_tr]
        out["val"] += mine[add_tr:add_tr + add_vl]
        out["test"] += mine[add_tr + add_vl:]
    for s in sets:
        out[s] += dup[s]

    assert sum(len(out[s]) for s in sets) == len(occurrences), "Split failed"
    tr_, vl_, te_ = set(out["train"]), set(out["val"]), set(out["test"])
    assert not (tr_ & vl_) and not (tr_ & te_) and not (vl_ & te_), "the sets of a split must not overlap"
    if existing_split is not None:
        for s in sets:
            out[s] += existing_split[s]
    return out


def _fold_blocks(ids: List[str], k: int, num_folds: int):
    """fold i: test = the i-th of k consecutive blocks, val = the block after it (wrapping around), train = the rest"""
    assert k > 2, "k must be larger than 2"
    n = len(ids)
    cut = lambda j: int(j * n / k)      # noqa: E731
    tr, vl, te = [], [], []
    for i in range(num_folds):
        a, b, c = cut(i), cut(i + 1), cut(i + 2)
        te.append(ids[a:b])
        if c < n:
            vl.append(ids[b:c])
            tr.append(ids[:a] + ids[c:])
        else:
            vl.append(ids[:c - n] + ids[b:])
            tr.append(ids[c - n:a])
    return tr, vl, te


def get_k_fold_split_ids(ids: Sequence[str], ds_names: Sequence[str], k: int = 10, seed: int = 0, num_folds: Optional[int] = None) -> List[Dict[str, List[str]]]:
    """k-fold splits by molecule id (reference utils/torch_utils.py:11-135): the test sets of the k folds partition the ids, every
    subdataset is folded on its own; an id that occurs in several subdatasets is folded with ONE of them (a seeded choice)."""
    rng = random.Random(seed)
    num_folds = k if num_folds is None else num_folds
    ids, ds_names = list(ids), list(ds_names)
    occurrences = Counter(ids)
    per_ds: Dict[str, List[str]] = {name: [] for name in sorted(set(ds_names))}
    homes: Dict[str, List[str]] = {}
    for i, name in zip(ids, ds_names):
        if occurrences[i] == 1:
            per_ds[name].append(i)
        else:
            homes.setdefault(i, []).append(name)
    for i, names in homes.items():
        per_ds[rng.choice(names)].append(i)
    out = [{"train": [], "val": [], "test": []} for _ in range(num_folds)]
    for name, mine in per_ds.items():
        rng.shuffle(mine)
        tr, vl, te = _fold_blocks(mine, k, num_folds)
        for f in range(num_folds):
            out[f]["train"] += tr[f]
            out[f]["val"] += vl[f]
            out[f]["test"] += te[f]
    for f in out:
        assert len(f["train"]) + len(f["val"]) + len(f["test"]) == len(occurrences), "Split failed"
    return out


# ------------------------------------------------------------------------------------------------ the shard file
def _write_shard(path: str, graphs: Sequence[MolBatch]) -> None:
    tables = []            # header entries; `parts` = per-molecule numpy arrays
    keys = {}              # (ntype, key) -> list of arrays (None where a molecule lacks the feature)
    n = len(graphs)
    for m, g in enumerate(graphs):
        if g.batch_size != 1:
            raise ValueError("a dataset item is ONE molecule")
        for nt in NTYPES:
            for key, t in g._data[nt].items():
                keys.setdefault((nt, key), [None] * n)[m] = t.detach().cpu().contiguous().numpy()
        keys.setdefault(("__edges__", "src"), [None] * n)[m] = g._src.cpu().numpy()
        keys.setdefault(("__edges__", "dst"), [None] * n)[m] = g._dst.cpu().numpy()
    blobs, offset = [], 0
    for (nt, key), parts in keys.items():
        first = next(p for p in parts if p is not None)
        entry = {"ntype": nt, "key": key, "dtype": first.dtype.str, "shapes": [], "offsets": []}
        for p in parts:
            if p is None:
                entry["shapes"].append(None)
                entry["offsets"].append(-1)
                continue
            if p.dtype != first.dtype:
                raise ValueError(f"feature {nt}/{key}: mixed dtypes {p.dtype} / {first.dtype}")
            offset = (offset + _ALIGN - 1) // _ALIGN * _ALIGN
            entry["shapes"].append(list(p.shape))
            entry["offsets"].append(offset)
            blobs.append((offset, p))
            offset += p.nbytes
        tables.append(entry)
    header = json.dumps({"n": n, "tables": tables, "num_nodes": {nt: [int(g._bnn[nt][0]) for g in graphs] for nt in NTYPES}}).encode()
    head_len = 20 + len(header)
    data0 = (head_len + _ALIGN - 1) // _ALIGN * _ALIGN
    with open(path, "wb") as fh:
        fh.write(_MAGIC)
        fh.write(np.uint32(_VERSION).tobytes())
        fh.write(np.uint64(len(header)).tobytes())
        fh.write(header)
        fh.write(b"\0" * (data0 - head_len))
        pos = 0
        for off, p in blobs:
            fh.write(b"\0" * (off - pos))
            fh.write(np.ascontiguousarray(p).tobytes())
            pos = off + p.nbytes


def _read_shard(path: str):
    """-> (one-molecule graphs whose tensors are views into the memory-mapped file, the map)"""
    with open(path, "rb") as fh:
        head = fh.read(20)
        if head[:8] != _MAGIC:
            raise ValueError(f"{path}: not a grappa shard file")
        if int(np.frombuffer(head[8:12], dtype=np.uint32)[0]) != _VERSION:
            raise ValueError(f"{path}: unknown shard version")
        hlen = int(np.frombuffer(head[12:20], dtype=np.uint64)[0])
        meta = json.loads(fh.read(hlen).decode())
    data0 = (20 + hlen + _ALIGN - 1) // _ALIGN * _ALIGN
    mm = np.memmap(path, dtype=np.uint8, mode="c", offset=data0) if os.path.getsize(path) > data0 else np.zeros(0, dtype=np.uint8)
    n = meta["n"]
    per_mol = [{nt: {} for nt in list(NTYPES) + ["__edges__"]} for _ in range(n)]
    for t in meta["tables"]:
        dt = np.dtype(t["dtype"])
        for m in range(n):
            shape = t["shapes"][m]
            if shape is None:
                continue
            count = int(np.prod(shape)) if shape else 1
            off = t["offsets"][m]
            arr = mm[off:off + count * dt.itemsize].view(dt).reshape(shape)      # a view into the map: no copy
            per_mol[m][t["ntype"]][t["key"]] = torch.from_numpy(arr)
    graphs = []
    for m in range(n):
        d = per_mol[m]
        graphs.append(MolBatch(d["__edges__"]["src"], d["__edges__"]["dst"], {nt: d[nt] for nt in NTYPES},
                               {nt: np.array([meta["num_nodes"][nt][m]]) for nt in NTYPES}))
    return graphs, mm


# ------------------------------------------------------------------------------------------------ the dataset
class Dataset(torch.utils.data.Dataset):
    """graphs: one-molecule `MolBatch` each; mol_ids: the molecule's id (splits never separate two items of one id); subdataset: the
    name of the data source every item came from (a list, or one name for all)."""

    def __init__(self, graphs: Optional[List[MolBatch]] = None, mol_ids: Optional[List[str]] = None, subdataset: Union[List[str], str, None] = None):
        graphs = [] if graphs is None else graphs
        mol_ids = [] if mol_ids is None else mol_ids
        subdataset = [] if subdataset is None else subdataset
        if isinstance(subdataset, str):
            subdataset = [subdataset] * len(graphs)
        self.graphs, self.mol_ids, self.subdataset = graphs, mol_ids, subdataset
        assert len(graphs) == len(mol_ids) == len(subdataset)

    def __len__(self):
        return len(self.graphs)

    def __getitem__(self, idx):
        return self.graphs[idx], self.subdataset[idx]

    def _take(self, idx: Sequence[int]) -> "Dataset":
        return Dataset([self.graphs[i] for i in idx], [self.mol_ids[i] for i in idx], [self.subdataset[i] for i in idx])

    def __add__(self, other: "Dataset") -> "Dataset":
        return Dataset(self.graphs + other.graphs, self.mol_ids + other.mol_ids, self.subdataset + other.subdataset)

    # ---- splits
    def split(self, train_ids: List[str], val_ids: List[str], test_ids: List[str], check_overlap: bool = True):
        """-> (train, val, test) datasets by membership of the items' molecule ids; an id in none of the lists goes to the test set"""
        tr, vl, te = set(train_ids), set(val_ids), set(test_ids)
        if check_overlap:
            assert not (tr & vl) and not (tr & te) and not (vl & te)
        idx_tr = [i for i, m in enumerate(self.mol_ids) if m in tr]
        idx_vl = [i for i, m in enumerate(self.mol_ids) if m in vl]
        used = set(idx_tr) | set(idx_vl)
        idx_te = [i for i in range(len(self.mol_ids)) if i not in used]
        return self._take(idx_tr), self._take(idx_vl), self._take(idx_te)

    def calc_split_ids(self, partition, seed: int = 0, existing_split: Optional[Dict[str, List[str]]] = None):
        return calc_split_ids(self.mol_ids, self.subdataset, partition, seed=seed, existing_split=existing_split)

    def get_k_fold_split_ids(self, k: int, seed: int = 0, num_folds: Optional[int] = None):
        return get_k_fold_split_ids(self.mol_ids, self.subdataset, k=k, seed=seed, num_folds=num_folds)

    # ---- selections
    def slice(self, start, stop) -> "Dataset":
        return Dataset(self.graphs[start:stop], self.mol_ids[start:stop], self.subdataset[start:stop])

    def where(self, condition: Sequence[bool]) -> "Dataset":
        return self._take([i for i in range(len(self.graphs)) if condition[i]])

    def shuffle(self, seed: int = 0) -> "Dataset":
        """in place (and returned); the permutation of the reference's `np.random.seed(seed); np.random.permutation(n)`"""
        perm = np.random.RandomState(seed).permutation(len(self.graphs))
        self.graphs[:] = [self.graphs[i] for i in perm]
        self.mol_ids[:] = [self.mol_ids[i] for i in perm]
        self.subdataset[:] = [self.subdataset[i] for i in perm]
        return self

    def subsampled(self, factor: float, seed: int = 0) -> "Dataset":
        n = len(self.graphs)
        keep = set(np.random.RandomState(seed).permutation(n)[:int(n * factor)].tolist())
        return self.where([i in keep for i in range(n)])

    # ---- feature hygiene in front of batching (every molecule of a batch must carry the same features)
    def remove_uncommon_features(self, create_feats: Dict[str, Union[float, torch.Tensor]] = {"is_radical": 0.}) -> None:
        if not self.graphs:
            return
        for g in self.graphs:
            n1 = g.nodes["n1"].data
            for k, v in create_feats.items():
                if k not in n1:
                    n = g.num_nodes("n1")
                    n1[k] = torch.repeat_interleave(v, n, dim=0) if isinstance(v, torch.Tensor) else torch.ones(n) * v
        for nt in NTYPES:
            common = set(self.graphs[0].nodes[nt].data.keys())
            for g in self.graphs:
                common &= set(g.nodes[nt].data.keys())
            removed = set()
            for g in self.graphs:
                for k in set(g.nodes[nt].data.keys()) - common:
                    removed.add(k)
                    del g.nodes[nt].data[k]
            if removed:
                print(f"Removed features:\n  {removed}")

    def clean(self, keep_feats: Optional[List[str]] = None) -> None:
        """drop every atom feature that is not named (plus the ones every run needs): less to keep in HBM"""
        if keep_feats is None or not self.graphs:
            return
        keep = set(keep_feats) | {"xyz", "atomic_number", "partial_charge", "ring_encoding"}
        for g in self.graphs:
            for k in [k for k in g.nodes["n1"].data.keys() if k not in keep]:
                del g.nodes["n1"].data[k]

    # ---- construction
    @classmethod
    def from_moldata(cls, moldata_list, subdataset: Union[List[str], str]) -> "Dataset":
        return cls([md.to_dgl() for md in moldata_list], [md.mol_id for md in moldata_list], subdataset)

    @classmethod
    def from_npz_dir(cls, path: Union[str, Path], subdataset: Optional[str] = None) -> "Dataset":
        """every MolData `.npz` record of a directory (sorted by file name); subdataset name = the directory's name unless given"""
        from .moldata import MolData
        path = Path(path)
        files = sorted(p for p in path.iterdir() if p.suffix == ".npz")
        if not files:
            raise FileNotFoundError(f"no .npz records in {path}")
        return cls.from_moldata([MolData.load(str(f)) for f in files], subdataset or path.name)

    # ---- disk
    def save(self, path: Union[str, Path]) -> None:
        path = Path(path)
        path.mkdir(parents=True, exist_ok=True)
        _write_shard(str(path / SHARD_NAME), self.graphs)
        with open(path / "mol_ids.json", "w") as f:
            json.dump(self.mol_ids, f)
        with open(path / "subdataset.json", "w") as f:
            json.dump(self.subdataset, f)

    @classmethod
    def load(cls, path: Union[str, Path]) -> "Dataset":
        path = Path(path)
        if not (path / SHARD_NAME).exists() and (path / "graphs.bin").exists():
            raise FileNotFoundError(f"{path} holds a DGL `graphs.bin`, which only DGL can read; convert the MolData records instead "
                                    f"(Dataset.from_npz_dir(...).save(...))")
        graphs, shard_map = _read_shard(str(path / SHARD_NAME))
        with open(path / "mol_ids.json") as f:
            mol_ids = json.load(f)
        with open(path / "subdataset.json") as f:
            subdataset = json.load(f)
        ds = cls(graphs, mol_ids, subdataset)
        ds.shard_map = shard_map          # the mapped file the graphs' tensors point into
        return ds

    def to_device(self, device="cuda"):
        """the whole dataset resident in HBM, batches assembled there (`DeviceDataset`)"""
        from .device_dataset import DeviceDataset
        return DeviceDataset([(g, s) for g, s in zip(self.graphs, self.subdataset)], device=device)
