"""Collate / loader in front of the hot path.

Mirrors the reference's data/GraphDataLoader.py (`get_collate_fn` :12-75, `GraphDataLoader` :79-148) on `MolBatch`:
dummy conformations are dropped, the number of conformations is unified by `conf_strategy` (int | 'min' | 'max' | 'all' |
'mean') through `set_number_confs` (random sub-sampling / padding with flagged dummy copies), and the molecules are
concatenated with shifted tuple indices.  Differences: no `copy.deepcopy` per graph (the reference needs it because DGL
batching mutates `idxs` in place and autograd objects are shared; `batch()` here never mutates its inputs), and
`GraphDataLoader.to(device)` uploads with pinned, non-blocking copies.
"""
from collections import defaultdict
from typing import Dict, List, Sequence, Tuple, Union

import numpy as np
import torch
from torch.utils.data import DataLoader

from .batch import MolBatch, batch, delete_dummy_confs, set_number_confs


def _shallow_copy(g: MolBatch) -> MolBatch:
    """new container, same tensors (set_number_confs / delete_dummy_confs only rebind dict entries)"""
    return MolBatch(g._src, g._dst, {nt: dict(d) for nt, d in g._data.items()}, g._bnn)


def get_collate_fn(conf_strategy: Union[str, int] = "min", deep_copies_of_same_graphs: bool = False):
    def collate_fn(items: List[Tuple[MolBatch, str]]):
        assert isinstance(items, list), f"batch must be a list, but got {type(items)}"
        assert isinstance(items[0], tuple), f"batch must be a list of tuples, but got {type(items[0])}"
        assert isinstance(items[0][0], MolBatch), f"batch must be a list of tuples where the first element is a MolBatch, but got {type(items[0][0])}"
        graphs, names = zip(*items)
        graphs = [delete_dummy_confs(_shallow_copy(g)) for g in graphs]
        counts = [g.nodes["n1"].data["xyz"].shape[1] for g in graphs]
        if isinstance(conf_strategy, int):
            n_confs = min(conf_strategy, max(counts))
        elif conf_strategy == "min":
            n_confs = min(counts)
        elif conf_strategy in ("max", "all"):
            n_confs = max(counts)
        elif conf_strategy == "mean":
            n_confs = int(np.mean(counts))
        else:
            raise ValueError(f"Unknown conf_strategy: {conf_strategy}")
        graphs = [set_number_confs(g, n_confs) for g in graphs]
        return batch(graphs), names

    return collate_fn


class GraphDataLoader(DataLoader):
    """dataset: any sequence of (MolBatch, subdataset_name).  Sampling weights / balancing as in the reference (:100-138)."""

    def __init__(self, dataset, *args, shuffle=False, weights: Dict[str, float] = {}, conf_strategy: Union[str, int] = "mean",
                 balance_factor: float = 0., **kwargs):
        assert isinstance(weights, dict), f"weights must be a dict, but got {type(weights)}"
        assert isinstance(conf_strategy, (str, int)), f"conf_strategy must be a str or int, but got {type(conf_strategy)}"
        assert 0 <= balance_factor <= 1, f"balance_factor must be between 0 and 1, but got {balance_factor}"
        if shuffle and (len(weights) or balance_factor > 0):
            names = [n for _, n in dataset]
            sample_weights = np.array([weights.get(n, 1.0) for n in names], dtype=np.float64)
            if balance_factor > 0:
                occ = {n: names.count(n) / len(names) for n in set(names)}
                balanced = 1.0 / float(len(occ))
                ratio = {n: float((1.0 - balance_factor) * balanced + balance_factor * occ[n]) for n in occ}
                sample_weights = sample_weights * np.array([1.0 / ratio[n] for n in names])
            sampler = torch.utils.data.WeightedRandomSampler(sample_weights.tolist(), len(sample_weights), replacement=True)
            super().__init__(dataset, *args, collate_fn=get_collate_fn(conf_strategy, True), sampler=sampler, **kwargs)
        elif len(weights) > 0:
            raise ValueError("Weights are only supported with shuffle=True")
        else:
            super().__init__(dataset, *args, collate_fn=get_collate_fn(conf_strategy), shuffle=shuffle, **kwargs)

    def to(self, device):
        for g, names in self:
            yield g.to(device), names
