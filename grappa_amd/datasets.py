"""Synthetic workloads on real molecular graphs.

The reference's datasets are GitHub-release downloads (unreachable offline).  The molecule pool
`data/espaloma_pool.npz` holds the graphs (element numbers + bonds + one seeded 3-D embedding) of the
3,481 Espaloma test/validation molecules whose explicit-H SMILES the reference ships as data; everything
else is seeded noise of the shapes and magnitudes the reference's MolData carries
(data/MolData.py:155-197): partial charges N(0,0.3) re-centred, xyz = embedding + N(0,0.08 A) per
conformation, energy_ref N(0,3) kcal/mol, gradient_ref N(0,10) kcal/mol/A.
"""
from __future__ import annotations

import os
from functools import lru_cache
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .batch import MolBatch, batch, set_number_confs
from .molecule import Molecule

_POOL_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "espaloma_pool.npz")


@lru_cache(maxsize=1)
def pool():
    p = np.load(_POOL_PATH)
    return {k: p[k] for k in p.files}


def pool_size() -> int:
    return len(pool()["atom_ptr"]) - 1


def pool_atom_counts() -> np.ndarray:
    return np.diff(pool()["atom_ptr"])


def pool_molecule(i: int):
    P = pool()
    a0, a1 = P["atom_ptr"][i], P["atom_ptr"][i + 1]
    b0, b1 = P["bond_ptr"][i], P["bond_ptr"][i + 1]
    return P["z"][a0:a1].astype(np.int64), P["bonds"][b0:b1].astype(np.int64), P["xyz"][a0:a1].astype(np.float32)


_MOL_CACHE: Dict[int, Molecule] = {}


def molecule_from_pool(i: int, charge_model: str = "am1BCC", seed: int = 0) -> Molecule:
    """graph + features are deterministic per pool index (cached); charges are seeded."""
    key = (i, charge_model, seed)
    if key not in _MOL_CACHE:
        z, bonds, _ = pool_molecule(i)
        rng = np.random.default_rng(seed * 100003 + i)
        q = rng.normal(0, 0.3, size=len(z)).astype(np.float32)
        q -= q.mean()
        _MOL_CACHE[key] = Molecule.from_graph(z.tolist(), bonds.tolist(), q.tolist(), charge_model=charge_model)
    return _MOL_CACHE[key]


def graph_from_pool(i: int, n_confs: int, seed: int = 0, with_refs: bool = True, charge_model: str = "am1BCC") -> MolBatch:
    mol = molecule_from_pool(i, charge_model, seed)
    _, _, xyz0 = pool_molecule(i)
    n = len(xyz0)
    rng = np.random.default_rng(seed * 7919 + i)
    g = mol.to_dgl()
    xyz = xyz0[:, None, :] + rng.normal(0, 0.08, size=(n, n_confs, 3)).astype(np.float32)
    g.nodes["n1"].data["xyz"] = torch.from_numpy(xyz.astype(np.float32))
    if with_refs:
        g.nodes["g"].data["energy_ref"] = torch.from_numpy(rng.normal(0, 3, size=(1, n_confs)).astype(np.float32))
        g.nodes["n1"].data["gradient_ref"] = torch.from_numpy(rng.normal(0, 10, size=(n, n_confs, 3)).astype(np.float32))
    return set_number_confs(g, n_confs)


def build_batch_from_pool(mol_ids: Sequence[int], n_confs: int = 32, seed: int = 0, with_refs: bool = True) -> MolBatch:
    return batch([graph_from_pool(int(i), n_confs, seed, with_refs) for i in mol_ids])


def select_molecules(n: int, seed: int = 0, min_atoms: int = 1, max_atoms: int = 10 ** 9) -> List[int]:
    """n pool indices with min_atoms <= atoms <= max_atoms, sampled with replacement by a seeded rng."""
    counts = pool_atom_counts()
    cand = np.nonzero((counts >= min_atoms) & (counts <= max_atoms))[0]
    rng = np.random.default_rng(seed)
    return rng.choice(cand, size=n, replace=True).tolist()


@lru_cache(maxsize=1)
def _amide_counts() -> np.ndarray:
    """per pool molecule: number of amide carbons (a carbon bonded to a terminal oxygen AND a nitrogen) -- the N-C(=O) motif of
    a peptide backbone, read off the bond graph alone (the pool stores no bond orders)."""
    P = pool()
    z, bonds = P["z"], P["bonds"]
    n_mol = len(P["atom_ptr"]) - 1
    out = np.zeros(n_mol, dtype=np.int64)
    for i in range(n_mol):
        a0, a1 = P["atom_ptr"][i], P["atom_ptr"][i + 1]
        b = bonds[P["bond_ptr"][i]:P["bond_ptr"][i + 1]].astype(np.int64)
        zi = z[a0:a1]
        deg = np.bincount(b.reshape(-1), minlength=a1 - a0)
        has_o = np.zeros(a1 - a0, dtype=bool)
        has_n = np.zeros(a1 - a0, dtype=bool)
        for u, v in ((b[:, 0], b[:, 1]), (b[:, 1], b[:, 0])):
            has_o[u[(zi[v] == 8) & (deg[v] == 1)]] = True
            has_n[u[zi[v] == 7]] = True
        out[i] = int(np.sum((zi == 6) & has_o & has_n))
    return out


def select_dipeptide_like(n: int, seed: int = 0, min_atoms: int = 25, max_atoms: int = 60) -> List[int]:
    """pool molecules with >= 2 N-C(=O) motifs and 25-60 atoms, sampled with replacement by a seeded rng"""
    counts = pool_atom_counts()
    cand = np.nonzero((counts >= min_atoms) & (counts <= max_atoms) & (_amide_counts() >= 2))[0]
    rng = np.random.default_rng(seed)
    return rng.choice(cand, size=n, replace=True).tolist()


WORKLOADS = {
    # name: (batch, min_atoms, max_atoms, n_confs)   -- BASELINE.json configs, SURVEY.md section 8(d)
    "C1-dipeptide-b8": (8, 25, 60, 32),
    "C2-pubchem-b256": (256, 20, 40, 32),
    "C3-espaloma-b1024": (1024, 1, 10 ** 9, 32),
    "C4-espaloma-b4096": (4096, 1, 10 ** 9, 32),
}


WORKLOAD_DESCRIPTIONS = {
    "C1-dipeptide-b8": "8 dipeptide-like molecules (>= 2 amide motifs, 25-60 atoms) of the Espaloma pool",
    "C2-pubchem-b256": "256 small molecules (20-40 atoms) of the Espaloma pool",
    "C3-espaloma-b1024": "1024 molecules drawn from the whole Espaloma pool (mean ~38 atoms, up to 126)",
    "C4-espaloma-b4096": "4096 molecules drawn from the whole Espaloma pool (mean ~38 atoms, up to 126)",
}


def workload_molecule_ids(name: str, seed: int = 0) -> List[int]:
    b, lo, hi, _ = WORKLOADS[name]
    if name.startswith("C1-dipeptide"):
        return select_dipeptide_like(b, seed=seed, min_atoms=lo, max_atoms=hi)
    return select_molecules(b, seed=seed, min_atoms=lo, max_atoms=hi)


def build_workload(name: str, seed: int = 0, mol_ids: Optional[Sequence[int]] = None) -> MolBatch:
    _, _, _, c = WORKLOADS[name]
    ids = workload_molecule_ids(name, seed) if mol_ids is None else list(mol_ids)
    return build_batch_from_pool(ids, n_confs=c, seed=seed)


def protein_like_graph(n_atoms_min: int = 50000, seed: int = 0) -> MolBatch:
    """C5 stand-in: ONE graph of >= n_atoms_min atoms made of disjoint copies of the largest pool molecules
    (inference only: no conformations)."""
    counts = pool_atom_counts()
    big = np.argsort(-counts)[:32]
    graphs, total, j = [], 0, 0
    while total < n_atoms_min:
        i = int(big[j % len(big)])
        graphs.append(molecule_from_pool(i, "amber99", seed).to_dgl())
        total += int(counts[i])
        j += 1
    g = batch(graphs)
    # present it as a single molecule (one row at the 'g' level), as Grappa.predict would see a protein
    one = {nt: np.array([g.num_nodes(nt)]) for nt in g.ntypes}
    one["g"] = np.array([1])
    data = {nt: dict(g.nodes[nt].data) for nt in g.ntypes}
    data["g"] = {}
    return MolBatch(g._src, g._dst, data, one)


_T4_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "t4_lysozyme.npz")


def t4_lysozyme_molecule() -> Molecule:
    """the all-atom T4 lysozyme graph (2,634 atoms, amber99 template charges; data/build_t4.py made it from the structure and
    residue templates the reference ships).  Impropers follow the planar-centre rule of the synthetic workloads."""
    d = np.load(_T4_PATH)
    return Molecule.from_graph(d["z"].astype(np.int64).tolist(), d["bonds"].astype(np.int64).tolist(), d["charges"].astype(np.float32).tolist(),
                               charge_model="amber99")


def protein_graph_t4(copies: int = 19) -> MolBatch:
    """BASELINE.json configs[4] / SURVEY 8(d) C5: ONE graph of `copies` disjoint T4 lysozymes (19 -> 50,046 atoms), inference only"""
    one = t4_lysozyme_molecule().to_dgl()
    g = batch([one] * copies)
    counts = {nt: np.array([g.num_nodes(nt)]) for nt in g.ntypes}
    counts["g"] = np.array([1])
    data = {nt: dict(g.nodes[nt].data) for nt in g.ntypes}
    data["g"] = {}
    return MolBatch(g._src, g._dst, data, counts)
