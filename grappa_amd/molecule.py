"""Input dataclass `Molecule` (host side of the hot path's entry, G0).

Mirrors the interface of the reference's data/Molecule.py (ctor :61-110, `__post_init__`
:126-156, `add_features` :270-346, `to_dgl` :429-537, `from_dict`/`to_dict` :540-595,
`random` :675-690) but builds a `MolBatch` instead of a DGL heterograph and derives the
ring/degree features without RDKit (featurize.py).  Force-field front-ends
(`from_openmm_system`, `from_openff_molecule`, `from_smiles`) are outside the hot path.
"""
from typing import Dict, List, Optional, Tuple, Union

import numpy as np
import torch

from . import constants, featurize, tuple_indices
from .batch import MolBatch, single_graph


class Molecule:
    def __init__(self, atoms, bonds, impropers, atomic_numbers, partial_charges,
                 additional_features: Optional[Dict[str, np.ndarray]] = None,
                 angles=None, propers=None, improper_in_correct_format: bool = False,
                 ring_encoding: bool = True, degree: bool = True, mass_encoding: bool = True,
                 mapped_smiles: str = None, charge_model: str = "amber99") -> None:
        self.atoms = atoms
        self.bonds = bonds
        self.impropers = impropers
        self.atomic_numbers = atomic_numbers
        self.partial_charges = partial_charges
        self.additional_features = additional_features
        self.angles = angles
        self.propers = propers
        self.neighbor_dict = None
        self.charge_model = charge_model

        if not improper_in_correct_format:
            self.process_impropers()
        self.__post_init__()
        if mass_encoding:
            self.add_features(["mass"])
        if ring_encoding:
            self.add_features(["ring_encoding"])
        if degree:
            self.add_features(["degree"])
        if mapped_smiles is not None:
            raise NotImplementedError("sp_hybridization from mapped SMILES needs openff-toolkit (outside the hot path)")

    def process_impropers(self):
        if self.neighbor_dict is None:
            self.neighbor_dict = tuple_indices.get_neighbor_dict(self.bonds, sort=True)
        _, self.impropers = tuple_indices.get_torsions(self.impropers, self.neighbor_dict,
                                                        central_atom_position=constants.IMPROPER_CENTRAL_IDX)

    def __post_init__(self):
        if self.angles is None or self.propers is None:
            if self.neighbor_dict is None:
                # the usual case (no caller-supplied neighbour lists): native O(atoms) enumeration, reference row order
                d = tuple_indices.get_idx_tuples(self.bonds)
            else:
                d = tuple_indices.get_idx_tuples(self.bonds, self.neighbor_dict, is_sorted=False)
            if self.angles is None:
                self.angles = d["angles"]
            if self.propers is None:
                self.propers = d["propers"]
        if self.additional_features is None:
            self.additional_features = {}
        if self.charge_model not in constants.CHARGE_MODELS:
            raise ValueError(f"charge_model must be one of {constants.CHARGE_MODELS} but is {self.charge_model}")
        if "charge_model" not in self.additional_features:
            onehot = np.array([cm == self.charge_model for cm in constants.CHARGE_MODELS], dtype=np.float32)
            self.additional_features["charge_model"] = np.tile(onehot, (len(self.atoms), 1))
        if "is_radical" not in self.additional_features:
            self.additional_features["is_radical"] = np.zeros((len(self.atoms),), dtype=np.float32)

    def _bonds_by_idx(self):
        idx = {a: i for i, a in enumerate(self.atoms)}
        return np.array([(idx[b[0]], idx[b[1]]) for b in self.bonds], dtype=np.int64).reshape(-1, 2)

    def add_features(self, feat_names: Union[str, List[str]] = ("ring_encoding", "degree", "mass"), **kwargs):
        if isinstance(feat_names, str):
            feat_names = [feat_names]
        for name in feat_names:
            if name == "ring_encoding":
                self.additional_features[name] = featurize.ring_encoding(len(self.atoms), self._bonds_by_idx())
            elif name == "degree":
                self.additional_features[name] = featurize.degree_encoding(len(self.atoms), self._bonds_by_idx())
            elif name == "mass":
                m = np.array([constants.ATOMIC_MASSES[int(z)] for z in self.atomic_numbers], dtype=np.float32)
                self.additional_features[name] = np.stack((m, np.log(m)), axis=1)
            else:
                raise NotImplementedError(f"Feature {name} not implemented yet.")

    def sort(self):
        for i, b in enumerate(self.bonds):
            self.bonds[i] = (b[0], b[1]) if b[0] < b[1] else (b[1], b[0])
        for i, a in enumerate(self.angles):
            self.angles[i] = (a[0], a[1], a[2]) if a[0] < a[2] else (a[2], a[1], a[0])
        for i, p in enumerate(self.propers):
            self.propers[i] = (p[0], p[1], p[2], p[3]) if p[0] < p[3] else (p[3], p[2], p[1], p[0])

    # ---------------------------------------------------------------------------------------
    def to_dgl(self, max_element=constants.MAX_ELEMENT, exclude_feats: List[str] = []) -> MolBatch:
        """-> single-molecule MolBatch with node types g, n1, n2, n3, n4, n4_improper.
        `n1` carries 'ids' (= self.atoms); the tuple levels carry 'idxs' (positions in self.atoms)."""
        assert max_element > 0, f"max_element must be larger than 0 but is {max_element}"
        assert not any(x is None for x in (self.angles, self.propers)), "angles and propers must not be None"
        ids_arr = np.asarray(self.atoms, dtype=np.int64)
        identity = ids_arr.size == 0 or (ids_arr[0] == 0 and ids_arr[-1] == ids_arr.size - 1 and bool(np.all(np.diff(ids_arr) == 1)))
        if not identity:
            order = np.argsort(ids_arr, kind="stable")
            sorted_ids = ids_arr[order]

        def table(rows, s):
            """atom ids -> positions in self.atoms (vectorised: the id list is 0 .. n-1 for most molecules)"""
            t = np.asarray(rows, dtype=np.int64).reshape(-1, s)
            if identity or t.size == 0:
                if t.size and (t.min() < 0 or t.max() >= ids_arr.size):
                    raise KeyError(int(t.max() if t.max() >= ids_arr.size else t.min()))
                return t
            pos = np.searchsorted(sorted_ids, t)
            if np.any(pos >= sorted_ids.size) or np.any(sorted_ids[np.minimum(pos, sorted_ids.size - 1)] != t):
                raise KeyError(int(t[(pos >= sorted_ids.size) | (sorted_ids[np.minimum(pos, sorted_ids.size - 1)] != t)][0]))
            return order[pos]

        idxs = {"n2": table(self.bonds, 2), "n3": table(self.angles, 3), "n4": table(self.propers, 4),
                "n4_improper": table(self.impropers, 4)}
        z = np.asarray(self.atomic_numbers, dtype=np.int64)
        if np.any(z > max_element):
            raise ValueError(f"max_element ({max_element}) must be larger than the largest atomic number ({z.max()})")
        if np.any(z < 1):
            raise ValueError(f"min_element must be larger than 0 but is {z.min()}")
        assert len(z) == len(self.partial_charges) == len(self.atoms)
        n1 = {
            "atomic_number": torch.nn.functional.one_hot(torch.from_numpy(z) - 1, num_classes=max_element).float(),
            "partial_charge": torch.tensor(np.asarray(self.partial_charges, dtype=np.float32)),
        }
        for feat, val in self.additional_features.items():
            if feat in exclude_feats:
                continue
            n1[feat] = torch.tensor(np.asarray(val), dtype=torch.float32)
        return single_graph(len(self.atoms), idxs["n2"], idxs, n1, ids=np.asarray(self.atoms, dtype=np.int64))

    to_graph = to_dgl

    def to_dict(self):
        d = {"atoms": np.array(self.atoms, dtype=np.int64), "bonds": np.array(self.bonds, dtype=np.int64),
             "impropers": np.array(self.impropers, dtype=np.int64), "atomic_numbers": np.array(self.atomic_numbers, dtype=np.int64),
             "partial_charges": np.array(self.partial_charges, dtype=np.float32)}
        if self.angles is not None:
            d["angles"] = np.array(self.angles, dtype=np.int64)
        if self.propers is not None:
            d["propers"] = np.array(self.propers, dtype=np.int64)
        for k, v in self.additional_features.items():
            d[k] = np.asarray(v)
        return d

    @classmethod
    def from_dict(cls, array_dict: Dict):
        core = ["atoms", "bonds", "angles", "propers", "impropers", "atomic_numbers", "partial_charges"]
        add = {k: v for k, v in array_dict.items() if k not in core}
        assert all(f.shape[0] == array_dict["atoms"].shape[0] for f in add.values())
        return cls(atoms=array_dict["atoms"], bonds=array_dict["bonds"], angles=array_dict["angles"],
                   propers=array_dict["propers"], impropers=array_dict["impropers"],
                   atomic_numbers=array_dict["atomic_numbers"], partial_charges=array_dict["partial_charges"],
                   additional_features=add, improper_in_correct_format=True,
                   ring_encoding="ring_encoding" not in add, degree="degree" not in add, mass_encoding="mass" not in add)

    @classmethod
    def random(cls):
        """(A-B-C-D, E-B) toy molecule, reference data/Molecule.py:675-690."""
        return cls(atoms=[0, 1, 2, 3, 4], bonds=[(0, 1), (1, 2), (2, 3), (1, 4)], angles=[(0, 1, 2), (1, 2, 3), (1, 2, 4)],
                   propers=[(0, 1, 2, 3)], impropers=[(0, 2, 1, 4)], atomic_numbers=[1, 2, 3, 4, 5],
                   partial_charges=[0.0, 0.2, 0.3, -0.5, 0.0])

    @classmethod
    def from_graph(cls, atomic_numbers, bonds, partial_charges, charge_model="am1BCC", impropers="planar"):
        """Build from element numbers + bond list (atom ids = 0..n-1).  impropers='planar' applies the
        synthetic-workload rule of tuple_indices.improper_centres_from_bonds."""
        n = len(atomic_numbers)
        bonds = [tuple(int(x) for x in b) for b in bonds]
        if isinstance(impropers, str):
            nd = tuple_indices.get_neighbor_dict(bonds, sort=True)
            impropers = tuple_indices.improper_centres_from_bonds(bonds, nd, atomic_numbers)
        return cls(atoms=list(range(n)), bonds=bonds, impropers=impropers, atomic_numbers=list(atomic_numbers),
                   partial_charges=list(partial_charges), charge_model=charge_model)
