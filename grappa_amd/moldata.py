"""`MolData`: one training record (molecule graph + conformations + reference energies/forces + classical parameters)
read from the reference's `.npz` schema and turned into a `MolBatch`.

Mirrors the reference's data/MolData.py: `to_dict` :200-258 / `from_dict` :262-336 / `load` :346-352 (np.load with
allow_pickle) and `to_dgl` :155-197 (xyz (N,C,3), energy_ref (1,C) centred, gradient_ref (N,C,3), `energy_<ff>` /
`gradient_<ff>` per stored force field) plus data/Parameters.py:458-511 `write_to_dgl` (k_ref / eq_ref; torsion
(|k|, phase in {0, pi, 2pi}) -> signed k, zero-padded / truncated to N_PERIODICITY_* columns).
Only reading and graph construction are in scope; creation from OpenMM / OpenFF / QM data is not.
"""
from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np
import torch

from . import constants
from .batch import MolBatch
from .molecule import Molecule

_PARAM_KEYS = ["bond_k", "bond_eq", "angle_k", "angle_eq", "proper_ks", "proper_phases", "improper_ks", "improper_phases"]
_META_KEYS = ["xyz", "mol_id", "pdb", "mapped_smiles", "smiles", "sequence"]


def _signed_k(ks: np.ndarray, phases: np.ndarray, nan_if_bad: bool) -> np.ndarray:
    ok0 = np.isclose(phases, 0, atol=1e-2) + np.isclose(phases, 2 * np.pi, atol=1e-2)
    okpi = np.isclose(phases, np.pi, atol=1e-2)
    if not np.all(ok0 + okpi + np.isnan(phases)):
        return np.zeros_like(ks) * np.nan if nan_if_bad else np.zeros_like(ks)
    return np.where(ok0, ks, -ks)


def _fit_columns(x: torch.Tensor, n: int) -> torch.Tensor:
    if x.shape[1] < n:
        return torch.cat([x, torch.zeros_like(x[:, :(n - x.shape[1])])], dim=1)
    return x[:, :n]


@dataclass
class MolData:
    molecule: Molecule
    xyz: np.ndarray                 # (n_confs, n_atoms, 3)
    energy: np.ndarray              # (n_confs,)
    gradient: np.ndarray            # (n_confs, n_atoms, 3)
    energy_ref: np.ndarray
    gradient_ref: np.ndarray
    mol_id: str
    params: Dict[str, np.ndarray] = field(default_factory=dict)       # classical parameters (bond_k, ..., improper_phases)
    ff_energy: Dict[str, np.ndarray] = field(default_factory=dict)
    ff_gradient: Dict[str, np.ndarray] = field(default_factory=dict)
    improper_energy_ref: Optional[np.ndarray] = None
    improper_gradient_ref: Optional[np.ndarray] = None
    extras: Dict[str, np.ndarray] = field(default_factory=dict)       # strings (smiles, pdb, ...) and nonbonded_* arrays

    def __post_init__(self):
        self.mol_id = str(self.mol_id)
        assert self.mol_id not in ("", "None") or True
        self.ff_energy.setdefault("qm", self.energy)
        self.ff_gradient.setdefault("qm", self.gradient)
        for k, v in self.ff_energy.items():
            assert v.shape == self.energy.shape, f"Shape of ff_energy {k} does not match energy: {v.shape} vs {self.energy.shape}"

    # ------------------------------------------------------------------ npz schema
    @classmethod
    def from_dict(cls, d) -> "MolData":
        keys = list(d.keys())
        mol_dict = {k: np.asarray(d[k]) for k in keys
                    if k not in _META_KEYS + _PARAM_KEYS and "energy" not in k and "gradient" not in k}
        molecule = Molecule.from_dict(mol_dict)
        params = {k: np.asarray(d[k]) for k in keys if k in _PARAM_KEYS}
        ff_energy = {k.split("_", 1)[1]: np.asarray(d[k]) for k in keys if k.startswith("energy_") and k != "energy_ref"}
        ff_gradient = {k.split("_", 1)[1]: np.asarray(d[k]) for k in keys if k.startswith("gradient_") and k != "gradient_ref"}
        extras = {k: np.asarray(d[k]) for k in keys if k in ("pdb", "mapped_smiles", "smiles", "sequence") or k.startswith("nonbonded_")}
        return cls(molecule=molecule, xyz=np.asarray(d["xyz"]), energy=np.asarray(d["energy"]), gradient=np.asarray(d["gradient"]),
                   energy_ref=np.asarray(d["energy_ref"]), gradient_ref=np.asarray(d["gradient_ref"]), mol_id=str(np.asarray(d["mol_id"])),
                   params=params, ff_energy=ff_energy, ff_gradient=ff_gradient,
                   improper_energy_ref=np.asarray(d["improper_energy_ref"]) if "improper_energy_ref" in keys else None,
                   improper_gradient_ref=np.asarray(d["improper_gradient_ref"]) if "improper_gradient_ref" in keys else None, extras=extras)

    @classmethod
    def load(cls, path: str) -> "MolData":
        return cls.from_dict(np.load(path, allow_pickle=True))

    def to_dict(self) -> Dict[str, np.ndarray]:
        d = {"xyz": self.xyz, "energy": self.energy, "gradient": self.gradient, "energy_ref": self.energy_ref,
             "gradient_ref": self.gradient_ref, "mol_id": np.array(str(self.mol_id))}
        if self.improper_energy_ref is not None:
            d["improper_energy_ref"] = self.improper_energy_ref
        if self.improper_gradient_ref is not None:
            d["improper_gradient_ref"] = self.improper_gradient_ref
        d.update(self.molecule.to_dict())
        d.update(self.params)
        d.update(self.extras)
        for k, v in self.ff_energy.items():
            d[f"energy_{k}"] = v
        for k, v in self.ff_gradient.items():
            d[f"gradient_{k}"] = v
        return d

    def save(self, path: str) -> None:
        np.savez(path, **self.to_dict())

    # ------------------------------------------------------------------ graph
    def to_dgl(self, max_element=constants.MAX_ELEMENT, exclude_feats=[]) -> MolBatch:
        g = self.molecule.to_dgl(max_element=max_element, exclude_feats=exclude_feats)
        gd, n1 = g.nodes["g"].data, g.nodes["n1"].data
        gd["energy_ref"] = torch.tensor(self.energy_ref.reshape(1, -1), dtype=torch.float32)
        gd["energy_ref"] -= gd["energy_ref"].mean(dim=1)
        n1["gradient_ref"] = torch.tensor(self.gradient_ref.transpose(1, 0, 2), dtype=torch.float32)
        if self.improper_energy_ref is not None:
            gd["improper_energy_ref"] = torch.tensor(self.improper_energy_ref.reshape(1, -1), dtype=torch.float32)
            gd["improper_energy_ref"] -= gd["improper_energy_ref"].mean(dim=1)
        if self.improper_gradient_ref is not None:
            n1["improper_gradient_ref"] = torch.tensor(self.improper_gradient_ref.transpose(1, 0, 2), dtype=torch.float32)
        for k, v in self.ff_energy.items():
            gd[f"energy_{k}"] = torch.tensor(v.reshape(1, -1), dtype=torch.float32)
        for k, v in self.ff_gradient.items():
            n1[f"gradient_{k}"] = torch.tensor(v.transpose(1, 0, 2), dtype=torch.float32)
        n1["xyz"] = torch.tensor(self.xyz.transpose(1, 0, 2), dtype=torch.float32)
        self._write_params(g)
        return g

    to_graph = to_dgl

    def _write_params(self, g: MolBatch, suffix: str = "_ref") -> None:
        p = self.params
        T = {lvl: g.num_nodes(lvl) for lvl in constants.TUPLE_LEVELS}
        nan = lambda *shape: np.full(shape, np.nan, dtype=np.float32)      # noqa: E731  (records without classical parameters)
        g.nodes["n2"].data["k" + suffix] = torch.tensor(p.get("bond_k", nan(T["n2"])), dtype=torch.float32)
        g.nodes["n2"].data["eq" + suffix] = torch.tensor(p.get("bond_eq", nan(T["n2"])), dtype=torch.float32)
        g.nodes["n3"].data["k" + suffix] = torch.tensor(p.get("angle_k", nan(T["n3"])), dtype=torch.float32)
        g.nodes["n3"].data["eq" + suffix] = torch.tensor(p.get("angle_eq", nan(T["n3"])), dtype=torch.float32)
        pk = np.asarray(p.get("proper_ks", nan(T["n4"], constants.N_PERIODICITY_PROPER)))
        pp = np.asarray(p.get("proper_phases", nan(T["n4"], constants.N_PERIODICITY_PROPER)))
        assert np.all((pk >= 0) + np.isnan(pk)), "The proper torsion force constants must be positive"
        # (explicit widths: an empty level has no rows to infer a width from)
        g.nodes["n4"].data["k_ref"] = _fit_columns(torch.tensor(_signed_k(pk, pp, True), dtype=torch.float32).reshape(T["n4"], pk.shape[-1] if pk.ndim > 1 else 1),
                                                   constants.N_PERIODICITY_PROPER)
        ik = np.asarray(p.get("improper_ks", nan(T["n4_improper"], constants.N_PERIODICITY_IMPROPER)))
        ip = np.asarray(p.get("improper_phases", nan(T["n4_improper"], constants.N_PERIODICITY_IMPROPER)))
        assert np.all((ik >= 0) + np.isnan(ik)), "The improper torsion force constants must be positive."
        g.nodes["n4_improper"].data["k_ref"] = _fit_columns(
            torch.tensor(_signed_k(ik, ip, False), dtype=torch.float32).reshape(T["n4_improper"], ik.shape[-1] if ik.ndim > 1 else 1),
            constants.N_PERIODICITY_IMPROPER)
