"""`Parameters`: the output dataclass of `Grappa.predict` (reference data/Parameters.py:18-140).
Only construction from a parametrised graph is on the hot path; the OpenMM/plotting methods of the
reference are out of scope."""
from dataclasses import dataclass
from typing import Optional

import numpy as np


@dataclass
class Parameters:
    atoms: np.ndarray
    bonds: np.ndarray
    bond_k: np.ndarray
    bond_eq: np.ndarray
    angles: np.ndarray
    angle_k: np.ndarray
    angle_eq: np.ndarray
    propers: np.ndarray
    proper_ks: np.ndarray
    proper_phases: np.ndarray
    impropers: Optional[np.ndarray]
    improper_ks: Optional[np.ndarray]
    improper_phases: Optional[np.ndarray]

    @classmethod
    def from_dgl(cls, g, suffix: str = "", check_eq_values: bool = True):
        """g: parametrised (single-molecule) graph with 'ids' at n1 and 'idxs' at the tuple levels.
        Signed torsion constants become (|k|, phase in {0, pi}); note the reference's asymmetry: propers use
        k >= 0 -> phase 0, impropers use k > 0 -> phase 0 (Parameters.py:105-109 vs :117-121)."""
        def arr(nt, key):
            return g.nodes[nt].data[key].detach().cpu().numpy()

        atom_ids = arr("n1", "ids")
        bonds = atom_ids[arr("n2", "idxs")]
        bond_k, bond_eq = arr("n2", f"k{suffix}"), arr("n2", f"eq{suffix}")
        angles = atom_ids[arr("n3", "idxs")]
        angle_k, angle_eq = arr("n3", f"k{suffix}"), arr("n3", f"eq{suffix}")
        if check_eq_values:
            MAX_ANGLE, MAX_BOND_LENGTH = 45, 0.5
            if np.any(angle_eq < np.pi / 180 * MAX_ANGLE):
                n_smaller = int(np.sum(angle_eq < np.pi / 180 * MAX_ANGLE))
                raise RuntimeError(f"{n_smaller} angles are smaller than 20 degrees. This can lead to numerical instabilities in the model.\n"
                                   f"The smallest angle is {np.min(angle_eq) * 180 / np.pi} degrees at atom ids {angles[np.argmin(angle_eq)]}.")
            if np.any(bond_eq < MAX_BOND_LENGTH):
                n_smaller = int(np.sum(bond_eq < MAX_BOND_LENGTH))
                raise RuntimeError(f"{n_smaller} bond eq lengths are smaller than 0.5 Angstrom. This can lead to numerical instabilities in the model.\n"
                                   f"The smallest bond eq length is {np.min(bond_eq)} Angstrom at atom ids {bonds[np.argmin(bond_eq)]}.")
        proper_ks = arr("n4", f"k{suffix}")
        proper_phases = np.where(proper_ks >= 0., np.zeros_like(proper_ks), np.zeros_like(proper_ks) + np.pi)
        proper_ks = np.abs(proper_ks)
        propers = atom_ids[arr("n4", "idxs")]
        improper_ks = arr("n4_improper", f"k{suffix}")
        improper_phases = np.where(improper_ks > 0, np.zeros_like(improper_ks), np.zeros_like(improper_ks) + np.pi)
        improper_ks = np.abs(improper_ks)
        impropers = atom_ids[arr("n4_improper", "idxs")]
        return cls(atoms=atom_ids, bonds=bonds, bond_k=bond_k, bond_eq=bond_eq, angles=angles, angle_k=angle_k, angle_eq=angle_eq,
                   propers=propers, proper_ks=proper_ks, proper_phases=proper_phases, impropers=impropers, improper_ks=improper_ks,
                   improper_phases=improper_phases)
