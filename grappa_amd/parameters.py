"""`Parameters`: what `Grappa.predict` hands back (the reference's data/Parameters.py:18-140 defines the record).
Only the construction from a parametrised graph is on the hot path; the OpenMM / plotting methods of the
reference are out of scope.

Contract of `from_dgl` (pinned by tests/golden/ref_predict.npz, produced by the reference's own `Grappa.predict`):
  * index tables are returned in ATOM-ID space (`ids` of n1 applied to the `idxs` of every tuple level);
  * bonds / angles carry k and eq as predicted;
  * torsions carry the magnitude |k| and a phase in {0, pi} that encodes the sign; a vanishing k is given
    phase 0 for propers and phase pi for impropers (`>=` vs `>` in the reference);
  * implausible equilibrium values (an angle below 45 degrees, a bond below 0.5 Angstrom) raise RuntimeError.
"""
from dataclasses import dataclass
from typing import Optional

import numpy as np

MIN_ANGLE_EQ_DEG = 45.0
MIN_BOND_EQ_ANGSTROM = 0.5


def _sign_to_phase(k: np.ndarray, zero_is_positive: bool):
    """signed Fourier coefficient -> (magnitude, phase): a negative coefficient is a cosine shifted by pi."""
    positive = (k >= 0.0) if zero_is_positive else (k > 0.0)
    phase = np.full_like(k, np.pi)
    phase[positive] = 0.0
    return np.abs(k), phase


def _too_small(kind: str, unit: str, values: np.ndarray, limit: float, ids: np.ndarray, scale: float = 1.0) -> Optional[str]:
    bad = np.flatnonzero(values < limit)
    if bad.size == 0:
        return None
    worst = int(np.argmin(values))
    return (f"{bad.size} predicted {kind} equilibrium value(s) lie below {limit * scale:g} {unit}; the smallest is "
            f"{float(values[worst]) * scale:.4g} {unit} (atom ids {ids[worst].tolist()}). Values like this make MD unstable: "
            f"the input (charges, bonds, element types) is probably outside of what the model was trained on.")


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
        """g: parametrised single-molecule graph (`ids` at n1, `idxs` + `k`/`eq` at the tuple levels)."""
        def table(level, key):
            return g.nodes[level].data[key].detach().cpu().numpy()

        ids = table("n1", "ids")

        def atom_ids_of(level):
            idxs = table(level, "idxs")
            # an empty level comes back one-dimensional, as from the reference (its graphs store `torch.tensor([])` there)
            return ids[idxs] if idxs.shape[0] else ids[:0]

        fields = {"atoms": ids}
        for level, name in (("n2", "bond"), ("n3", "angle")):
            fields[name + "s"] = atom_ids_of(level)
            fields[name + "_k"] = table(level, "k" + suffix)
            fields[name + "_eq"] = table(level, "eq" + suffix)
        if check_eq_values:
            problems = [_too_small("angle", "degrees", fields["angle_eq"], np.deg2rad(MIN_ANGLE_EQ_DEG), fields["angles"], 180.0 / np.pi),
                        _too_small("bond", "Angstrom", fields["bond_eq"], MIN_BOND_EQ_ANGSTROM, fields["bonds"])]
            problems = [p for p in problems if p]
            if problems:
                raise RuntimeError(" ".join(problems))
        for level, name, zero_is_positive in (("n4", "proper", True), ("n4_improper", "improper", False)):
            fields[name + "s"] = atom_ids_of(level)
            fields[name + "_ks"], fields[name + "_phases"] = _sign_to_phase(table(level, "k" + suffix), zero_is_positive)
        return cls(**fields)
