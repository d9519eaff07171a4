"""Minimal parser for fully bracketed, explicit-hydrogen SMILES (the form of the Espaloma
split files the reference ships under dataset_creation/get_espaloma_split/*.json).  Returns
the molecular graph only (atomic numbers, bonds); stereo marks, charges and bond orders are
read past.  Used to build the synthetic benchmark/golden molecule pool."""
import re
from typing import List, Tuple

_SYMBOLS = {"H": 1, "B": 5, "C": 6, "N": 7, "O": 8, "F": 9, "Na": 11, "Mg": 12, "Si": 14, "P": 15, "S": 16,
            "Cl": 17, "K": 19, "Ca": 20, "Br": 35, "I": 53, "Li": 3, "Se": 34}
_ATOM_RE = re.compile(r"\[(\d*)([A-Za-z][a-z]?)([^\]]*)\]")


def parse_bracket_smiles(smi: str) -> Tuple[List[int], List[Tuple[int, int]]]:
    z: List[int] = []
    bonds: List[Tuple[int, int]] = []
    prev = -1
    branch: List[int] = []
    ring = {}
    i = 0
    n = len(smi)
    while i < n:
        c = smi[i]
        if c == "[":
            m = _ATOM_RE.match(smi, i)
            if m is None:
                raise ValueError(f"cannot parse atom at {i} in {smi}")
            sym = m.group(2)
            sym = sym[0].upper() + sym[1:]
            if sym not in _SYMBOLS:
                if sym[0] in _SYMBOLS:
                    sym = sym[0]
                else:
                    raise ValueError(f"unknown element {sym} in {smi}")
            z.append(_SYMBOLS[sym])
            cur = len(z) - 1
            if prev >= 0:
                bonds.append((min(prev, cur), max(prev, cur)))
            prev = cur
            i = m.end()
        elif c == "(":
            branch.append(prev)
            i += 1
        elif c == ")":
            prev = branch.pop()
            i += 1
        elif c in "-=#:/\\.":
            if c == ".":
                prev = -1
            i += 1
        elif c == "%":
            lab = smi[i + 1:i + 3]
            i += 3
            if lab in ring:
                o = ring.pop(lab)
                bonds.append((min(o, prev), max(o, prev)))
            else:
                ring[lab] = prev
        elif c.isdigit():
            if c in ring:
                o = ring.pop(c)
                bonds.append((min(o, prev), max(o, prev)))
            else:
                ring[c] = prev
            i += 1
        else:
            raise ValueError(f"unexpected character {c!r} at {i} in {smi}")
    if ring:
        raise ValueError(f"unclosed ring bond in {smi}")
    return z, bonds
