"""Enumeration of angles / proper torsions / improper torsions from a bond list.

Host-side mirror of the reference's utils/tuple_indices.py (get_idx_tuples :7-63,
get_neighbor_dict :66-83, is_improper :87-127, get_torsions :144-217): same sets AND the same
row order, because the row order of the tuple tables is the row order of the parameters that
`Grappa.predict` returns.  Written independently on adjacency lists; only the iteration order
(atoms in order of first appearance in the bond list, neighbours ascending) is kept.
"""
from typing import Dict, List, Sequence, Tuple

import numpy as np

from .constants import IMPROPER_CENTRAL_IDX


def get_neighbor_dict(bonds: Sequence[Tuple[int, int]], sort: bool = True) -> Dict[int, List[int]]:
    nb: Dict[int, List[int]] = {}
    for bond in bonds:
        a, b = int(bond[0]), int(bond[1])
        if a == b:
            raise AssertionError(f"Encountered self-bond: {bond}")
        nb.setdefault(a, []).append(b)
        nb.setdefault(b, []).append(a)
    if sort:
        for k in nb:
            nb[k].sort()
    return nb


def get_idx_tuples(bonds, neighbor_dict=None, is_sorted: bool = False):
    """-> {'bonds','angles','propers'} with angle[0] < angle[2] and proper[0] < proper[3].
    Without caller-supplied neighbour lists the enumeration runs natively (libgrappa_host.so, include/grappa_host.h
    grappa_topo_enumerate: O(atoms), same sets and row order); with them the dict's own iteration order is honoured here."""
    if neighbor_dict is None:
        from . import _hostlib
        angles, propers = _hostlib.enumerate_tuples(bonds)
        if not is_sorted:
            bonds = [tuple(int(x) for x in sorted(b)) for b in bonds]
        return {"bonds": bonds, "angles": [tuple(r) for r in angles.tolist()], "propers": [tuple(r) for r in propers.tolist()]}
    return get_idx_tuples_py(bonds, neighbor_dict, is_sorted)


def get_idx_tuples_py(bonds, neighbor_dict=None, is_sorted: bool = False):
    """the same enumeration on Python dicts (caller-supplied neighbour lists; cross-check of the native path in the tests)"""
    if neighbor_dict is None:
        neighbor_dict = get_neighbor_dict(bonds, sort=True)
    elif not is_sorted:
        for k in neighbor_dict:
            neighbor_dict[k] = sorted(neighbor_dict[k])
    angles, propers = [], []
    for a1, nb1 in neighbor_dict.items():
        for a2 in nb1:
            for a3 in neighbor_dict[a2]:
                if a3 == a1:
                    continue
                if a1 < a3:
                    angles.append((a1, a2, a3))
                for a4 in neighbor_dict[a3]:
                    if a4 >= a1:       # neighbour lists ascend: nothing smaller follows
                        break
                    if a4 == a2:
                        continue
                    propers.append((a4, a3, a2, a1))
    if not is_sorted:
        bonds = [tuple(int(x) for x in sorted(b)) for b in bonds]
    return {"bonds": bonds, "angles": angles, "propers": propers}


def is_improper(ids, neighbor_dict, central_atom_position=None):
    ids = tuple(int(i) for i in ids)
    if central_atom_position is not None:
        c = ids[central_atom_position]
        nbs = neighbor_dict[c]
        for i, a in enumerate(ids):
            if i != central_atom_position and a not in nbs:
                return False, None
        return True, central_atom_position
    for pos in (2, 1, 0, 3):
        c = ids[pos]
        nbs = neighbor_dict[c]
        if all(a in nbs for a in ids if a != c):
            return True, ids.index(c)
    return False, None


def is_proper(ids, neighbor_dict):
    return (ids[0] in neighbor_dict[ids[1]]) and (ids[1] in neighbor_dict[ids[2]]) and (ids[2] in neighbor_dict[ids[3]])


def get_torsions(torsion_ids, neighbor_dict, central_atom_position=IMPROPER_CENTRAL_IDX):
    """-> (propers, impropers); every improper centre is expanded to its three cyclic orderings
    of the outer atoms with the central atom at `central_atom_position`."""
    propers, impropers = [], []
    seen_improper, seen_proper = set(), set()
    for torsion in torsion_ids:
        torsion = tuple(int(t) for t in torsion)
        key = tuple(sorted(torsion))
        if key in seen_improper or key in seen_proper:
            continue
        imp, cidx = is_improper(torsion, neighbor_dict)
        prop = is_proper(torsion, neighbor_dict)
        if imp and prop:
            imp = False
        if not imp and not prop:
            raise RuntimeError(f"Encountered torsion that is neither proper nor improper: {torsion}")
        if not imp:
            propers.append(torsion)
            seen_proper.add(key)
        else:
            central = torsion[cidx]
            others = [torsion[i] for i in range(4) if i != cidx]
            orderings = [others, [others[i] for i in (1, 2, 0)], [others[i] for i in (2, 0, 1)]]
            for o in orderings:
                row, j = [], 0
                for pos in range(4):
                    if pos == central_atom_position:
                        row.append(central)
                    else:
                        row.append(o[j])
                        j += 1
                impropers.append(tuple(row))
            seen_improper.add(key)
    return propers, impropers


def improper_centres_from_bonds(bonds, neighbor_dict=None, atomic_numbers=None):
    """Synthetic-workload helper (no reference counterpart: the reference takes impropers from
    OpenMM / OpenFF force fields): one improper (sorted outer atoms + centre) for every atom with
    exactly three neighbours that is C or N -- the planar-centre pattern amber/smirnoff use."""
    if neighbor_dict is None:
        neighbor_dict = get_neighbor_dict(bonds, sort=True)
    out = []

    def sp2_carbon(x):
        return len(neighbor_dict[x]) == 3 and (atomic_numbers is None or int(atomic_numbers[x]) == 6)

    for a, nbs in neighbor_dict.items():
        if len(nbs) != 3:
            continue
        if atomic_numbers is not None:
            z = int(atomic_numbers[a])
            if z not in (6, 7):
                continue
            if z == 7 and not any(sp2_carbon(b) for b in nbs):   # amine nitrogens stay pyramidal
                continue
        o = sorted(nbs)
        out.append((o[0], o[1], a, o[2]))
    return out
