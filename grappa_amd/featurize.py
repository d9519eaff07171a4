"""Graph-only atom features without RDKit.

`ring_encoding` (7): [in any ring, in ring of size 3,4,5,6,7,8]; `degree` (6): one-hot of the
number of bonded neighbours 1..6.  Definitions follow the reference's utils/rdkit_utils.py:7-24
(get_ring_encoding: atom.IsInRing(), atom.IsInRingSize(3..8)) and :55-67 (get_degree) on a graph
built from bonds only (rdkit_graph_from_bonds :27-52: no bond orders, no aromaticity).
RDKit answers IsInRingSize from its (symmetrised) SSSR; here the ring set is the set of
*relevant cycles* (Vismara 1997: cycles that are not a GF(2) sum of strictly shorter cycles),
which coincides with the symmetrised SSSR for the ring systems of organic molecules.
"""
from typing import List, Sequence, Tuple

import numpy as np

MAX_RING = 8


def _adjacency(n_atoms: int, bonds: np.ndarray) -> List[List[int]]:
    adj: List[List[int]] = [[] for _ in range(n_atoms)]
    for a, b in bonds:
        adj[int(a)].append(int(b))
        adj[int(b)].append(int(a))
    return adj


def _ring_atoms(n_atoms: int, adj: List[List[int]]) -> np.ndarray:
    """atoms with at least one incident non-bridge edge (iterative lowlink DFS)."""
    disc = [-1] * n_atoms
    low = [0] * n_atoms
    in_ring = np.zeros(n_atoms, dtype=bool)
    timer = 0
    for root in range(n_atoms):
        if disc[root] != -1:
            continue
        stack = [(root, -1, 0)]
        disc[root] = low[root] = timer
        timer += 1
        while stack:
            v, parent, i = stack.pop()
            if i < len(adj[v]):
                stack.append((v, parent, i + 1))
                w = adj[v][i]
                if w == parent:
                    continue
                if disc[w] == -1:
                    disc[w] = low[w] = timer
                    timer += 1
                    stack.append((w, v, 0))
                else:
                    low[v] = min(low[v], disc[w])
            else:
                if parent != -1:
                    low[parent] = min(low[parent], low[v])
                    if low[v] <= disc[parent]:      # edge (parent, v) is not a bridge
                        in_ring[v] = True
                        in_ring[parent] = True
    return in_ring


def _small_cycles(n_atoms: int, adj: List[List[int]], in_ring: np.ndarray):
    """all simple cycles of length 3..MAX_RING, each once, as (length, edge-bitset, atoms)."""
    edge_id = {}
    for a in range(n_atoms):
        for b in adj[a]:
            if a < b:
                edge_id[(a, b)] = len(edge_id)
    cycles = {}
    for start in range(n_atoms):
        if not in_ring[start]:
            continue
        # paths start -> ... using only atoms > start (so that every cycle is found from its smallest atom)
        stack = [(start, [start])]
        while stack:
            v, path = stack.pop()
            for w in adj[v]:
                if w == start and len(path) >= 3:
                    if path[1] < path[-1]:          # fix the orientation
                        bits = 0
                        for i in range(len(path)):
                            a, b = path[i], path[(i + 1) % len(path)]
                            bits |= 1 << edge_id[(a, b) if a < b else (b, a)]
                        cycles[bits] = tuple(path)
                elif w > start and in_ring[w] and w not in path and len(path) < MAX_RING:
                    stack.append((w, path + [w]))
    return sorted(((len(p), bits, p) for bits, p in cycles.items()), key=lambda c: c[0])


def _relevant(cycles):
    """keep cycles that are not a GF(2) combination of strictly shorter cycles."""
    basis = {}   # pivot bit -> vector (fully reduced incrementally)

    def reduce(v):
        while v:
            p = v.bit_length() - 1
            if p not in basis:
                return v
            v ^= basis[p]
        return 0

    out = []
    i = 0
    while i < len(cycles):
        j = i
        while j < len(cycles) and cycles[j][0] == cycles[i][0]:
            j += 1
        group = cycles[i:j]
        keep = [c for c in group if reduce(c[1]) != 0]
        out.extend(keep)
        for c in keep:                 # extend the basis only after the whole length class is tested
            r = reduce(c[1])
            if r:
                basis[r.bit_length() - 1] = r
        i = j
    return out


def ring_encoding(n_atoms: int, bonds: Sequence[Tuple[int, int]]) -> np.ndarray:
    """native (libgrappa_host.so grappa_ring_encoding: per ring system, O(atoms)); `ring_encoding_py` is the same definition in Python"""
    from . import _hostlib
    return _hostlib.ring_encoding(n_atoms, bonds)


def ring_encoding_py(n_atoms: int, bonds: Sequence[Tuple[int, int]]) -> np.ndarray:
    bonds = np.asarray(bonds, dtype=np.int64).reshape(-1, 2)
    adj = _adjacency(n_atoms, bonds)
    in_ring = _ring_atoms(n_atoms, adj)
    enc = np.zeros((n_atoms, 7), dtype=np.float32)
    enc[:, 0] = in_ring
    if in_ring.any():
        for length, _, atoms in _relevant(_small_cycles(n_atoms, adj, in_ring)):
            enc[list(atoms), length - 2] = 1.0
    return enc


def degree_encoding(n_atoms: int, bonds: Sequence[Tuple[int, int]]) -> np.ndarray:
    from . import _hostlib
    return _hostlib.degree_encoding(n_atoms, bonds)


def degree_encoding_py(n_atoms: int, bonds: Sequence[Tuple[int, int]]) -> np.ndarray:
    bonds = np.asarray(bonds, dtype=np.int64).reshape(-1, 2)
    deg = np.bincount(bonds.reshape(-1), minlength=n_atoms)
    enc = np.zeros((n_atoms, 6), dtype=np.float32)
    for i in range(1, 7):
        enc[:, i - 1] = deg == i
    return enc
