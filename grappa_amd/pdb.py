"""PDB + OpenMM force-field XML front-end: a protein structure file and a residue-template file ->
element numbers, bond list and template partial charges, i.e. the inputs of `Molecule.from_graph` / `Grappa.predict`.

The reference goes through OpenMM for this (`Molecule.from_openmm_system`, data/Molecule.py:270-346: topology from
`PDBFile`, charges from `ForceField.createSystem`); OpenMM is not available offline, so the two file formats are read directly:

  * force-field XML (`<Residues><Residue name=..><Atom name type/><Bond from to/><ExternalBond from/>`, charges per atom type in
    `<NonbondedForce><Atom type charge/>`, elements in `<AtomTypes><Type name element/>`),
  * PDB `ATOM`/`HETATM` records (columns 13-16 atom name, 18-20 residue name, 22 chain, 23-27 residue number + insertion code).

A residue is matched to the template with the same set of atom names, trying the plain name, the N-/C-terminal variants
(`N<name>` / `C<name>`, with the PDB's N-terminal `H` read as the template's `H1`), and the protonation variants of HIS / CYS / ASP /
GLU / LYS.  Bonds = template bonds + the peptide bond C(i)-N(i+1) between consecutive residues of a chain that both expose those
external bonds + disulfide bridges (CYX SG atoms closer than 2.5 A).
"""
from __future__ import annotations

import xml.etree.ElementTree as ET
from typing import Dict, List, NamedTuple, Tuple

import numpy as np

_ELEMENT_Z = {"H": 1, "C": 6, "N": 7, "O": 8, "F": 9, "Na": 11, "Mg": 12, "P": 15, "S": 16, "Cl": 17, "K": 19, "Ca": 20, "Zn": 30, "Br": 35, "I": 53}
_VARIANTS = {"HIS": ["HIS", "HID", "HIE", "HIP"], "CYS": ["CYS", "CYX", "CYM"], "ASP": ["ASP", "ASH"], "GLU": ["GLU", "GLH"],
             "LYS": ["LYS", "LYN"]}


class Template(NamedTuple):
    names: Tuple[str, ...]
    types: Tuple[str, ...]
    bonds: Tuple[Tuple[int, int], ...]
    external: Tuple[int, ...]


class ForceFieldTemplates:
    def __init__(self, xml_path: str):
        root = ET.parse(xml_path).getroot()
        self.element = {t.get("name"): t.get("element") for t in root.iter("Type")}
        self.charge = {}
        for nb in root.iter("NonbondedForce"):
            for a in nb.iter("Atom"):
                if a.get("type") is not None and a.get("charge") is not None:
                    self.charge[a.get("type")] = float(a.get("charge"))
        self.residues: Dict[str, Template] = {}
        for r in root.iter("Residue"):
            atoms = r.findall("Atom")
            names = tuple(a.get("name") for a in atoms)
            index = {n: i for i, n in enumerate(names)}

            def ref(b, key):
                return int(b.get(key)) if b.get(key) is not None else index[b.get("atomName" + ("1" if key == "from" else "2"))]

            bonds = tuple((ref(b, "from"), ref(b, "to")) for b in r.findall("Bond"))
            ext = tuple(int(e.get("from")) if e.get("from") is not None else index[e.get("atomName")] for e in r.findall("ExternalBond"))
            self.residues[r.get("name")] = Template(names, tuple(a.get("type") for a in atoms), bonds, ext)

    def match(self, resname: str, atom_names: List[str], first: bool, last: bool):
        """-> (template name, permutation: template atom i = residue atom perm[i])"""
        base = _VARIANTS.get(resname, [resname])
        cands = list(base)
        if first:
            cands = ["N" + b for b in base] + cands
        if last:
            cands = ["C" + b for b in base] + cands
        have = list(atom_names)
        for c in cands:
            t = self.residues.get(c)
            if t is None or len(t.names) != len(have):
                continue
            names = list(have)
            if "H1" in t.names and "H1" not in names and "H" in names:        # PDB writers name the first N-terminal hydrogen H
                names[names.index("H")] = "H1"
            if sorted(names) == sorted(t.names):
                pos = {n: i for i, n in enumerate(names)}
                return c, [pos[n] for n in t.names]
        raise ValueError(f"no residue template matches {resname} with atoms {sorted(atom_names)}")


def read_pdb_atoms(pdb_path: str):
    """-> list of residues [(chain, resname, [(atom name, xyz)])] of the first model, in file order"""
    residues, key = [], None
    with open(pdb_path) as f:
        for line in f:
            rec = line[:6]
            if rec == "ENDMDL":
                break
            if rec not in ("ATOM  ", "HETATM"):
                continue
            k = (line[21], line[22:27])
            if k != key:
                residues.append((line[21], line[17:20].strip(), []))
                key = k
            residues[-1][2].append((line[12:16].strip(), (float(line[30:38]), float(line[38:46]), float(line[46:54]))))
    return residues


def graph_from_pdb(pdb_path: str, ffxml_path: str):
    """-> dict(z (n,) int64, bonds (m,2) int64 atom indices in file order, charges (n,) float32, xyz (n,3) float32 in Angstrom,
    residue_ptr (R+1,), residue_templates [R])"""
    ff = ForceFieldTemplates(ffxml_path)
    residues = read_pdb_atoms(pdb_path)
    z, q, xyz, bonds, ptr, tnames = [], [], [], [], [0], []
    ext_atoms = []           # per residue: {atom name: global index} of the atoms that carry an external bond
    for ri, (chain, resname, atoms) in enumerate(residues):
        first = ri == 0 or residues[ri - 1][0] != chain
        last = ri == len(residues) - 1 or residues[ri + 1][0] != chain
        tname, perm = ff.match(resname, [a[0] for a in atoms], first, last)
        t = ff.residues[tname]
        base = ptr[-1]
        glob = [base + p for p in perm]                                       # template atom i -> global atom index
        zr, qr = [0] * len(atoms), [0.0] * len(atoms)
        for i, ty in enumerate(t.types):
            zr[perm[i]] = _ELEMENT_Z[ff.element[ty]]
            qr[perm[i]] = ff.charge[ty]
        z += zr
        q += qr
        xyz += [a[1] for a in atoms]
        bonds += [(glob[a], glob[b]) for a, b in t.bonds]
        ext_atoms.append({t.names[i]: glob[i] for i in t.external})
        ptr.append(base + len(atoms))
        tnames.append(tname)
    for ri in range(len(residues) - 1):                                       # peptide bonds
        if residues[ri][0] == residues[ri + 1][0] and "C" in ext_atoms[ri] and "N" in ext_atoms[ri + 1]:
            bonds.append((ext_atoms[ri]["C"], ext_atoms[ri + 1]["N"]))
    xyz = np.asarray(xyz, dtype=np.float32)
    sg = [e["SG"] for e, t in zip(ext_atoms, tnames) if "SG" in e and t.endswith("CYX")]
    used = set()
    for i, a in enumerate(sg):                                                # disulfide bridges
        for b in sg[i + 1:]:
            if a not in used and b not in used and float(np.linalg.norm(xyz[a] - xyz[b])) < 2.5:
                bonds.append((a, b))
                used.update((a, b))
    return {"z": np.asarray(z, dtype=np.int64), "bonds": np.asarray(bonds, dtype=np.int64).reshape(-1, 2), "charges": np.asarray(q, dtype=np.float32),
            "xyz": xyz, "residue_ptr": np.asarray(ptr, dtype=np.int64), "residue_templates": tnames}
