"""CPU: the PDB + force-field-XML front-end (grappa_amd/pdb.py) on a hand-written two-residue structure with a
hand-written template file (terminal variants, the N-terminal H -> H1 alias, peptide bond, charges by atom type), and the
committed T4 lysozyme graph it produced from the reference's example structure (grappa_amd/data/build_t4.py)."""
import numpy as np
import pytest

XML = """<ForceField>
 <AtomTypes>
  <Type name="n3" class="N3" element="N" mass="14"/> <Type name="h" class="H" element="H" mass="1"/>
  <Type name="ct" class="CT" element="C" mass="12"/> <Type name="c" class="C" element="C" mass="12"/>
  <Type name="o" class="O" element="O" mass="16"/> <Type name="n" class="N" element="N" mass="14"/>
  <Type name="o2" class="O2" element="O" mass="16"/>
 </AtomTypes>
 <Residues>
  <Residue name="GLY"><Atom name="N" type="n"/><Atom name="H" type="h"/><Atom name="CA" type="ct"/><Atom name="HA2" type="h"/>
   <Atom name="HA3" type="h"/><Atom name="C" type="c"/><Atom name="O" type="o"/>
   <Bond from="0" to="1"/><Bond from="0" to="2"/><Bond from="2" to="3"/><Bond from="2" to="4"/><Bond from="2" to="5"/><Bond from="5" to="6"/>
   <ExternalBond from="0"/><ExternalBond from="5"/></Residue>
  <Residue name="NGLY"><Atom name="N" type="n3"/><Atom name="H1" type="h"/><Atom name="H2" type="h"/><Atom name="H3" type="h"/>
   <Atom name="CA" type="ct"/><Atom name="HA2" type="h"/><Atom name="HA3" type="h"/><Atom name="C" type="c"/><Atom name="O" type="o"/>
   <Bond from="0" to="1"/><Bond from="0" to="2"/><Bond from="0" to="3"/><Bond from="0" to="4"/><Bond from="4" to="5"/><Bond from="4" to="6"/>
   <Bond from="4" to="7"/><Bond from="7" to="8"/><ExternalBond from="7"/></Residue>
  <Residue name="CGLY"><Atom name="N" type="n"/><Atom name="H" type="h"/><Atom name="CA" type="ct"/><Atom name="HA2" type="h"/>
   <Atom name="HA3" type="h"/><Atom name="C" type="c"/><Atom name="O" type="o2"/><Atom name="OXT" type="o2"/>
   <Bond from="0" to="1"/><Bond from="0" to="2"/><Bond from="2" to="3"/><Bond from="2" to="4"/><Bond from="2" to="5"/><Bond from="5" to="6"/>
   <Bond from="5" to="7"/><ExternalBond from="0"/></Residue>
 </Residues>
 <NonbondedForce coulomb14scale="0.8" lj14scale="0.5">
  <Atom type="n3" charge="0.3" sigma="1" epsilon="1"/><Atom type="h" charge="0.1" sigma="1" epsilon="1"/>
  <Atom type="ct" charge="-0.05" sigma="1" epsilon="1"/><Atom type="c" charge="0.6" sigma="1" epsilon="1"/>
  <Atom type="o" charge="-0.55" sigma="1" epsilon="1"/><Atom type="n" charge="-0.4" sigma="1" epsilon="1"/>
  <Atom type="o2" charge="-0.8" sigma="1" epsilon="1"/>
 </NonbondedForce>
</ForceField>
"""


def _atom(i, name, res, resid, x):
    return f"ATOM  {i:5d} {name:<4s} {res:>3s} A{resid:4d}    {x:8.3f}{0.0:8.3f}{0.0:8.3f}  1.00  0.00\n"


def test_two_residue_peptide(tmp_path):
    from grappa_amd.pdb import graph_from_pdb
    # residue 1 lists its atoms in a different order than the template, and calls the first N-terminal hydrogen "H"
    r1 = ["CA", "N", "H", "H2", "H3", "HA2", "HA3", "C", "O"]
    r2 = ["N", "H", "CA", "HA2", "HA3", "C", "O", "OXT"]
    pdb = "".join(_atom(i + 1, n, "GLY", 1, float(i)) for i, n in enumerate(r1))
    pdb += "".join(_atom(len(r1) + i + 1, n, "GLY", 2, 20.0 + i) for i, n in enumerate(r2)) + "END\n"
    (tmp_path / "gg.pdb").write_text(pdb)
    (tmp_path / "ff.xml").write_text(XML)
    g = graph_from_pdb(str(tmp_path / "gg.pdb"), str(tmp_path / "ff.xml"))
    assert g["residue_templates"] == ["NGLY", "CGLY"] and g["residue_ptr"].tolist() == [0, 9, 17]
    assert g["z"].tolist() == [6, 7, 1, 1, 1, 1, 1, 6, 8] + [7, 1, 6, 1, 1, 6, 8, 8]
    want = {(1, 2), (1, 3), (1, 4), (1, 0), (0, 5), (0, 6), (0, 7), (7, 8),             # NGLY in file order (CA first)
            (9, 10), (9, 11), (11, 12), (11, 13), (11, 14), (14, 15), (14, 16), (7, 9)}   # CGLY + the peptide bond C(1)-N(2)
    assert {tuple(sorted(b)) for b in g["bonds"].tolist()} == {tuple(sorted(b)) for b in want}
    q = g["charges"]
    assert np.isclose(q[1], 0.3) and np.isclose(q[0], -0.05) and np.isclose(q[15], -0.8) and np.isclose(q[16], -0.8)
    assert np.isclose(q.sum(), (0.3 + 0.3 - 0.05 + 0.2 + 0.6 - 0.55) + (-0.4 + 0.1 - 0.05 + 0.2 + 0.6 - 1.6), atol=1e-6)
    with pytest.raises(ValueError):
        (tmp_path / "bad.pdb").write_text(_atom(1, "XX", "GLY", 1, 0.0))
        graph_from_pdb(str(tmp_path / "bad.pdb"), str(tmp_path / "ff.xml"))


def test_t4_lysozyme_data_file():
    """facts of the construct that do not depend on the parser's internals: 164 residues, 2,634 atoms, one chain = one connected
    component, 21 rings (5 PHE + 6 TYR + 3 TRP x 2 + 3 PRO + 1 HIS), net charge +8, chemically sensible degrees"""
    from grappa_amd import _hostlib
    from grappa_amd.datasets import _T4_PATH, protein_graph_t4
    d = np.load(_T4_PATH)
    n = len(d["z"])
    assert n == 2634 and len(d["residue_templates"]) == 164 and len(d["bonds"]) == n - 1 + 21
    assert abs(float(d["charges"].sum()) - 8.0) < 1e-3
    assert str(d["residue_templates"][0]) == "NMET" and str(d["residue_templates"][-1]) == "CLEU"
    deg = np.bincount(d["bonds"].reshape(-1), minlength=n)
    z = d["z"]
    assert (deg[z == 1] == 1).all() and (deg[z == 6] >= 3).all() and deg.max() == 4
    bl = np.linalg.norm(d["xyz"][d["bonds"][:, 0]] - d["xyz"][d["bonds"][:, 1]], axis=1)
    assert 0.9 < bl.min() and bl.max() < 1.9
    ring = _hostlib.ring_encoding(n, d["bonds"])
    assert int(ring[:, 0].sum()) == 5 * 6 + 6 * 6 + 3 * 9 + 3 * 5 + 5            # PHE, TYR: 6 atoms; TRP: 9; PRO: 5; HIS: 5
    g = protein_graph_t4(2)
    assert g.num_nodes("n1") == 2 * n and g.batch_size == 1 and g.num_nodes("n2") == 2 * len(d["bonds"])
