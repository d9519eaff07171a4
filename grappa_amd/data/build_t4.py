"""Build `t4_lysozyme.npz`: the all-atom graph of T4 lysozyme for BASELINE.json configs[4] (50k-atom protein inference).

Runs ONLY in the build container: it reads the structure the reference ships as example data
(/root/reference/examples/usage/T4.pdb: 164 residues, 2,634 atoms incl. hydrogens) and the residue templates + amber99 charges of
/root/reference/src/grappa/utils/amber99sbildn-star_.xml through `grappa_amd.pdb.graph_from_pdb`.  The output is data, not code:
element numbers, bond list (atom indices), template partial charges and the crystal coordinates.

    python grappa_amd/data/build_t4.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from grappa_amd.pdb import graph_from_pdb  # noqa: E402

if __name__ == "__main__":
    g = graph_from_pdb("/root/reference/examples/usage/T4.pdb", "/root/reference/src/grappa/utils/amber99sbildn-star_.xml")
    n = len(g["z"])
    assert n == 2634 and len(g["bonds"]) == n - 1 + 21          # one chain; rings: 5 PHE + 6 TYR + 2x3 TRP + 3 PRO + 1 HIS
    assert abs(float(g["charges"].sum()) - 8.0) < 1e-3           # net charge of the T4L* construct
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "t4_lysozyme.npz")
    np.savez_compressed(out, z=g["z"].astype(np.int16), bonds=g["bonds"].astype(np.int32), charges=g["charges"], xyz=g["xyz"],
                        residue_ptr=g["residue_ptr"].astype(np.int32), residue_templates=np.array(g["residue_templates"]))
    print("wrote", out, os.path.getsize(out), "bytes")
