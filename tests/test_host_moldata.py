"""CPU: MolData .npz reader against the graph the reference's own MolData.from_dict(...).to_dgl() builds from
the same record (tests/golden/ref_moldata.npz)."""
import os

import numpy as np
import torch

import golden_utils as gu
from grappa_amd.moldata import MolData


def test_moldata_record_to_graph_matches_reference(tmp_path):
    fx = gu.load("ref_moldata.npz")
    record = {k[len("record::"):]: fx[k] for k in fx.files if k.startswith("record::")}
    md = MolData.from_dict(record)
    path = os.path.join(tmp_path, "rec.npz")
    md.save(path)                                   # round trip through the on-disk schema
    g = MolData.load(path).to_dgl()
    n = 0
    for key in fx.files:
        if not key.startswith("graph::"):
            continue
        _, nt, feat = key.split("::")
        got = g.nodes[nt].data[feat].numpy()
        assert got.shape == fx[key].shape, key
        assert np.array_equal(got, fx[key], equal_nan=True), key
        n += 1
    assert n >= 20
    assert g.nodes["n1"].data["xyz"].shape == (len(record["atoms"]), 5, 3)
