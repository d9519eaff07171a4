"""CPU: collate against the reference's collate_fn output recorded in tests/golden/ref_collate.npz
(same torch seed -> same randperm sub-sampling), plus unbatch(batch(x)) == x (reference tests/dgl_utils.py:42-53)."""
import numpy as np
import torch

import golden_utils as gu
from grappa_amd.batch import batch, set_number_confs, unbatch
from grappa_amd.dataloader import GraphDataLoader, get_collate_fn
from grappa_amd.molecule import Molecule


def _graphs(fx):
    out = []
    for m in gu.molecules_of(fx):
        mol = Molecule(atoms=list(range(len(m["z"]))), bonds=[tuple(int(x) for x in b) for b in m["bonds"]],
                       impropers=[tuple(int(x) for x in r) for r in m["impropers"]], atomic_numbers=[int(x) for x in m["z"]],
                       partial_charges=[float(x) for x in m["q"]], charge_model=str(m["charge_model"]))
        g = mol.to_dgl()
        g.nodes["n1"].data["xyz"] = torch.from_numpy(m["xyz"].copy())
        g.nodes["g"].data["energy_ref"] = torch.from_numpy(m["energy_ref"].copy())
        g.nodes["n1"].data["gradient_ref"] = torch.from_numpy(m["gradient_ref"].copy())
        out.append(set_number_confs(g, m["xyz"].shape[1]))
    return out


def test_collate_matches_reference():
    fx = gu.load("ref_collate.npz")
    out = gu.outputs_of(fx)
    for strategy in (4, "min", "max", "mean"):
        graphs = _graphs(fx)
        torch.manual_seed(1234)
        gb, names = get_collate_fn(conf_strategy=strategy)([(g, f"ds{i % 2}") for i, g in enumerate(graphs)])
        assert names == ("ds0", "ds1", "ds0", "ds1")
        for key, val in (("xyz", gb.nodes["n1"].data["xyz"]), ("is_dummy", gb.nodes["g"].data["is_dummy"]),
                         ("energy_ref", gb.nodes["g"].data["energy_ref"]), ("gradient_ref", gb.nodes["n1"].data["gradient_ref"]),
                         ("n4_idxs", gb.nodes["n4"].data["idxs"])):
            assert np.array_equal(out[f"{strategy}::{key}"], val.numpy()), (strategy, key)
        # the inputs are not mutated by collate
        assert [g.nodes["n1"].data["xyz"].shape[1] for g in graphs] == [9, 6, 3, 7]


def test_unbatch_inverts_batch_bit_exactly():
    fx = gu.load("ref_collate.npz")
    graphs = [set_number_confs(g, 5) for g in _graphs(fx)]
    back = unbatch(batch(graphs))
    for a, b in zip(graphs, back):
        for nt in a.ntypes:
            assert a.num_nodes(nt) == b.num_nodes(nt)
            for k, v in a.nodes[nt].data.items():
                if k == "is_dummy":
                    continue
                w = b.nodes[nt].data[k]
                if "xyz" in k or "energy" in k or "gradient" in k:      # dummy conformations are removed by unbatch
                    keep = a.nodes["g"].data["is_dummy"][0] == 0
                    v = v[:, keep] if v.dim() >= 2 else v
                assert torch.equal(v, w), (nt, k)
        assert torch.equal(torch.sort(torch.stack(a.edges()), dim=1)[0], torch.sort(torch.stack(b.edges()), dim=1)[0])
    # every atom appears in some bond (reference tests/dgl_utils.py:46-53)
    for g in back:
        assert len(torch.unique(g.nodes["n2"].data["idxs"])) == g.num_nodes("n1")


def test_loader_iterates_batches():
    fx = gu.load("ref_collate.npz")
    ds = [(g, "a" if i < 2 else "b") for i, g in enumerate(_graphs(fx))]
    loader = GraphDataLoader(ds, batch_size=2, conf_strategy=4)
    sizes = [(g.batch_size, g.nodes["n1"].data["xyz"].shape[1], names) for g, names in loader]
    assert sizes == [(2, 4, ("a", "a")), (2, 4, ("b", "b"))]
    torch.manual_seed(0)
    loader = GraphDataLoader(ds, batch_size=4, shuffle=True, weights={"a": 3.0}, balance_factor=0.5, conf_strategy="min")
    g, names = next(iter(loader))
    assert g.batch_size == 4 and g.nodes["n1"].data["xyz"].shape[1] >= 3
