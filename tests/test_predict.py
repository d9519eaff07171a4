"""P1: `Grappa.predict` -> `Parameters.from_dgl` against tests/golden/ref_predict.npz, which holds what the
REFERENCE's own `Grappa.predict` (grappa.py:36-57, data/Parameters.py:62-140) returned for four pool molecules with the
production config, and what its `Parameters.from_dgl` made of a hand-filled graph (exact zeros and both signs in the torsion
tables; the two raise conditions).  CPU: `from_dgl` bit-exact on the hand-filled graph, the oracle's forward through
`from_dgl`; GPU (-m gpu): the product's `Grappa.predict` on the HIP kernels."""
import numpy as np
import pytest
import torch

import golden_utils as gu

FIELDS = ["atoms", "bonds", "bond_k", "bond_eq", "angles", "angle_k", "angle_eq", "propers", "proper_ks", "proper_phases",
          "impropers", "improper_ks", "improper_phases"]
INDEX_FIELDS = {"atoms", "bonds", "angles", "propers", "impropers"}
FLOOR = {"bond_k": 1e-3, "angle_k": 1e-3, "bond_eq": 1e-4, "angle_eq": 1e-4, "proper_ks": 5e-2, "improper_ks": 5e-2}   # tests/test_host_model.py
TOL = 1e-4


def _hand_graph(fx):
    g = gu.molecule_of(gu.molecules_of(fx)[0]).to_dgl()
    for lvl in ("n2", "n3", "n4", "n4_improper"):
        g.nodes[lvl].data["k"] = torch.from_numpy(fx[f"hand::{lvl}_k"].copy())
        if lvl in ("n2", "n3"):
            g.nodes[lvl].data["eq"] = torch.from_numpy(fx[f"hand::{lvl}_eq"].copy())
    return g


def _compare(params, fx, prefix, exact):
    for f in FIELDS:
        ref, got = fx[prefix + f], np.asarray(getattr(params, f))
        assert got.shape == ref.shape and got.dtype == ref.dtype, (f, got.shape, got.dtype, ref.shape, ref.dtype)
        if f in INDEX_FIELDS or exact:
            assert np.array_equal(got, ref), f
        elif f.endswith("phases"):
            # a phase can only differ where the reference's k is at rounding distance from zero (the torsion cutoff sets those
            # to exactly zero on both sides, where `>=` / `>` decide)
            ks = fx[prefix + f.replace("phases", "ks")]
            assert np.array_equal(got[ks > 1e-3], ref[ks > 1e-3]), f
        else:
            assert gu.rel_err(got, ref, FLOOR[f]) < TOL, f


def test_from_dgl_matches_the_reference_on_a_hand_filled_graph():
    from grappa_amd.parameters import Parameters
    fx = gu.load("ref_predict.npz")
    g = _hand_graph(fx)
    _compare(Parameters.from_dgl(g), fx, "hand::out::", exact=True)
    # exact zeros: phase 0 for propers (k >= 0), phase pi for impropers (k > 0)  -- SURVEY Q8
    p = Parameters.from_dgl(g)
    assert (fx["hand::n4_k"] == 0).any() and (fx["hand::n4_improper_k"] == 0).any()
    assert (p.proper_phases[fx["hand::n4_k"] == 0] == 0).all()
    assert np.allclose(p.improper_phases[fx["hand::n4_improper_k"] == 0], np.pi)
    # raise conditions: the reference raised RuntimeError for an angle below 45 degrees and for a bond below 0.5 Angstrom, and
    # accepted values one per cent above the limits
    assert fx["hand::raises"].tolist() == ["RuntimeError", "RuntimeError"]
    for lvl, bad in (("n3", np.pi / 180 * 44.9), ("n2", 0.499)):
        keep = g.nodes[lvl].data["eq"].clone()
        g.nodes[lvl].data["eq"][1] = bad
        with pytest.raises(RuntimeError):
            Parameters.from_dgl(g)
        Parameters.from_dgl(g, check_eq_values=False)
        g.nodes[lvl].data["eq"][1] = bad * 1.01
        Parameters.from_dgl(g)
        g.nodes[lvl].data["eq"] = keep


def test_oracle_predict_matches_reference_predict():
    """the oracle's forward + the product's from_dgl == the reference's Grappa.predict (pins the oracle for P1)."""
    from grappa_amd import get_default_model_config
    from grappa_amd.parameters import Parameters
    from oracle import cpu_ref
    fx = gu.load("ref_predict.npz")
    model = cpu_ref.RefGrappaModel(**get_default_model_config())
    model.load_state_dict(gu.keyed_state_dict(model))
    model.eval()
    for i, m in enumerate(gu.molecules_of(fx)):
        with torch.no_grad():
            g = model(gu.molecule_of(m).to_dgl())
        _compare(Parameters.from_dgl(g), fx, f"pred{i}::", exact=False)


@pytest.mark.gpu
def test_predict_on_the_gpu_matches_reference_predict():
    from grappa_amd import Grappa, get_default_model_config, model_from_config
    fx = gu.load("ref_predict.npz")
    model = model_from_config(get_default_model_config())
    model.load_state_dict(gu.keyed_state_dict(model))
    wrapper = Grappa(model, device="cuda")
    mols = gu.molecules_of(fx)
    assert len(mols) == 4 and len(mols[3]["impropers"]) == 0          # one molecule without impropers: (0, 3) tables
    for i, m in enumerate(mols):
        params = wrapper.predict(gu.molecule_of(m))
        _compare(params, fx, f"pred{i}::", exact=False)


@pytest.mark.gpu
def test_predict_with_the_reference_water_guard_parametrises_water_like_the_reference_would():
    """the reference's water guard compares Z - 1 with {1, 8} and so never fires on water (utils/dgl_utils.py:231-234 with data/Molecule.py:521):
    `Grappa(..., reference_water_guard=True)` reproduces that -- water gets parameters, equal to the oracle's -- while the default (the intended
    guard) raises"""
    from grappa_amd import Grappa, get_default_model_config, model_from_config
    from grappa_amd.molecule import Molecule
    from grappa_amd.parameters import Parameters
    from oracle import cpu_ref
    model = model_from_config(get_default_model_config())
    model.load_state_dict(gu.keyed_state_dict(model))
    water = Molecule(atoms=[1, 2, 3], bonds=[(1, 2), (1, 3)], impropers=[], atomic_numbers=[8, 1, 1], partial_charges=[-0.8, 0.4, 0.4])
    with pytest.raises(ValueError):
        Grappa(model, device="cuda").predict(water)
    got = Grappa(model, device="cuda", reference_water_guard=True).predict(water)
    ref = cpu_ref.RefGrappaModel(**get_default_model_config())
    ref.load_state_dict(gu.keyed_state_dict(ref))
    ref.eval()
    with torch.no_grad():
        want = Parameters.from_dgl(ref(water.to_dgl()))
    assert got.bond_k.shape == (2,) and got.angle_k.shape == (1,)
    for name in ("bond_k", "bond_eq", "angle_k", "angle_eq"):
        a, b = np.asarray(getattr(got, name)), np.asarray(getattr(want, name))
        assert np.allclose(a, b, rtol=1e-4, atol=1e-4 * max(1e-3, float(np.abs(b).max()))), (name, a, b)
