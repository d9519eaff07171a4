"""CPU: the oracle (oracle/cpu_ref.py) against the golden fixtures produced by the reference's own
modules (oracle/make_goldens.py).  This is what pins the oracle (section (3) of the task brief)."""
import numpy as np
import pytest
import torch

import golden_utils as gu
from oracle import cpu_ref

FLOORS = {"k": 1e-3, "eq": 1e-4, "energy": 1e-3, "gradient": 1e-2}
TOL = 2e-5     # fp32 CPU restatement vs fp32 CPU reference


def _run(fx_name, n_confs, with_param_refs):
    fx = gu.load(fx_name)
    cfg = gu.config_of(fx)
    mols = gu.molecules_of(fx)
    g = gu.build_batch(mols, n_confs, with_param_refs, (cfg["n_periodicity_proper"], cfg["n_periodicity_improper"]))
    model = cpu_ref.RefGrappaModel(**cfg)
    model.load_state_dict(gu.weights_for(fx, model))
    model.eval()
    g = cpu_ref.RefEnergy(**gu.energy_kwargs_of(fx))(model(g))
    loss = cpu_ref.RefMolwiseLoss(**gu.loss_kwargs_of(fx))(g)
    loss.backward()
    return fx, g, loss, model


def _check_outputs(out, g, loss, scaled=False):
    for lvl in ["n2", "n3", "n4", "n4_improper"]:
        assert np.array_equal(out[f"{lvl}_idxs"], g.nodes[lvl].data["idxs"].numpy()), lvl
        assert gu.rel_err(g.nodes[lvl].data["k"].detach(), out[f"{lvl}_k"], FLOORS["k"]) < TOL, lvl
        if lvl in ("n2", "n3"):
            assert gu.rel_err(g.nodes[lvl].data["eq"].detach(), out[f"{lvl}_eq"], FLOORS["eq"]) < TOL, lvl
    assert gu.rel_err(g.nodes["n1"].data["h"].detach(), out["h"], 1e-3) < TOL
    if scaled:
        # ref_tiny_*: the key-derived weights of the tiny model give forces up to 1e3 kcal/mol/A, whose fp32 ulp (6e-5) is already 2e-4 of a
        # component that cancels to 0.4 -- the floor follows the tensor as in tests/test_host_model.py (1e-3 / 1e-2 of its largest element)
        assert gu.rel_err_scaled(g.nodes["g"].data["energy"].detach(), out["energy"], 1e-3, FLOORS["energy"]) < 1e-4
        assert gu.rel_err_scaled(g.nodes["n1"].data["gradient"].detach(), out["gradient"], 1e-2, FLOORS["gradient"]) < 1e-4
    else:
        assert gu.rel_err(g.nodes["g"].data["energy"].detach(), out["energy"], FLOORS["energy"]) < 1e-4
        assert gu.rel_err(g.nodes["n1"].data["gradient"].detach(), out["gradient"], FLOORS["gradient"]) < 1e-4
    assert gu.rel_err(loss.detach(), out["loss"], 1e-6) < 1e-4


# ref_tiny_*: the reference's wrong_symmetry=True / harmonic_gate=True / n_periodicity_proper=3 (gated) / Energy(offset_torsion=True)
@pytest.mark.parametrize("name,n_confs,refs", [("ref_small_att.npz", 4, True), ("ref_small_conv.npz", 5, False),
                                                 ("ref_tiny_wrongsym.npz", 4, True), ("ref_tiny_harmonic_gate.npz", 4, True),
                                                 ("ref_tiny_nper3.npz", 4, True), ("ref_tiny_offset_torsion.npz", 4, True), ("ref_tiny_nopos.npz", 4, True)])
def test_small_config_matches_reference(name, n_confs, refs):
    fx, g, loss, model = _run(name, n_confs, refs)
    out = gu.outputs_of(fx)
    _check_outputs(out, g, loss, scaled=name.startswith("ref_tiny"))
    # every parameter gradient (double backward through the forces included)
    n = 0
    for k, p in model.named_parameters():
        ref = out.get("grad::" + k)
        if ref is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        scale = max(float(np.abs(ref).max()), 1e-8)
        # measured: 1e-6 on the ref_small_* fixtures; 2.1e-5 on one bias gradient of ref_tiny_wrongsym (a sum over all improper tokens of six
        # permuted copies each, |g| = 84: fp32 summation order), hence 5e-5 there -- still half of the north star's 1e-4
        assert float(np.abs(p.grad.numpy() - ref).max()) / scale < (5e-5 if name.startswith("ref_tiny") else TOL), k
        n += 1
    assert n > 50
    if "is_dummy" in fx.files:
        assert np.array_equal(fx["is_dummy"], g.nodes["g"].data["is_dummy"].numpy())


def test_state_dict_layout_matches_reference():
    fx = gu.load("ref_small_att.npz")
    sd_ref = gu.state_dict_of(fx)
    model = cpu_ref.RefGrappaModel(**gu.config_of(fx))
    sd = model.state_dict()
    assert list(sd.keys()) == list(sd_ref.keys())
    for k in sd:
        assert tuple(sd[k].shape) == tuple(sd_ref[k].shape), k
        assert sd[k].dtype == sd_ref[k].dtype, k


def test_energy_on_classical_parameters():
    fx = gu.load("ref_energy.npz")
    out = gu.outputs_of(fx)
    g = gu.build_batch(gu.molecules_of(fx), 6, True, nan_refs=False)
    g = cpu_ref.RefEnergy(suffix="_ref", write_suffix="_classical")(g)
    for lvl in ["n2", "n3", "n4", "n4_improper"]:
        assert np.allclose(g.nodes[lvl].data["k_ref"].numpy(), out[f"{lvl}_k_ref"], equal_nan=True)
        assert gu.rel_err(g.nodes[lvl].data["x"].detach(), out[f"{lvl}_x"], 1e-3) < 1e-5, lvl
    assert gu.rel_err(g.nodes["g"].data["energy_classical"].detach(), out["energy"], 1e-3) < 1e-5
    assert gu.rel_err(g.nodes["n1"].data["gradient_classical"].detach(), out["gradient"], 1e-2) < 1e-4
    g = cpu_ref.RefEnergy(suffix="_ref", write_suffix="_offs", offset_torsion=True)(g)
    assert gu.rel_err(g.nodes["g"].data["energy_offs"].detach(), out["energy_offs"], 1e-3) < 1e-5


def test_noise_effect_is_bounded():
    fx = gu.load("ref_small_att.npz")
    # the reference's dihedral noise moves E/F by << the parity tolerance
    assert float(fx["noise::energy_maxabs"][0]) < 1e-2
    assert float(fx["noise::gradient_maxabs"][0]) < 5e-1
