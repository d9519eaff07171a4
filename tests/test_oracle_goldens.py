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
    model.load_state_dict(gu.state_dict_of(fx))
    model.eval()
    g = cpu_ref.RefEnergy()(model(g))
    loss = cpu_ref.RefMolwiseLoss(**gu.loss_kwargs_of(fx))(g)
    loss.backward()
    return fx, g, loss, model


def _check_outputs(out, g, loss):
    for lvl in ["n2", "n3", "n4", "n4_improper"]:
        assert np.array_equal(out[f"{lvl}_idxs"], g.nodes[lvl].data["idxs"].numpy()), lvl
        assert gu.rel_err(g.nodes[lvl].data["k"].detach(), out[f"{lvl}_k"], FLOORS["k"]) < TOL, lvl
        if lvl in ("n2", "n3"):
            assert gu.rel_err(g.nodes[lvl].data["eq"].detach(), out[f"{lvl}_eq"], FLOORS["eq"]) < TOL, lvl
    assert gu.rel_err(g.nodes["n1"].data["h"].detach(), out["h"], 1e-3) < TOL
    assert gu.rel_err(g.nodes["g"].data["energy"].detach(), out["energy"], FLOORS["energy"]) < 1e-4
    assert gu.rel_err(g.nodes["n1"].data["gradient"].detach(), out["gradient"], FLOORS["gradient"]) < 1e-4
    assert gu.rel_err(loss.detach(), out["loss"], 1e-6) < 1e-4


@pytest.mark.parametrize("name,n_confs,refs", [("ref_small_att.npz", 4, True), ("ref_small_conv.npz", 5, False)])
def test_small_config_matches_reference(name, n_confs, refs):
    fx, g, loss, model = _run(name, n_confs, refs)
    out = gu.outputs_of(fx)
    _check_outputs(out, g, loss)
    # every parameter gradient (double backward through the forces included)
    n = 0
    for k, p in model.named_parameters():
        ref = out.get("grad::" + k)
        if ref is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        scale = max(float(np.abs(ref).max()), 1e-8)
        assert float(np.abs(p.grad.numpy() - ref).max()) / scale < TOL, k      # measured: 1e-6
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
