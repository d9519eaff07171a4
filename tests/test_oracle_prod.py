"""CPU: production-size golden (weights regenerated from the state-dict keys) against the oracle."""
import numpy as np
import torch

import golden_utils as gu
from oracle import cpu_ref


def test_production_golden_matches_oracle():
    fx = gu.load("ref_prod.npz")
    cfg = gu.config_of(fx)
    model = cpu_ref.RefGrappaModel(**cfg)
    model.load_state_dict(gu.keyed_state_dict(model))
    model.eval()
    g = gu.build_batch(gu.molecules_of(fx), 4, True)
    g = cpu_ref.RefEnergy()(model(g))
    loss = cpu_ref.RefMolwiseLoss(**gu.loss_kwargs_of(fx))(g)
    loss.backward()
    out = gu.outputs_of(fx)
    for lvl in ["n2", "n3", "n4", "n4_improper"]:
        floor = 5e-2 if lvl.startswith("n4") else 1e-3
        assert gu.rel_err(g.nodes[lvl].data["k"].detach(), out[f"{lvl}_k"], floor) < 1e-4, lvl
    assert gu.rel_err_scaled(g.nodes["g"].data["energy"].detach(), out["energy"], 1e-3, 1e-3) < 1e-4
    assert gu.rel_err_scaled(g.nodes["n1"].data["gradient"].detach(), out["gradient"], 1e-2, 1e-2) < 1e-4
    assert gu.rel_err(loss.detach(), out["loss"], 1e-6) < 1e-4
    norms = dict(zip(out["grad_norm_keys"].tolist(), out["grad_norm_vals"].tolist()))
    for k, p in model.named_parameters():
        if k in norms and norms[k] > 1e-6:
            assert abs(float(p.grad.norm()) - norms[k]) / norms[k] < 2e-5, k      # measured: 1e-6
