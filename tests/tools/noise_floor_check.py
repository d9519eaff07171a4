"""How far are the GPU and the fp32 oracle from a float64 run of the oracle, on the batch of
tests/test_gpu_e2e.py::test_production_config_against_oracle_on_a_larger_batch?  Prints, per quantity, the test's error metric for
GPU vs fp32 oracle (what the test asserts), GPU vs float64 oracle and fp32 oracle vs float64 oracle.
    [GRAPPA_HIP_LIB=...] python tests/tools/noise_floor_check.py
(lives under tests/: it runs the oracle as the checker, which only tests, smoke() and bench.py's cpu_baseline may do)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_utils as gu  # noqa: E402
from grappa_amd import Energy, MolwiseLoss, get_default_model_config, model_from_config  # noqa: E402
from grappa_amd.datasets import build_batch_from_pool  # noqa: E402
from oracle import cpu_ref  # noqa: E402

cfg = get_default_model_config()
ids = list(range(500, 532))
lk = dict(gradient_weight=0.8, energy_weight=1.0, param_weight=0.0, proper_regularisation=1e-3)
model = model_from_config(cfg)
sd = gu.keyed_state_dict(model)
model.load_state_dict(sd)
model = model.to("cuda").eval()
g = Energy()(model(build_batch_from_pool(ids, n_confs=8, seed=3).to("cuda")))


def oracle(double):
    m = cpu_ref.RefGrappaModel(**cfg)
    m.load_state_dict(sd)
    m.eval()
    gc = build_batch_from_pool(ids, n_confs=8, seed=3)
    if double:
        m = m.double()
        for nt in gc.ntypes:
            for k, v in list(gc.nodes[nt].data.items()):
                if torch.is_tensor(v) and v.dtype == torch.float32:
                    gc.nodes[nt].data[k] = v.double()
    with torch.no_grad():
        return cpu_ref.RefEnergy()(m(gc))


r32 = oracle(False)
try:
    r64 = oracle(True)
except RuntimeError as ex:          # the restatement builds its input features in fp32
    print('float64 oracle not available:', str(ex)[:80])
    r64 = r32


def get(gr, what, lvl=None):
    t = gr.nodes["g"].data[f"energy_{lvl}"] if what == "energy_lvl" else (gr.nodes[lvl].data["k"] if what == "k" else gr.nodes["g"].data["energy"])
    return t.detach().cpu().double().numpy()


for lvl in ("n2", "n3", "n4", "n4_improper"):
    for what, ff, fa in (("energy_lvl", 1e-2, 1e-3), ("k", None, None)):
        a, b32, b64 = get(g, what, lvl), get(r32, what, lvl), get(r64, what, lvl)
        if what == "k":
            fl = 0.05 if lvl.startswith("n4") else 1e-3
            e = lambda x, y: gu.rel_err(x, y, fl)      # noqa: E731
        else:
            e = lambda x, y: gu.rel_err_scaled(x, y, ff, fa)      # noqa: E731
        print(f"{lvl:12s} {what:10s} GPU vs fp32 oracle {e(a, b32):.3e} | GPU vs float64 {e(a, b64):.3e} | fp32 oracle vs float64 {e(b32, b64):.3e}")
