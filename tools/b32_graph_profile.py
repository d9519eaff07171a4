#!/usr/bin/env python3
"""tools/b32_graph_profile.py [replays] -- the recorded b32 train step (32 molecules x 32 conformations, bench.py's `b32_train`) replayed
`replays` times; run under `rocprofv3 --kernel-trace --stats` to see which kernels a replay consists of (tools/kstats.py on the summary).
Prints the wall time per replay."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grappa_amd import Energy, MolwiseLoss, get_default_model_config, model_from_config  # noqa: E402
from grappa_amd.capture import CapturedTrainStep  # noqa: E402
from grappa_amd.datasets import build_batch_from_pool, workload_molecule_ids  # noqa: E402
from grappa_amd.optim import FlatParams, FusedAdam  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
torch.manual_seed(0)
model = model_from_config(get_default_model_config()).to("cuda").train()
flat = FlatParams(model)
opt = FusedAdam(flat, lr=1e-5)
ids = workload_molecule_ids("C2-pubchem-b256", seed=0)[:32]
g = build_batch_from_pool(ids, n_confs=32, seed=0).to("cuda")
loss_fn = MolwiseLoss(gradient_weight=0.8, energy_weight=1.0, param_weight=0.0, proper_regularisation=1e-3)
cap = CapturedTrainStep(model, Energy(), loss_fn, opt, g, warmup=3)
for _ in range(5):
    cap()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    cap()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"b32 recorded step: {1e3 * dt / n:.3f} ms per replay over {n} replays; loss {float(cap.loss):.4f}")
