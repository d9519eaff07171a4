"""GPU: (1) the recorded (hipGraph) batch-32 train step, replayed N times -- run under `rocprofv3 --kernel-trace --stats` for its kernel times;
(2) where the time of a recorded `Grappa.predict` call goes (host graph preparation, copies into the captured tensors, replay, copy back,
Parameters.from_dgl).   python tools/recorded_profile.py train|predict"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from grappa_amd import Energy, Grappa, MolwiseLoss, Parameters, get_default_model_config, model_from_config, ops  # noqa: E402
from grappa_amd.batch import check_disconnected_graphs  # noqa: E402
from grappa_amd.capture import CapturedForward, CapturedTrainStep  # noqa: E402
from grappa_amd.datasets import build_batch_from_pool, molecule_from_pool, pool_atom_counts, workload_molecule_ids  # noqa: E402
from grappa_amd.optim import FlatParams, FusedAdam  # noqa: E402


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "train"
    model = model_from_config(get_default_model_config())
    bench.keyed_init(model)
    if what == "train":
        n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
        model = model.to("cuda").train()
        opt = FusedAdam(FlatParams(model), lr=1.5e-5, max_grad_norm=10.0)
        g = build_batch_from_pool(workload_molecule_ids("C2-pubchem-b256", seed=0)[:n], n_confs=32, seed=0).to("cuda")
        ops.manual_seed(1)
        step = CapturedTrainStep(model, Energy(), MolwiseLoss(**bench.LOSS_KW), opt, g)
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            step()
        torch.cuda.synchronize()
        print(f"recorded train step, {n} molecules: {1e3 * (time.perf_counter() - t0) / 40:.2f} ms per step", flush=True)
        return
    gr = Grappa(model, device="cuda")
    mol = molecule_from_pool(int(np.argmin(np.abs(pool_atom_counts() - 40))))
    for _ in range(4):
        gr.predict(mol)
    ent = next(iter(gr._graphs.entries.values()))
    t = {"to_dgl": 0.0, "plan+tables(host)": 0.0, "load(copies)": 0.0, "replay+sync": 0.0, "to_cpu": 0.0, "from_dgl": 0.0}
    n = 30
    for _ in range(n):
        torch.cuda.synchronize()
        a = time.perf_counter()
        g = mol.to_dgl(max_element=gr.max_element, exclude_feats=[])
        check_disconnected_graphs(g)
        b = time.perf_counter()
        plan = g.plan()
        for lvl in ent.plan.__dict__.get("_pos_tables", {}):
            plan.position_tables(lvl)
        c = time.perf_counter()
        ent.load(g)
        d = time.perf_counter()
        ent.replay()
        torch.cuda.synchronize()
        e = time.perf_counter()
        ent.read_outputs(g)
        f = time.perf_counter()
        Parameters.from_dgl(g)
        h = time.perf_counter()
        for key, v in zip(t, (b - a, c - b, d - c, e - d, f - e, h - f)):
            t[key] += v
    print("recorded predict, ms per call: " + ", ".join(f"{k} {1e3 * v / n:.3f}" for k, v in t.items()) + f"; total {1e3 * sum(t.values()) / n:.3f}", flush=True)


if __name__ == "__main__":
    main()
