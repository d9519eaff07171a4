"""GPU: recorded training on real epochs at batch 32 (the bench extra `b32_train_epochs` on its own, with a per-phase breakdown of the host's
share of a step): production model, 1,024 resident molecules of the C2 range x 32 conformations, shuffled batches of 32.
    python tools/b32_epochs.py [n_epochs] [n_buckets]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from grappa_amd import get_default_model_config, model_from_config, ops  # noqa: E402
from grappa_amd.datasets import WORKLOADS, graph_from_pool, select_molecules  # noqa: E402
from grappa_amd.device_dataset import DeviceDataset  # noqa: E402
from grappa_amd.trainer import Trainer  # noqa: E402


def main():
    n_epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    n_buckets = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    n_mols = int(os.environ.get("N_MOLS", "1024"))
    torch.manual_seed(0)
    ops.manual_seed(1)
    ids = select_molecules(n_mols, seed=11, min_atoms=WORKLOADS["C2-pubchem-b256"][1], max_atoms=WORKLOADS["C2-pubchem-b256"][2])
    t0 = time.perf_counter()
    items = [(graph_from_pool(int(i), n_confs=32, seed=0), "pool") for i in ids]
    ds = DeviceDataset(items, device="cuda")
    print(f"dataset of {n_mols} molecules built in {time.perf_counter() - t0:.1f} s", flush=True)
    for mode in ("eager", "recorded"):
        model = model_from_config(get_default_model_config()).to("cuda").train()
        tr = Trainer(model, ds, None, batch_size=32, conf_strategy=32, lr=1e-5, gradient_clip_val=10.0, start_qm_epochs=0, warmup_steps=2,
                     energy_weight=1.0, gradient_weight=0.8, param_weight=0.0, recorded=(mode == "recorded"), shape_buckets=n_buckets, seed=3)
        for e in range(n_epochs if mode == "recorded" else 2):
            torch.cuda.synchronize()
            t = time.perf_counter()
            loss = tr.train_epoch(e)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
            nb = (n_mols + 31) // 32
            print(f"{mode:8s} epoch {e}: {dt:6.2f} s = {1e3 * dt / nb:6.2f} ms/step = {n_mols / dt:7.0f} mol/s   loss {loss:.4f}" +
                  (f"   {tr.recorded_stats}" if mode == "recorded" else ""), flush=True)
        if mode == "recorded":
            print("bucket caps:", tr._buckets.caps)
            # the host's share of a recorded step, phase by phase (synchronised: not the overlapped cost)
            from grappa_amd.trainer import epoch_batches
            batches = [b for b in epoch_batches(ds.names, 32, generator=torch.Generator().manual_seed(9)) if len(b) == 32][:20]
            ph = {"collate+pad": 0.0, "signature+lookup": 0.0, "load": 0.0, "replay": 0.0}
            from grappa_amd.capture import train_signature
            for b in batches:
                tot = ds.totals(b)
                caps = tr._buckets.choose(tot)
                if caps is None:
                    continue
                torch.cuda.synchronize(); t = time.perf_counter()
                g, names = ds.collate(b, 32, pad_to=caps)
                g.plan().param_weight_rows = tr.loss_fn.param_weights_of(list(names), g.plan().B).to("cuda")
                torch.cuda.synchronize(); ph["collate+pad"] += time.perf_counter() - t; t = time.perf_counter()
                key = (train_signature(g), tr._step_stamp())
                step = tr._steps.get(key)
                torch.cuda.synchronize(); ph["signature+lookup"] += time.perf_counter() - t; t = time.perf_counter()
                if step is None:
                    continue
                step.load(g)
                torch.cuda.synchronize(); ph["load"] += time.perf_counter() - t; t = time.perf_counter()
                step()
                torch.cuda.synchronize(); ph["replay"] += time.perf_counter() - t
            print("per step, synchronised phases (ms):", {k: round(1e3 * v / len(batches), 3) for k, v in ph.items()}, flush=True)
            # the same phases WITHOUT synchronising: the host's own time per phase -- a phase that takes about a replay's time waits for the GPU
            hp = {"totals+choose": 0.0, "write_pad": 0.0, "collate": 0.0, "pw": 0.0, "signature": 0.0, "load": 0.0, "replay": 0.0}
            torch.cuda.synchronize()
            t_all = time.perf_counter()
            for b in batches:
                t = time.perf_counter()
                tot = ds.totals(b)
                caps = tr._buckets.choose(tot)
                hp["totals+choose"] += time.perf_counter() - t; t = time.perf_counter()
                ds._write_pad(ds.pad_sizes(tot, caps))
                hp["write_pad"] += time.perf_counter() - t; t = time.perf_counter()
                g, names = ds.collate(b, 32, pad_to=caps)
                hp["collate"] += time.perf_counter() - t; t = time.perf_counter()
                g.plan().param_weight_rows = tr.loss_fn.param_weights_of(list(names), g.plan().B).pin_memory().to("cuda", non_blocking=True)
                hp["pw"] += time.perf_counter() - t; t = time.perf_counter()
                step = tr._steps.get((train_signature(g), tr._step_stamp()))
                hp["signature"] += time.perf_counter() - t; t = time.perf_counter()
                step.load(g)
                hp["load"] += time.perf_counter() - t; t = time.perf_counter()
                step()
                hp["replay"] += time.perf_counter() - t
            t_host = time.perf_counter() - t_all
            torch.cuda.synchronize()
            t_tot = time.perf_counter() - t_all
            print("per step, host time per phase, unsynchronised (ms; collate includes a second write_pad):", {k: round(1e3 * v / len(batches), 3) for k, v in hp.items()},
                  f"host loop {1e3 * t_host / len(batches):.2f} ms/step, with the final sync {1e3 * t_tot / len(batches):.2f} ms/step", flush=True)
        del tr, model
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
