"""GPU: how long the HOST needs to enqueue one C2 train step (no synchronisation inside) against the step's GPU time: the room there is
for more, smaller launches (e.g. two half batches in flight) before the step becomes host-bound."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from grappa_amd import Energy, MolwiseLoss, get_default_model_config, model_from_config, ops  # noqa: E402
from grappa_amd.datasets import build_workload  # noqa: E402
from grappa_amd.optim import FlatParams, FusedAdam  # noqa: E402


def main():
    model = model_from_config(get_default_model_config())
    bench.keyed_init(model)
    model = model.to("cuda").train()
    flat = FlatParams(model)
    opt = FusedAdam(flat, lr=1.5e-5, max_grad_norm=10.0)
    g = build_workload("C2-pubchem-b256", seed=0).to("cuda")
    loss_fn = MolwiseLoss(**bench.LOSS_KW)
    energy = Energy()
    ops.manual_seed(1)

    def step():
        opt.zero_grad()
        for lvl in ("n2", "n3", "n4", "n4_improper"):
            for k in ("k", "eq"):
                g.nodes[lvl].data.pop(k, None)
        t0 = time.perf_counter()
        out = energy(model(g))
        loss = loss_fn(out)
        t1 = time.perf_counter()
        loss.backward()
        t2 = time.perf_counter()
        opt.step()
        t3 = time.perf_counter()
        return t1 - t0, t2 - t1, t3 - t2

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    for streams in (4, 1):
        model.parameter_writer.head_streams = streams
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        fw = bw = op = 0.0
        n = 20
        t_all = time.perf_counter()
        for _ in range(n):
            torch.cuda.synchronize()
            a, b, c = step()
            fw, bw, op = fw + a, bw + b, op + c
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t_all) / n
        print(f"head streams {streams}: host enqueue per step: forward+loss {1e3 * fw / n:.1f} ms, backward {1e3 * bw / n:.1f} ms, optimiser {1e3 * op / n:.2f} ms; "
              f"step wall (synchronised every step) {1e3 * wall:.1f} ms", flush=True)


if __name__ == "__main__":
    main()
