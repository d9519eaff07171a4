"""GPU: where the HOST time of one eager train step goes (cProfile over 10 steps of the batch-32 workload): the step is launch-bound there."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from grappa_amd import Energy, MolwiseLoss, get_default_model_config, model_from_config, ops  # noqa: E402
from grappa_amd.datasets import build_batch_from_pool, workload_molecule_ids  # noqa: E402
from grappa_amd.optim import FlatParams, FusedAdam  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    model = model_from_config(get_default_model_config())
    bench.keyed_init(model)
    model = model.to("cuda").train()
    opt = FusedAdam(FlatParams(model), lr=1.5e-5, max_grad_norm=10.0)
    g = build_batch_from_pool(workload_molecule_ids("C2-pubchem-b256", seed=0)[:n], n_confs=32, seed=0).to("cuda")
    loss_fn, energy = MolwiseLoss(**bench.LOSS_KW), Energy()
    ops.manual_seed(1)

    def step():
        opt.zero_grad()
        for lvl in ("n2", "n3", "n4", "n4_improper"):
            for k in ("k", "eq"):
                g.nodes[lvl].data.pop(k, None)
        loss_fn(energy(model(g))).backward()
        opt.step()

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)
    st.sort_stats("cumulative").print_stats(18)


if __name__ == "__main__":
    main()
