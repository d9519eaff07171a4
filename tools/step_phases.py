"""GPU: where the C2 train step's time goes by PHASE (HIP events on one queue: GRAPPA_HEAD_STREAMS=1, weight gradients not set aside):
GNN forward | writer heads forward | energy + loss | heads backward (incl. energy / loss backward) | GNN backward | optimiser.
Says how much of the step the GNN's 8,233-row products (132 workgroups on 256 CUs) can be blamed for."""
import os
import sys

os.environ["GRAPPA_HEAD_STREAMS"] = "1"
os.environ["GRAPPA_WGRADS_ASIDE"] = "0"
import torch  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from grappa_amd import Energy, MolwiseLoss, get_default_model_config, model_from_config, ops  # noqa: E402
from grappa_amd.datasets import build_workload  # noqa: E402
from grappa_amd.optim import FlatParams, FusedAdam  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C2-pubchem-b256"
    model = model_from_config(get_default_model_config())
    bench.keyed_init(model)
    model = model.to("cuda").train()
    flat = FlatParams(model)
    opt = FusedAdam(flat, lr=1.5e-5, max_grad_norm=10.0)
    g = build_workload(name, seed=0).to("cuda")
    loss_fn = MolwiseLoss(**bench.LOSS_KW)
    energy = Energy()
    ops.manual_seed(1)
    ev = {}

    def mark(k):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        ev[k] = e

    from grappa_amd.backend import get_backend
    be = get_backend()

    def heads_done():
        be.flush_wgrads()          # the heads' queued weight gradients belong to the heads' phase
        mark("heads_bwd")
    model.on_heads_backward_done = heads_done

    def step():
        opt.zero_grad()
        for lvl in ("n2", "n3", "n4", "n4_improper"):
            for k in ("k", "eq"):
                g.nodes[lvl].data.pop(k, None)
        mark("start")
        g.plan()
        model.gnn(g)
        mark("gnn_fwd")
        h = g.nodes["n1"].data["h"]
        h.register_hook(lambda grad: (heads_done(), None)[1])
        model.parameter_writer(g)
        mark("heads_fwd")
        loss = loss_fn(energy(g))
        mark("loss")
        loss.backward()
        be.flush_wgrads()
        mark("bwd")
        opt.step()
        mark("opt")

    model.on_heads_backward_done = None
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    keys = ["start", "gnn_fwd", "heads_fwd", "loss", "heads_bwd", "bwd", "opt"]
    acc = {k: 0.0 for k in keys[1:]}
    n = 10
    for _ in range(n):
        step()
        torch.cuda.synchronize()
        for a, b in zip(keys[:-1], keys[1:]):
            acc[b] += ev[a].elapsed_time(ev[b])
    tot = sum(acc.values()) / n
    print(f"{name}: atoms {g.plan().N}; one queue; {tot:.2f} ms per step")
    names = {"gnn_fwd": "GNN forward", "heads_fwd": "writer heads forward", "loss": "energy + loss forward", "heads_bwd": "loss / energy / heads backward (+ their weight gradients)",
             "bwd": "GNN backward (+ its weight gradients)", "opt": "clip + Adam"}
    for k in keys[1:]:
        print(f"  {names[k]:62s} {acc[k] / n:7.2f} ms  {100 * acc[k] / n / tot:5.1f} %")


if __name__ == "__main__":
    main()
