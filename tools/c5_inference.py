"""GPU: BASELINE.json configs[4] / SURVEY 8(d): single-graph inference latency on ONE 50,046-atom protein graph (19 x T4 lysozyme), the
production model in eval mode under no_grad, default (fp32-grade) arithmetic.  `rocprofv3 --kernel-trace --stats -- python3
tools/c5_inference.py` gives the kernel table of the same call."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from grappa_amd import get_default_model_config, model_from_config  # noqa: E402
from grappa_amd.datasets import protein_graph_t4  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    model = model_from_config(get_default_model_config())
    bench.keyed_init(model)
    model = model.to("cuda").eval()
    g = protein_graph_t4(19).to("cuda")
    tup = {l: g.num_nodes(l) for l in ["n2", "n3", "n4", "n4_improper"]}
    with torch.no_grad():
        for _ in range(3):
            model(g)
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            model(g)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
    ts.sort()
    print(f"C5: {g.num_nodes('n1')} atoms, tuples {tup}: forward median {ts[len(ts) // 2]:.2f} ms, min {ts[0]:.2f}, max {ts[-1]:.2f} over {reps}; "
          f"peak memory {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB")


if __name__ == "__main__":
    main()
