"""GPU: latency of `Grappa.predict(Molecule)` for one small molecule (the reference's everyday call): where the time goes between the
host-side graph preparation, the forward pass (launch-bound at this size) and the extraction of `Parameters`."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import bench  # noqa: E402
from grappa_amd import Grappa, Molecule, Parameters, get_default_model_config, model_from_config  # noqa: E402
from grappa_amd.batch import check_disconnected_graphs  # noqa: E402


def main():
    model = model_from_config(get_default_model_config())
    bench.keyed_init(model)
    gr = Grappa(model, device="cuda")
    torch.manual_seed(0)
    import numpy as np
    from grappa_amd.datasets import molecule_from_pool, pool_atom_counts
    want = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    mol = molecule_from_pool(int(np.argmin(np.abs(pool_atom_counts() - want))))          # the pool molecule nearest to the asked size
    for _ in range(3):
        gr.predict(mol)
    torch.cuda.synchronize()
    n = 20
    t = {"to_dgl": 0.0, "to_device": 0.0, "forward": 0.0, "to_cpu": 0.0, "from_dgl": 0.0}
    t0 = time.perf_counter()
    for _ in range(n):
        a = time.perf_counter()
        g = mol.to_dgl(max_element=gr.max_element, exclude_feats=[])
        check_disconnected_graphs(g)
        b = time.perf_counter()
        g = g.to("cuda")
        torch.cuda.synchronize()
        c = time.perf_counter()
        with torch.no_grad():
            g = gr.model(g)
        torch.cuda.synchronize()
        d = time.perf_counter()
        g = g.to("cpu")
        e = time.perf_counter()
        Parameters.from_dgl(g)
        f = time.perf_counter()
        for k, v in zip(t, (b - a, c - b, d - c, e - d, f - e)):
            t[k] += v
    tot = (time.perf_counter() - t0) / n
    print(f"predict on {mol.to_dgl().num_nodes('n1')} atoms: {1e3 * tot:.2f} ms per call: " + ", ".join(f"{k} {1e3 * v / n:.2f}" for k, v in t.items()))


if __name__ == "__main__":
    main()
