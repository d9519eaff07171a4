"""GPU: cProfile of `Grappa.predict` on one 40-atom molecule through the cache of recorded forwards (host side of a 3 ms call)."""
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import golden_utils as gu
    from grappa_amd import Grappa, get_default_model_config, model_from_config
    from grappa_amd.datasets import molecule_from_pool, pool_atom_counts
    model = model_from_config(get_default_model_config())
    model.load_state_dict(gu.keyed_state_dict(model))
    gr = Grappa(model, device="cuda")
    mol = molecule_from_pool(int(np.argmin(np.abs(pool_atom_counts() - 40))))
    for _ in range(5):
        gr.predict(mol)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        gr.predict(mol)
    dt = (time.perf_counter() - t0) / 200
    print(f"predict: {dt * 1e3:.3f} ms per call")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200):
        gr.predict(mol)
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30)
    print(s.getvalue()[:6000])
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(30)
    print(s.getvalue()[:6000])


if __name__ == "__main__":
    main()
