#!/usr/bin/env python3
"""tools/c5_profile.py [forwards] -- the C5 inference forward (ONE graph of 19 all-atom T4 lysozymes, 50,046 atoms, eval mode, no_grad: bench.py's
`c5_inference`) repeated; run under `rocprofv3 --kernel-trace --stats` to see what a forward consists of (tools/kstats.py).  Prints the median wall time."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grappa_amd import get_default_model_config, model_from_config  # noqa: E402
from grappa_amd.datasets import protein_graph_t4  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
torch.manual_seed(0)
model = model_from_config(get_default_model_config()).to("cuda").eval()
g = protein_graph_t4(19).to("cuda")
lat = []
with torch.no_grad():
    for i in range(2 + n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        model(g)
        e1.record()
        torch.cuda.synchronize()
        if i >= 2:
            lat.append(e0.elapsed_time(e1))
lat.sort()
print(f"C5 forward: median {lat[len(lat) // 2]:.2f} ms over {n} (min {lat[0]:.2f}, max {lat[-1]:.2f}); tuples "
      + str({lv: int(g.num_nodes(lv)) for lv in ('n2', 'n3', 'n4', 'n4_improper')}))
