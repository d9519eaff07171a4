"""which dense-product operands still get a maxima pass of their own (precision f32_f16x3): one C2 train step, passes grouped by the
ops.py line that asked for them.   python tools/amax_passes.py > gpurun_out/amax_passes.txt"""
import collections
import os
import sys

os.environ.setdefault("GRAPPA_GEMM_PRECISION", "f32_f16x3")
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from grappa_amd import Energy, MolwiseLoss, get_default_model_config, model_from_config, ops  # noqa: E402
from grappa_amd import backend as B  # noqa: E402
from grappa_amd.datasets import WORKLOADS, build_batch_from_pool, workload_molecule_ids  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    name = "C2-pubchem-b256"
    model = model_from_config(get_default_model_config())
    bench.keyed_init(model)
    model = model.to(dev).train()
    energy, loss_fn = Energy(), MolwiseLoss(**bench.LOSS_KW)
    ops.manual_seed(1234)
    ids = workload_molecule_ids(name)[:256]
    g = build_batch_from_pool(ids, n_confs=WORKLOADS[name][3], seed=0).to(dev)

    def step():
        for lvl in ("n2", "n3", "n4", "n4_improper"):
            for k in ("k", "eq"):
                g.nodes[lvl].data.pop(k, None)
        model.zero_grad()
        loss_fn(energy(model(g))).backward()
        B.get_backend().flush_wgrads()

    step()
    torch.cuda.synchronize()
    B._AMAX_LOG = []
    step()
    torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for R, C, rows, cols, line in B._AMAX_LOG:
        a = agg[(line, C, rows, cols)]
        a[0] += 1
        a[1] += R * C * 4 / 1e6
    tot = sum(v[1] for v in agg.values())
    print(f"# {len(B._AMAX_LOG)} passes, {tot:.0f} MB read")
    for (line, C, rows, cols), (n, mb) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"ops.py:{line:4d}  width {C:5d} rows={int(rows)} cols={int(cols)}  x{n:3d}  {mb:9.1f} MB")


if __name__ == "__main__":
    main()
