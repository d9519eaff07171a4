"""Collate throughput on the GPU box: host path (get_collate_fn -> batch -> .to(device) -> BatchPlan, what feeds the
engine without a resident dataset) vs DeviceDataset.collate (dataset packed in HBM, one gather launch per batch).
Prints molecules/s for both and the HBM rate of the gather (bytes written + read per batch / kernel time, HIP events).
Usage: python tools/collate_bench.py [--batch 256] [--pool 2048] [--confs 40] [--out-confs 32]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--pool", type=int, default=2048)
    ap.add_argument("--confs", type=int, default=40)
    ap.add_argument("--out-confs", type=int, default=32)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    from grappa_amd.dataloader import get_collate_fn
    from grappa_amd.datasets import graph_from_pool, pool_atom_counts
    from grappa_amd.device_dataset import DeviceDataset
    counts = pool_atom_counts()
    cand = np.nonzero((counts >= 20) & (counts <= 40))[0][: args.pool]
    t0 = time.perf_counter()
    items = [(graph_from_pool(int(i), n_confs=args.confs, seed=0), "ds") for i in cand]
    t_items = time.perf_counter() - t0
    t0 = time.perf_counter()
    ds = DeviceDataset(items, device="cuda")
    torch.cuda.synchronize()
    t_pack = time.perf_counter() - t0
    packed_bytes = sum(t.data.numel() * 4 for t in list(ds.plan_tables.values()) + list(ds.feat.values()))
    rng = np.random.default_rng(0)
    batches = [rng.choice(len(items), size=args.batch, replace=False) for _ in range(args.reps)]
    col = get_collate_fn(conf_strategy=args.out_confs)
    # host path
    torch.manual_seed(0)
    t0 = time.perf_counter()
    for ids in batches[:5]:
        g, _ = col([items[i] for i in ids])
        g = g.to("cuda")
        g.plan()
    torch.cuda.synchronize()
    t_host = (time.perf_counter() - t0) / 5
    # device path
    ds.collate(batches[0], args.out_confs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for ids in batches:
        g, _ = ds.collate(ids, args.out_confs)
    torch.cuda.synchronize()
    t_dev = (time.perf_counter() - t0) / len(batches)
    # gather kernel alone (HIP events on the launch stream around the whole collate; host work overlaps the previous batch's kernel)
    out_bytes = sum(v.numel() * v.element_size() for nt in g.ntypes for v in g.nodes[nt].data.values() if v.dtype != torch.int64)
    p = g.plan()
    out_bytes += sum(t.numel() * 4 for t in [p.indices, p.rev, p.inc_code] + list(p.idx32.values()) + list(p.inv_rows.values())) + 6 * p.N * 4
    be_prof = []
    from grappa_amd.backend import get_backend
    be = get_backend()
    orig = be.collate_gather

    def timed(tables, B):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig(tables, B)
        e1.record()
        be_prof.append((e0, e1))

    be.collate_gather = timed
    for ids in batches:
        ds.collate(ids, args.out_confs)
    torch.cuda.synchronize()
    be.collate_gather = orig
    k_ms = float(np.median([a.elapsed_time(b) for a, b in be_prof]))
    res = {"batch": args.batch, "dataset_molecules": len(items), "confs_in": args.confs, "confs_out": args.out_confs,
           "dataset_pack_s": t_pack, "packed_bytes": packed_bytes, "host_collate_ms": 1e3 * t_host, "host_molecules_per_s": args.batch / t_host,
           "device_collate_ms": 1e3 * t_dev, "device_molecules_per_s": args.batch / t_dev, "gather_kernel_ms": k_ms,
           "batch_bytes": out_bytes, "gather_GBps_read_plus_write": 2 * out_bytes / (k_ms * 1e-3) / 1e9, "build_items_s": t_items}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
