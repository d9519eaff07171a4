"""Repeated eval-mode forwards of the production model on the C2 batch: single-stream reference vs the writer heads on four HIP streams,
with NaN-filled temporaries and after torch.cuda.empty_cache() (fresh device allocations).  Prints which tuple levels differ.
Used to characterise the multi-stream discrepancy that keeps GRAPPA_HEAD_STREAMS at 1 by default."""
import sys, os, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import golden_utils as gu
from grappa_amd import get_default_model_config, model_from_config, ops
from grappa_amd.datasets import build_workload
model = model_from_config(get_default_model_config())
model.load_state_dict(gu.keyed_state_dict(model))
model = model.to("cuda").eval()
g_cpu = build_workload("C2-pubchem-b256", seed=0)
def run():
    with torch.no_grad():
        g = model(g_cpu.to("cuda"))
    torch.cuda.synchronize()
    return {lvl: g.nodes[lvl].data["k"].clone() for lvl in ("n2", "n3", "n4", "n4_improper")}
def cmp(tag, out, ref):
    msg = []
    for lvl in out:
        nan = int(torch.isnan(out[lvl]).sum())
        d = (out[lvl] - ref[lvl]).abs()
        rows = int((torch.nan_to_num(d, nan=1.0).reshape(d.shape[0], -1).max(1).values > 0).sum())
        if rows or nan:
            msg.append(f"{lvl}: {rows} rows differ, {nan} NaN")
    print(tag, "OK" if not msg else "; ".join(msg))
model.parameter_writer.head_streams = 1
ref = run(); cmp("single again", run(), ref)
# NaN-filled fresh buffers in SINGLE-stream mode: does any kernel read memory it did not write?
orig_new = ops._new
ops._new = lambda shape, like: torch.full(shape, float("nan"), dtype=torch.float32, device=like.device)
cmp("single, NaN-filled temporaries", run(), ref)
model.parameter_writer.head_streams = 4
cmp("multi first, NaN-filled temporaries", run(), ref)
cmp("multi second, NaN-filled", run(), ref)
ops._new = orig_new
torch.cuda.empty_cache()
cmp("multi after empty_cache", run(), ref)
cmp("multi next", run(), ref)
torch.cuda.empty_cache()
cmp("multi after empty_cache 2", run(), ref)
