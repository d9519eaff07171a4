"""GPU (-m gpu): the gradient reducer over a real RCCL communicator.  The box has ONE card, so the process group has one rank: what the
test can and does cover is that RCCL initialises on this image, that the asynchronous all-reduce of the writer-head bucket issued from
INSIDE the backward pass (ops.SplitHeadsFn's node, four head streams) and the second bucket in `finish()` run on RCCL's stream in the
right order behind the kernels that write the buffer, and that waiting on the work objects returns -- the sums over one rank are the
identity, so the gradient must come back equal (up to fp32 summation order) to a pass without any collective.  (Sums over two ranks: tests/test_host_train.py,
gloo.)  Runs in a child process: the process group must not leak into the test runner."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, socket
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl")          # "nccl" is RCCL on ROCm
from test_host_train import TINY, _loss
from grappa_amd import GrappaModel
from grappa_amd.backend import get_backend
from grappa_amd.dist import BucketedGradReducer
from grappa_amd.optim import FlatParams
torch.manual_seed(0)
model = GrappaModel(**TINY).to("cuda").train()
assert model.parameter_writer.head_streams == int(os.environ.get("GRAPPA_HEAD_STREAMS", "4"))      # (the default: four)
flat = FlatParams(model)
be = get_backend()
ids = [30, 31, 32, 33, 34, 35, 36, 37]

import grappa_amd.datasets as ds
_build = ds.build_batch_from_pool
ds.build_batch_from_pool = lambda *a, **k: _build(*a, **k).to("cuda")        # (_loss builds its batch through this name)

def close(a, b):
    # the reducer flushes the queued weight gradients when the heads' bucket leaves, so the grouped launches hold other products than
    # in a pass without it and cut their K elsewhere: fp32 rounding differs, a bucket sent too early would differ by whole gradients
    return float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())

def loss(part):
    return _loss(model, part, 5, global_b=len(ids))

loss(ids).backward(); be.flush_wgrads(); torch.cuda.synchronize()
want = flat.grad.clone()
assert float(want.abs().max()) > 0
BucketedGradReducer._active = staticmethod(lambda: True)          # one rank: make the reducer call the collectives anyway
for chunks in (1, 2):
    assert not BucketedGradReducer(model, flat).overlap           # default since round 4: both buckets after backward()
    red = BucketedGradReducer(model, flat, overlap=True)          # opt-in: the heads' bucket leaves from inside the backward pass
    assert red.overlap
    flat.zero_grad()
    red.begin_step(chunks)
    if chunks == 1:
        loss(ids).backward()
    else:
        loss(ids[:4]).backward(); assert not red._heads_sent
        loss(ids[4:]).backward()
    assert red._heads_sent and len(red._work) == 1
    red.finish()
    torch.cuda.synchronize()
    if chunks == 2:
        got2 = flat.grad.clone()
        model.on_heads_backward_done = None
        flat.zero_grad()
        loss(ids[:4]).backward(); loss(ids[4:]).backward(); be.flush_wgrads(); torch.cuda.synchronize()
        assert close(got2, flat.grad), float((got2 - flat.grad).abs().max())
    else:
        assert close(flat.grad, want), float((flat.grad - want).abs().max())
t = torch.ones(1 << 20, device="cuda")
dist.all_reduce(t); dist.barrier(); torch.cuda.synchronize()
assert float(t.sum()) == float(1 << 20)
dist.destroy_process_group()
print("RCCL_OK", torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else "")
'''


def test_reducer_buckets_over_a_one_rank_rccl_communicator():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", f"ROOT = {ROOT!r}\n" + CHILD], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
