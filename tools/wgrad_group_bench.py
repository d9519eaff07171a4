"""GPU: the weight gradients of one transformer layer of the four writer heads (C2 token counts), launched one by one
(grappa_gemm_f32, each with its own split-K + reduce) and as one group (grappa_gemm_f32_grouped); GRAPPA_GROUP_KPS forces the K chunk."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from grappa_amd.backend import get_backend  # noqa: E402

be = get_backend()
gen = torch.Generator(device="cuda").manual_seed(0)
tokens = [17158, 44325, 83328, 28248]
probs = []
for T in tokens:
    for Np, Kp in ((1536, 512), (512, 512), (512, 512), (512, 512)):
        probs.append((torch.randn((T, Np), generator=gen, device="cuda"), torch.randn((T, Kp), generator=gen, device="cuda"),
                      torch.zeros((Np, Kp), device="cuda"), torch.zeros((Np,), device="cuda")))
flops = sum(2.0 * dz.shape[0] * dz.shape[1] * x.shape[1] for dz, x, _, _ in probs)


def run(defer):
    be.defer_wgrads = defer
    for dz, x, dw, db in probs:
        if defer:
            am = None
            if be.wants_amax(True):          # the fp16-split arithmetic: row maxima of both operands ride along (as gemm_wgrad queues them)
                am = (be.amax(dz, None, rows=True), be.amax(x, None, rows=True))
            be._wq.append((dz, x, dw, db, am))
        else:
            be.gemm_wgrad(dz, x, dw, db)
    if defer:
        be._launch_wgrad_group()


for defer in (False, True):
    for _ in range(2):
        run(defer)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run(defer)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{'grouped' if defer else 'single '} kps={os.environ.get('GRAPPA_GROUP_KPS', 'auto'):>6s}: {ms:7.3f} ms  {flops / ms / 1e9:6.1f} TF", flush=True)
