"""GPU: the grouped weight-gradient kernel on the workload's shapes with fp32 and pair-format operands (C ABI 8), each format combination
alone (its specialised kernel) and all four in one launch (the mixed kernel): ms per launch by HIP events."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from grappa_amd.backend import get_backend  # noqa: E402


BIAS = True          # with the bias gradient (column sums of dz) as in the train step


def main():
    be = get_backend()
    torch.manual_seed(0)
    T = 83328
    shapes = [(512, 512), (1536, 512), (512, 512), (512, 512)]
    ops = []
    for Np, Kp in shapes:
        dz, x = torch.randn(T, Np, device="cuda"), torch.randn(T, Kp, device="cuda")
        ops.append((dz, x, be.to_pairs(dz), be.to_pairs(x), torch.zeros(Np, Kp, device="cuda"), torch.zeros(Np, device="cuda")))

    def items(fmt):
        out = []
        for i, (dz, x, rz, rx, dw, db) in enumerate(ops):
            pz, px = fmt[i]
            am = (rz if pz else be.amax(dz, None, rows=True), rx if px else be.amax(x, None, rows=True))
            out.append((None if pz else dz, None if px else x, dw, db if BIAS else None, am, rz.pairs if pz else None, rx.pairs if px else None))
        return out

    def timed(fmt, n=10):
        its = items(fmt)
        for _ in range(3):
            be._launch_wgrad_group(its)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            be._launch_wgrad_group(its)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    flops = sum(2.0 * T * a * b for a, b in shapes)
    for name, fmt in (("fp32 / fp32", [(0, 0)] * 4), ("pairs / fp32", [(1, 0)] * 4), ("fp32 / pairs", [(0, 1)] * 4), ("pairs / pairs", [(1, 1)] * 4),
                      ("mixed (one launch)", [(1, 0), (0, 1), (1, 1), (0, 0)])):
        ms = timed(fmt)
        print(f"{name:22s} {ms:7.3f} ms per grouped launch of 4 products = {flops / ms / 1e9:6.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
