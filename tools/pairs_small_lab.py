"""GPU: tile configuration x split-K sweep of the pair-format GEMM (plan override) and of the fp32-operand fp16-split kernel on the products
that do not fill the chip at C2 (8,233 atom rows, 17,158 / 14,124 token rows)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gemm_pairs_check as g  # noqa: E402

dev = "cuda"
lib = g.lib


def main():
    gen = torch.Generator(device=dev)
    gen.manual_seed(0)
    shapes = [(8233, 512, 512), (8233, 2048, 512), (8233, 512, 2048), (17158, 512, 512), (17158, 1536, 512), (17158, 512, 1536), (14124, 256, 256), (28248, 512, 512)]
    for (M, N, K) in shapes:
        A = torch.randn((M, K), generator=gen, device=dev)
        W = torch.randn((N, K), generator=gen, device=dev) * 0.05
        am_a, am_b = g.amax(A), g.amax(W)
        ap, bp = g.split_pairs(A, am_a), g.split_pairs(W, am_b)
        bias = torch.randn(N, generator=gen, device=dev)
        out = torch.empty((M, N), device=dev)
        ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
        res = []
        for pairs in (True, False):
            for cfg in ((6, 7, 8) if pairs else (5, 6)):
                for ns in (1, 2, 4):
                    if K // ns < 128:
                        continue
                    g.EXTRA = {"plan_cfg": cfg + 1, "plan_nsplit": ns, "plan_tail": 2}
                    try:
                        t = g.timeit(lambda: g.gemm(ap if pairs else A, bp if pairs else W, out, M, N, K, am_a, am_b, pairs, ws=ws, bias=bias), n=20)
                    except AssertionError as e:
                        continue
                    res.append((t, "pairs" if pairs else "f32  ", cfg, ns))
            g.EXTRA = {}
            t = g.timeit(lambda: g.gemm(ap if pairs else A, bp if pairs else W, out, M, N, K, am_a, am_b, pairs, ws=ws, bias=bias), n=20)
            res.append((t, "pairs" if pairs else "f32  ", -1, 0))
        print(f"M={M} N={N} K={K}:")
        for t, kind, cfg, ns in sorted(res):
            tiles = {5: (128, 128), 6: (256, 128), 7: (256, 256), 8: (128, 128), -1: (0, 0)}[cfg]
            nt = ((M + tiles[0] - 1) // tiles[0]) * ((N + tiles[1] - 1) // tiles[1]) * max(ns, 1) if cfg >= 0 else 0
            print(f"   {kind} cfg {cfg:2d} ({tiles[0]:3d}x{tiles[1]:3d}) nsplit {ns}  wgs {nt:5d}: {1e3 * t:7.1f} us  {2.0 * M * N * K / t / 1e9:6.1f} TF" + ("   <- the plan's own choice" if cfg < 0 else ""), flush=True)


if __name__ == "__main__":
    main()
