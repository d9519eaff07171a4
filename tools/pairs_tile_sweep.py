"""GPU: which tile serves which product -- the pair-format GEMM (both operands pairs) under each of its three tiles (plan override, one launch, no split-K,
no tail launch) beside the plan's own choice, over the row counts of the C2 / C3 levels x the layer shapes.  One line per shape; the calibration data of
gemm_f32.hip pairs_tile_for()."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gemm_pairs_check as g  # noqa: E402

dev = "cuda"


def main():
    gen = torch.Generator(device=dev)
    gen.manual_seed(0)
    rows = [4096, 8233, 14124, 17158, 20000, 24000, 28248, 32932, 36000, 40000, 44325, 56000, 66000, 83328, 100000]
    nk = [(512, 512), (1536, 512), (512, 1536), (2048, 512), (512, 2048), (256, 256)]
    print("#     M     N     K   tiles6/256 |  cfg6 256x128  cfg7 256x256  cfg8 128x128 |  plan's own (us) | best")
    for (N, K) in nk:
        for M in rows:
            A = torch.randn((M, K), generator=gen, device=dev)
            W = torch.randn((N, K), generator=gen, device=dev) * 0.05
            am_a, am_b = g.amax(A), g.amax(W)
            ap, bp = g.split_pairs(A, am_a), g.split_pairs(W, am_b)
            bias = torch.randn(N, generator=gen, device=dev)
            out = torch.empty((M, N), device=dev)
            ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
            t = {}
            for cfg in (6, 7, 8):
                g.EXTRA = {"plan_cfg": cfg + 1, "plan_nsplit": 1, "plan_tail": 2}
                try:
                    t[cfg] = 1e3 * g.timeit(lambda: g.gemm(ap, bp, out, M, N, K, am_a, am_b, True, ws=ws, bias=bias), n=20)
                except AssertionError:
                    t[cfg] = float("nan")
            g.EXTRA = {}
            own = 1e3 * g.timeit(lambda: g.gemm(ap, bp, out, M, N, K, am_a, am_b, True, ws=ws, bias=bias), n=20)
            best = min((v, k) for k, v in t.items() if v == v)
            tiles6 = ((M + 255) // 256) * ((N + 127) // 128)
            print(f"  {M:6d} {N:5d} {K:5d}   {tiles6 / 256.0:6.2f}    | {t[6]:9.1f}    {t[7]:9.1f}    {t[8]:9.1f}    | {own:9.1f}        | cfg{best[1]} {own / best[0]:.2f}x", flush=True)
            del A, W, ap, bp, out, ws


if __name__ == "__main__":
    main()
