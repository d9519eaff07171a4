"""GPU: randomised shapes x operand formats x forced plans x epilogues for the fp32-grade dense products, each against a float64 product
(error bound of the arithmetic, 3e-6 of the row's largest magnitude): the all-pairs kernels (three tiles), fp32 A + weight pairs, the
fp32-operand split kernel.  A forced plan the planner refuses (status != 0) is skipped and counted.
    python tools/gemm_fuzz.py [n_cases] [seed]          # tests/test_gpu_pairs.py runs 48 cases of seed 0"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gemm_pairs_check as g  # noqa: E402

dev = "cuda"


def one_case(rng, gen):
    kind = rng.choice(["pairs", "pairs", "wpairs", "wpairs", "split"])
    M = int(np.exp(rng.uniform(np.log(1), np.log(60000))))
    N = int(rng.choice([32, 64, 96, 128, 160, 256, 384, 512, 516, 768, 1536, 2048]))
    K = int(rng.choice([16, 32, 48, 64, 80, 96, 112, 128, 160, 192, 256, 272, 400, 512, 1104, 1536, 2048]))
    if kind != "split":              # (the pair formats are operands of the split arithmetic's kernels: M, N > 32; below that the native fp32 kernel, fp32 operands only)
        M, N = max(M, 33), max(N, 64)
    if M * max(N, K) > 48_000_000:
        M = 48_000_000 // max(N, K)
    dgrad = bool(rng.integers(0, 2)) and kind != "split"
    A = torch.randn((M, K), generator=gen, device=dev) * float(np.exp(rng.uniform(-6, 6)))
    if rng.integers(0, 3) == 0:
        A = A * torch.exp2(torch.randint(-30, 30, (M, 1), generator=gen, device=dev).float())
    W = torch.randn((K, N) if dgrad else (N, K), generator=gen, device=dev) * 0.05
    am_a, am_b = g.amax(A), g.amax(W, rows=not dgrad)
    ref = A.double() @ (W.double() if dgrad else W.double().t())
    kw, epi = {}, []
    if rng.integers(0, 2):
        kw["bias"] = torch.randn(N, generator=gen, device=dev)
        ref = ref + kw["bias"].double()
        epi.append("b")
    if rng.integers(0, 3) == 0:
        kw["act"] = 1
        ref = torch.nn.functional.elu(ref)
        epi.append("e")
    if rng.integers(0, 2):
        kw["res"] = torch.randn((M, N), generator=gen, device=dev)
        ref = ref + kw["res"].double()
        epi.append("r")
    plan = {}
    ns = int(rng.choice([0, 0, 1, 2, 3, 5, 8]))
    if ns:
        plan["plan_nsplit"] = ns
    if kind == "pairs" and rng.integers(0, 2):
        plan["plan_cfg"] = int(rng.choice([6, 7, 8])) + 1
    plan["plan_tail"] = int(rng.choice([0, 1, 2]))
    out = torch.full((M, N), float("nan"), device=dev)
    g.EXTRA = plan
    try:
        if kind == "pairs":
            g.gemm(g.split_pairs(A, am_a), g.split_pairs(W, am_b, transpose=dgrad), out, M, N, K, am_a, am_b, True, **kw)
        elif kind == "wpairs":
            g.gemm(A, g.split_pairs(W, am_b, transpose=dgrad), out, M, N, K, am_a, am_b, "b", **kw)
        else:
            g.gemm(A, W, out, M, N, K, am_a, am_b, False, **kw)
    except AssertionError:
        return None, (kind, M, N, K, dgrad, "".join(epi), plan)
    finally:
        g.EXTRA = {}
    torch.cuda.synchronize()
    scale = ref.abs().amax(dim=1, keepdim=True).clamp_min(1e-300)
    err = float(((out.double() - ref).abs() / scale).max()) if M else 0.0
    if not np.isfinite(err):
        err = float("inf")
    return err, (kind, M, N, K, dgrad, "".join(epi), plan)


def run(n_cases, seed, verbose=True):
    rng = np.random.default_rng(seed)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    worst, refused, bad = 0.0, 0, []
    for i in range(n_cases):
        err, what = one_case(rng, gen)
        if err is None:
            refused += 1
            if verbose:
                print(f"  case {i:4d} {what}: refused", flush=True)
            continue
        worst = max(worst, err)
        if err >= 3e-6:
            bad.append((err, what))
        if verbose and (i % 25 == 0 or err >= 3e-6):
            print(f"  case {i:4d} {what}: err {err:.2e}", flush=True)
    return worst, refused, bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    worst, refused, bad = run(n, seed)
    print(f"{n} cases of seed {seed}: worst error {worst:.2e}, {refused} forced plans refused, {len(bad)} over the bound")
    for b in bad:
        print("  OVER:", b)
    sys.exit(1 if bad else 0)
