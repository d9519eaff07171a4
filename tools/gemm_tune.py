"""Compare the GEMM launcher's cost-model plan with an exhaustive search over (tile config, split-K, tail) per workload shape:
   python tools/gemm_tune.py [precision]     (default f32_bf16x6)
Uses grappa_gemm_desc.plan_cfg / plan_nsplit / plan_tail (the tuning fields of the C ABI, per call since ABI 10).  Prints, per shape, the model's plan and time, the best
forced plan and time, and the step-weighted total of both."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _plan(lib, M, N, K, prec, v, cfg=-1, ns=0, tail=-1):
    """the plan the library would choose under a forced configuration (C ABI 10: the force is a field of the product's descriptor)"""
    from grappa_amd import _lib as _L
    d = _L.GemmDesc()
    d.M, d.N, d.K, d.precision = M, N, K, prec
    d.plan_cfg, d.plan_nsplit, d.plan_tail = (cfg + 1 if cfg >= 0 else 0), max(ns, 0), (2 if tail == 0 else 3 if tail == 1 else 0)
    return lib.grappa_gemm_f32_plan_desc(C.byref(d), *[C.byref(x) for x in v])


def timeit(be, A, B, Cm, M, N, K, ak, bk, mode, reps=8):
    for _ in range(2):
        be.gemm(A, B, Cm, M=M, N=N, K=K, a_kcontig=bool(ak), b_kcontig=bool(bk), precision=mode)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        be.gemm(A, B, Cm, M=M, N=N, K=K, a_kcontig=bool(ak), b_kcontig=bool(bk), precision=mode)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    import gemm_shapes_bench as gsb
    from grappa_amd import _lib
    from grappa_amd.backend import get_backend
    mode = sys.argv[1] if len(sys.argv) > 1 else "f32_bf16x6"
    prec = _lib.GEMM_PRECISIONS[mode]
    shapes = gsb.record_shapes()
    be = get_backend()
    lib = be.lib
    cfgs = [5, 6] if prec != 0 else [0, 1, 4]
    tot_model = tot_best = 0.0
    rows = []
    for (M, N, K, ak, bk, lda, ldb, ldc), cnt in sorted(shapes.items(), key=lambda kv: -kv[0][0] * kv[0][1] * kv[0][2] * kv[1]):
        if M <= 32 or N <= 32:
            continue
        A = torch.randn((M, K) if ak else (K, M), device="cuda")
        B = torch.randn((N, K) if bk else (K, N), device="cuda")
        Cm = torch.empty((M, N), device="cuda")
        be.plan_override = None
        v = [C.c_int() for _ in range(5)]
        lib.grappa_gemm_f32_plan(M, N, K, prec, *[C.byref(x) for x in v])
        model_plan = tuple(x.value for x in v)
        t_model = timeit(be, A, B, Cm, M, N, K, ak, bk, mode)
        best = (t_model, "model")
        max_split = max(1, K // 32)            # (GRAPPA_PLAN_MIN_KSTEPS=1, the default since round 4: K ranges down to one slab)
        splits = sorted({s for s in (1, 2, 3, 4, 5, 7, 9, 12, 15, 19, 24, 30, 38, 48, 60) if s <= max_split})
        for cfg in cfgs:
            for ns in splits:
                for tail in ((0, 1) if ns == 1 else (0,)):
                    be.plan_override = (cfg, ns, tail)
                    _plan(lib, M, N, K, prec, v, cfg, ns, tail)
                    if v[2].value != ns or (v[0].value, v[1].value) != {5: (128, 128), 6: (256, 128), 0: (128, 128), 1: (64, 64), 4: (128, 64)}[cfg]:
                        continue            # the launcher cannot realise this forced plan (split too fine for K, ...)
                    if tail == 1 and v[4].value == 0:
                        continue
                    try:
                        t = timeit(be, A, B, Cm, M, N, K, ak, bk, mode, reps=5)
                    except Exception as e:      # workspace too small for a forced plan etc.
                        continue
                    if t < best[0]:
                        best = (t, f"cfg{cfg} ns{ns} tail{tail}")
        be.plan_override = None
        tot_model += cnt * t_model
        tot_best += cnt * best[0]
        rows.append((cnt * (t_model - best[0]), M, N, K, ak, bk, cnt, model_plan, t_model, best))
    for gain, M, N, K, ak, bk, cnt, mp, tm, best in sorted(rows, key=lambda r: -r[0]):
        print(f"{M:7d} {N:5d} {K:7d} {ak} {bk} cnt {cnt:3d}  model tile {mp[0]}x{mp[1]} ns {mp[2]} tail {mp[3]}x{mp[4]}: {tm:.3f} ms | best {best[1]}: {best[0]:.3f} ms | step gain {gain:.3f} ms")
    print(f"total model {tot_model:.2f} ms/step, exhaustive best {tot_best:.2f} ms/step")


if __name__ == "__main__":
    main()
