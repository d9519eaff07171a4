"""GPU, diagnostic build only (csrc compiled with -DGQ_STAMP=1, see tools/pairs_stamps.sh): the timeline of the one-tile pair-format GEMM
kernel, workgroup by workgroup -- where a tile's time goes (first slab's arrival, main loop, of which waiting for LDS-DMA + barrier, scale-back,
epilogue) and how the grid's phases line up on the chip.
    GRAPPA_HIP_LIB=build/variants/libgrappa_stamp.so python tools/pairs_stamps.py 83328 512 512"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gemm_pairs_check as g  # noqa: E402

dev = "cuda"
WORDS = 16


def main():
    shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:] if "x" in a] or [(83328, 512, 512)]
    gen = torch.Generator(device=dev)
    gen.manual_seed(0)
    lib = g.lib
    il = not (os.environ.get("GRAPPA_PAIRS_IL", "1") == "0")
    read_stamps = lib.grappa_debug_pairs_il_stamps if il else lib.grappa_debug_pairs_stamps
    read_stamps.argtypes = [C.c_void_p, C.c_int]
    for (M, N, K) in shapes:
        A = torch.randn((M, K), generator=gen, device=dev)
        W = torch.randn((N, K), generator=gen, device=dev) * 0.05
        fill = os.environ.get("STAMP_FILL", "random")
        if fill == "zeros":                    # DVFS check: the same launches on operands that toggle nothing
            A.zero_(), W.zero_()
        elif fill == "coarse":                 # values exact in fp16: every LO half is zero
            A, W = A.half().float(), W.half().float()
        am_a, am_b = g.amax(A), g.amax(W)
        ap, bp = g.split_pairs(A, am_a), g.split_pairs(W, am_b)
        bias = torch.randn(N, generator=gen, device=dev)
        out = torch.empty((M, N), device=dev)
        ws = g.ws_for(M, N, K)
        for _ in range(5):
            g.gemm(ap, bp, out, M, N, K, am_a, am_b, True, ws=ws, bias=bias)
        torch.cuda.synchronize()
        t = g.timeit(lambda: g.gemm(ap, bp, out, M, N, K, am_a, am_b, True, ws=ws, bias=bias), n=10)
        g.gemm(ap, bp, out, M, N, K, am_a, am_b, True, ws=ws, bias=bias)
        torch.cuda.synchronize()
        nwg = ((M + 255) // 256) * ((N + 127) // 128)
        buf = np.zeros((min(nwg, 16384), WORDS), dtype=np.uint64)
        rc = read_stamps(buf.ctypes.data, buf.shape[0])
        assert rc == 0
        s = buf.astype(np.int64)
        rt0, t1, t2, t3, t4, t5, rt1, wait = (s[:, i] for i in range(8))
        xcc = s[:, 9]
        start = (rt0 - rt0.min()) * 0.01            # us (100 MHz)
        end = (rt1 - rt0.min()) * 0.01
        clk = (t5 - t1) / np.maximum((rt1 - rt0) * 0.01, 1e-9) / 1e3      # GHz
        print(f"== {'pinned pipeline (gemm_pairs_il)' if il else 'round-3 loop'}, fill {fill} M={M} N={N} K={K}: {nwg} workgroups, kernel {1e3 * t:.1f} us ({2.0 * M * N * K / t / 1e9:.1f} TF); stamped span {end.max():.1f} us; "
              f"in-kernel clock median {np.median(clk):.2f} GHz")
        cyc = lambda a: np.percentile(a, [10, 50, 90])      # noqa: E731
        us = lambda c: c / np.median(clk) / 1e3               # noqa: E731
        rows = [("first slab landed (prologue)", t2 - t1), ("main loop", t3 - t2), ("  of which: waiting for DMA + barrier", wait), ("scale-back + barrier", t4 - t3),
                ("epilogue (to the last store done)", t5 - t4), ("whole workgroup", t5 - t1)]
        for name, arr in rows:
            p10, p50, p90 = cyc(arr)
            print(f"  {name:40s} cycles p10 {p10:9.0f}  p50 {p50:9.0f}  p90 {p90:9.0f}   = {us(p50):6.2f} us")
        nslab = int(s[0, 12])
        print(f"  MFMA floor of the main loop: {nslab} slabs x 24 MFMAs x 32 cycles = {nslab * 768} cycles per wavefront alone on its SIMD, {2 * nslab * 768} with its partner")
        # generations: workgroups ordered by start time
        order = np.argsort(start)
        gens = [order[:512], order[512:1024], order[1024:]]
        for gi, idx in enumerate(gens):
            if len(idx) == 0:
                continue
            print(f"  generation {gi} ({len(idx)} workgroups): start {start[idx].min():6.1f} .. {start[idx].max():6.1f} us (median {np.median(start[idx]):6.1f}), "
                  f"end median {np.median(end[idx]):6.1f}, duration median {np.median(end[idx] - start[idx]):5.1f} us; loop {us(np.median((t3 - t2)[idx])):5.1f}, "
                  f"wait in loop {us(np.median(wait[idx])):5.1f}, epilogue {us(np.median((t5 - t4)[idx])):5.1f}, prologue {us(np.median((t2 - t1)[idx])):5.1f}")
        # how many workgroups are in their epilogue / main loop at each moment (10 us grid)
        ep0 = end - us(t5 - t4)
        lp0 = ep0 - us(t4 - t3) - us(t3 - t2)
        print("  t(us): workgroups resident / in main loop / in epilogue")
        for tt in np.arange(0, end.max(), max(end.max() / 24, 1.0)):
            res = int(((start <= tt) & (end > tt)).sum())
            inl = int(((lp0 <= tt) & (ep0 - us(t4 - t3) > tt)).sum())
            ine = int(((ep0 <= tt) & (end > tt)).sum())
            print(f"   {tt:6.1f}: {res:4d} {inl:4d} {ine:4d}")
        print("  per XCD: workgroups", [int((xcc == x).sum()) for x in range(8)])


if __name__ == "__main__":
    main()
