"""GPU box, lab build with -DWL_LAB_STAMP (tools/writer_layer_lab.sh stamp "-DWL_LAB_STAMP"): where a tile of the fused forward kernel spends its cycles.
   GRAPPA_HIP_LIB=tools/lab/libgrappa_hip_stamp.so python tools/writer_layer_stamps.py [s] [save 0/1]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grappa_amd.backend import get_backend      # noqa: E402

BF, F = torch.bfloat16, 512


def main():
    s = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    save = len(sys.argv) > 2 and sys.argv[2] == "1"
    be = get_backend()
    gen = torch.Generator(device="cuda").manual_seed(0)
    rn = lambda *sh: torch.randn(sh, generator=gen, device="cuda")      # noqa: E731
    k = 1.0 / F ** 0.5
    P = [1 + 0.1 * rn(F), 0.1 * rn(F), rn(3 * F, F) * k, 0.1 * rn(3 * F), rn(F, F) * k, 0.1 * rn(F), 1 + 0.1 * rn(F), 0.1 * rn(F), rn(F, F) * k, 0.1 * rn(F),
         rn(F, F) * k, 0.1 * rn(F)]
    TT = 64 // s
    T = TT * 2048
    M = s * T
    x = (rn(M, F) * 1.5).to(BF)
    out = torch.empty_like(x)
    sv = None
    if save:
        sv = dict(mean1=torch.empty(M, device="cuda"), rstd1=torch.empty(M, device="cuda"), meanf=torch.empty(M, device="cuda"), rstdf=torch.empty(M, device="cuda"),
                  x1=torch.empty_like(x), qkv=torch.empty((M, 3 * F), dtype=BF, device="cuda"), att=torch.empty_like(x), x2=torch.empty((be.lib.grappa_writer_head_tiles(s, T) * 64, F), dtype=BF, device="cuda"), x2_tiled=True,
                  x3=torch.empty_like(x), u=torch.empty_like(x))
    for _ in range(3):
        be.writer_layer_fwd(x, s, T, 8, 0.1 if save else 0.0, 11, 12, *P, out, save=sv)
    torch.cuda.synchronize()
    n, words = 2048, 48
    buf = (C.c_ulonglong * (n * words))()
    fn = be.lib.grappa_debug_writer_stamps
    fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int]
    assert fn(buf, n) == 0
    st = np.frombuffer(buf, dtype=np.uint64).reshape(n, words).astype(np.int64)
    names = {(0, 1): "phase 0: load x, LayerNorm -> image", (1, 2): "  barrier"}
    for hp in range(4):
        b = 6 * hp
        names[(2 + b if hp == 0 else 8 + b - 6, 3 + b)] = f"pair {hp}: q|k|v product (16 k-steps x 12 MFMA)"
        names[(3 + b, 4 + b)] = "  its epilogue (bias, bf16, staging) + out-proj weight fetch"
        names[(4 + b, 5 + b)] = "  barrier"
        names[(5 + b, 6 + b)] = "  qkv save copy + attention"
        names[(6 + b, 7 + b)] = "  barrier"
        names[(7 + b, 8 + b)] = "  next weights fetch + out-projection partial (4 k-steps x 16 MFMA)"
    names[(26, 27)] = "phase 2: x2 epilogue, LayerNorm statistics (2 barriers), x3 -> image, barrier"
    names[(27, 28)] = "phase 3: x3 save copy + FF1 product (16 x 16 MFMA)"
    names[(28, 29)] = "  its epilogue (ELU) + barrier"
    names[(29, 30)] = "phase 4: u save copy + FF2 product"
    names[(30, 31)] = "  its epilogue (dropout, residual, stores) until the last store has landed"
    if len(sys.argv) > 3 and sys.argv[3] == "bwd":
        dout = (rn(M, F) * 0.1).to(BF)
        for _ in range(3):
            be.writer_layer_bwd(dout, x, s, T, 8, 0.1, 11, 12, sv, P[0], P[1], P[2], P[4], P[6], P[7], P[8], P[10])
        torch.cuda.synchronize()
        assert fn(buf, n) == 0
        st = np.frombuffer(buf, dtype=np.uint64).reshape(n, words).astype(np.int64)
        names = {(0, 1): "phase 0: dout -> dropout mask -> image, dz2 store", (1, 2): "phase 1: u loads, barrier, dz2 W_2 product", (2, 3): "  its epilogue (ELU', dz1 store)",
                 (3, 4): "phase 2: dout / x2 loads, weight fetch, barrier", (4, 5): "  dz1 W_1 product", (5, 6): "  LayerNorm backward, dropout mask, dzo store (1 barrier)",
                 (6, 7): "phase 3: weight fetch, barrier", (7, 8): "  dzo W_o product", (8, 9): "  its epilogue + barrier"}
        for hp in range(4):
            names[(9 + 3 * hp if hp == 0 else 12 + 3 * (hp - 1), 10 + 3 * hp)] = f"pair {hp}: attention backward (+ dqkv store)" + (" + previous barrier" if hp else "")
            names[(10 + 3 * hp, 11 + 3 * hp)] = "  weight fetch + barrier"
            names[(11 + 3 * hp, 12 + 3 * hp)] = "  dqkv W_in partial (12 k-steps x 16 MFMA)"
        names[(21, 22)] = "  barrier"
        names[(22, 23)] = "phase 5: x loads, LayerNorm backward (1 barrier), dx store, until the last store has landed"
        tot = st[:, 23] - st[:, 0]
        print(f"BACKWARD s = {s}: {n} workgroups, whole tile: median {np.median(tot):.0f} cycles (p10 {np.percentile(tot, 10):.0f}, p90 {np.percentile(tot, 90):.0f})")
        for (a, b), name in names.items():
            med = float(np.median(st[:, b] - st[:, a]))
            print(f"  {med:9.0f} cycles  {100 * med / np.median(tot):5.1f} %  {name}")
        return
    tot = st[:, 31] - st[:, 0]
    print(f"s = {s}, save = {save}: {n} workgroups, whole tile: median {np.median(tot):.0f} cycles (p10 {np.percentile(tot, 10):.0f}, p90 {np.percentile(tot, 90):.0f})")
    acc = {}
    for (a, b), name in names.items():
        dlt = st[:, b] - st[:, a]
        med = float(np.median(dlt))
        print(f"  {med:9.0f} cycles  {100 * med / np.median(tot):5.1f} %  {name}")
        key = name.strip().split(":")[0] if name.startswith("pair") or name.startswith("  ") else name[:7]
        acc[name.strip()[:28]] = acc.get(name.strip()[:28], 0.0) + med
    print("  MFMA floor: 1,536 MFMAs x 16 cycles = 24,576 cycles per wavefront alone on its SIMD, 49,152 with its partner")


if __name__ == "__main__":
    main()
