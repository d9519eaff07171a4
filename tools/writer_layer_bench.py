"""GPU box: the fused writer-head layer (grappa_writer_head_fwd) alone, at the C3 batch's three big token tables; HIP events around 5 launches.
   python tools/writer_layer_bench.py [tag]      (GRAPPA_HIP_LIB=... selects a lab build of the library, tools/writer_layer_lab.sh)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grappa_amd.backend import get_backend      # noqa: E402

BF = torch.bfloat16
F = 512


def main():
    be = get_backend()
    gen = torch.Generator(device="cuda").manual_seed(0)
    rn = lambda *s: torch.randn(s, generator=gen, device="cuda")      # noqa: E731
    k = 1.0 / F ** 0.5
    P = [1 + 0.1 * rn(F), 0.1 * rn(F), rn(3 * F, F) * k, 0.1 * rn(3 * F), rn(F, F) * k, 0.1 * rn(F), 1 + 0.1 * rn(F), 0.1 * rn(F), rn(F, F) * k, 0.1 * rn(F),
         rn(F, F) * k, 0.1 * rn(F)]
    rows = []
    for s, T in ((4, 116000), (3, 70000), (2, 40000), (4, 4000)):
        M = s * T
        x = (rn(M, F) * 1.5).to(BF)
        out = torch.empty_like(x)
        sv = dict(mean1=torch.empty(M, device="cuda"), rstd1=torch.empty(M, device="cuda"), meanf=torch.empty(M, device="cuda"), rstdf=torch.empty(M, device="cuda"),
                  x1=torch.empty_like(x), qkv=torch.empty((M, 3 * F), dtype=BF, device="cuda"), att=torch.empty_like(x), x2=torch.empty((be.lib.grappa_writer_head_tiles(s, T) * 64, F), dtype=BF, device="cuda"), x2_tiled=True,
                  x3=torch.empty_like(x), u=torch.empty_like(x))
        for save, p in ((None, 0.0), (sv, 0.0), (sv, 0.1)):
            for _ in range(2):
                be.writer_layer_fwd(x, s, T, 8, p, 11, 12, *P, out, save=save)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                be.writer_layer_fwd(x, s, T, 8, p, 11, 12, *P, out, save=save)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            fl = 2.0 * M * F * 6 * F
            rows.append(f"s={s} T={T:6d} save={'yes' if save else 'no '} drop={p}: {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s  {ms * 1e3 / ((T + 64 // s - 1) // (64 // s) / 256):7.1f} us per round of 256 tiles")
        # the backward chain (grappa_writer_head_bwd) over what the last forward saved
        dout = (rn(M, F) * 0.1).to(BF)
        for _ in range(2):
            be.writer_layer_bwd(dout, x, s, T, 8, 0.1, 11, 12, sv, P[0], P[1], P[2], P[4], P[6], P[7], P[8], P[10])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            be.writer_layer_bwd(dout, x, s, T, 8, 0.1, 11, 12, sv, P[0], P[1], P[2], P[4], P[6], P[7], P[8], P[10])
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        rows.append(f"s={s} T={T:6d} BACKWARD drop=0.1      : {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s  {ms * 1e3 / ((T + 64 // s - 1) // (64 // s) / 256):7.1f} us per round of 256 tiles  (incl. the two small LayerNorm-partial reductions)")
    print("\n".join(rows))
    tag = sys.argv[1] if len(sys.argv) > 1 else "run"
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/writer_layer_bench_{tag}.txt", "a") as f:
        f.write(f"# lib={os.environ.get('GRAPPA_HIP_LIB', 'default')}\n" + "\n".join(rows) + "\n")


if __name__ == "__main__":
    main()
