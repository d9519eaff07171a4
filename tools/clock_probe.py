"""GPU clock under a loop of the default dense product (and, for comparison, under the LayerNorm kernel): the matrix-core peak in
MI355X_MICROARCH.md is quoted at 2.4 GHz; what the chip holds under this load scales the real ceiling (DVFS give-back).
    python tools/clock_probe.py > gpurun_out/clock_probe.txt
The probing side is a child `rocm-smi --showclocks --showpower --showmaxpower` (never touches the GPU context of this process).
With GRAPPA_HIP_LIB=build/variants/libgrappa_hip_bx_<knock-out>.so (tools/gemm_f16x3_check.py --build-variants): clock and socket
power under the kernel with one of its parts removed."""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from grappa_amd.backend import HipBackend  # noqa: E402


def sample(tag, stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showmaxpower"], capture_output=True, text=True, timeout=10).stdout
            lines = [ln.strip() for ln in r.splitlines() if "sclk" in ln or "Power" in ln or "power" in ln]
            out.append((tag, time.time(), " | ".join(lines)))
        except Exception as e:  # noqa: BLE001
            out.append((tag, time.time(), f"rocm-smi failed: {e}"))
        time.sleep(0.5)


def run(tag, fn, seconds, out):
    stop = threading.Event()
    th = threading.Thread(target=sample, args=(tag, stop, out))
    th.start()
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds:
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        n += 50
    stop.set()
    th.join()
    return n / (time.time() - t0)


def main():
    be = HipBackend()
    M, N, K = 83328, 512, 512
    A, B = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
    C = torch.empty(M, N, device="cuda")
    g, b = torch.ones(K, device="cuda"), torch.zeros(K, device="cuda")
    Y, mean, rstd = torch.empty_like(A), torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    out = []
    time.sleep(1.0)
    run("idle", lambda: None, 1.5, out)
    precs = ("f32_f16x3",) if os.environ.get("GRAPPA_HIP_LIB") else ("f32_f16x3", "f32_bf16x6", "f32")      # a knock-out build: its one arithmetic
    for prec in precs:
        rate = run(prec, lambda: be.gemm(A, B, C, M=M, N=N, K=K, precision=prec), 4.0, out)
        print(f"{prec}: {1e3 / rate:.3f} ms per product ({2.0 * M * N * K * rate / 1e12:.1f} TFLOP/s)")
    rate = run("layernorm", lambda: be.layernorm_fwd(A, g, b, Y, mean, rstd, amax=False), 3.0, out)
    print(f"layernorm_fwd: {1e3 / rate:.3f} ms")
    for tag, _, line in out:
        print(f"[{tag}] {line}")


if __name__ == "__main__":
    main()
