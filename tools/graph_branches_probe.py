"""GPU: does a hipGraph replay run independent chains side by side?  N chains of 100 dependent small products (1,062 x 512 x 512: ~16 us each,
36 workgroups: they would fit on the chip together several times over), each chain on a stream of its own, captured in ONE graph (fork /
join on the capturing stream) -- against the same chains replayed as N graphs on N streams, and launched eagerly on N streams.
    python tools/graph_branches_probe.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from grappa_amd.backend import get_backend
    be = get_backend()
    M, N, K, L = 1062, 512, 512, 100
    W = torch.randn(N, K, device="cuda") * 0.05
    bias = torch.zeros(N, device="cuda")

    def chain(x, y):
        # L dependent products ping-ponging between two buffers
        a, b = x, y
        for _ in range(L):
            be.gemm(a, W, b, M=M, N=N, K=K, bias=bias)
            a, b = b, a

    def bufs(n):
        return [(torch.randn(M, K, device="cuda"), torch.empty(M, N, device="cuda")) for _ in range(n)]

    main_s = torch.cuda.Stream()
    for n in (1, 2, 4):
        streams = [torch.cuda.Stream() for _ in range(n)]
        bs = bufs(n)
        with torch.cuda.stream(main_s):
            for s, (x, y) in zip(streams, bs):           # warm-up on the very streams (workspaces are keyed by stream)
                s.wait_stream(main_s)
                with torch.cuda.stream(s):
                    chain(x, y)
                main_s.wait_stream(s)
            main_s.synchronize()
            # one graph, n branches
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=main_s):
                for s, (x, y) in zip(streams, bs):
                    s.wait_stream(main_s)
                    with torch.cuda.stream(s):
                        chain(x, y)
                for s in streams:
                    main_s.wait_stream(s)
            g.replay()
            main_s.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main_s)
            for _ in range(5):
                g.replay()
            e1.record(main_s)
            main_s.synchronize()
            t_one = e0.elapsed_time(e1) / 5
        # n graphs on n streams
        graphs = []
        for s, (x, y) in zip(streams, bs):
            gi = torch.cuda.CUDAGraph()
            with torch.cuda.stream(s):
                with torch.cuda.graph(gi, stream=s):
                    chain(x, y)
            graphs.append(gi)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            for s, gi in zip(streams, graphs):
                with torch.cuda.stream(s):
                    gi.replay()
            torch.cuda.synchronize()
        t_many = (time.perf_counter() - t0) * 1e3 / 5
        # eager on n streams
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s, (x, y) in zip(streams, bs):
            with torch.cuda.stream(s):
                chain(x, y)
        torch.cuda.synchronize()
        t_eager = (time.perf_counter() - t0) * 1e3
        print(f"{n} chains x {L} products: one graph with {n} branches {t_one:6.2f} ms | {n} graphs on {n} streams {t_many:6.2f} ms | eager on {n} streams "
              f"{t_eager:6.2f} ms   (one chain alone = the n = 1 row)")


if __name__ == "__main__":
    main()
