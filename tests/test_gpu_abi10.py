"""C ABI 10: no mutable process-wide state in libgrappa_hip.so -- plan override, tail launches, split-K reduction and dropout salt travel with
every call ("no global state, re-entrant per stream").  Two host threads enqueue the SAME product with DIFFERENT options on two
streams at the same time; each gets the bits of its own options."""
import ctypes as C
import os
import sys
import threading

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.gpu


def _desc(gp, a, b, out, M, N, K, am_a, am_b, **opts):
    from grappa_amd import _lib
    d = _lib.GemmDesc()
    d.M, d.N, d.K, d.a_kcontig, d.b_kcontig = M, N, K, 1, 1
    d.A, d.lda, d.B, d.ldb, d.C, d.ldc = a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0)
    d.a_amax, d.b_amax, d.precision = am_a.data_ptr(), am_b.data_ptr(), gp.F16X3
    for k, v in opts.items():
        setattr(d, k, v)
    return d


def test_two_threads_two_streams_two_option_sets():
    import gemm_pairs_check as gp
    lib = gp.lib
    gen = torch.Generator(device="cuda").manual_seed(3)
    M, N, K = 50000, 512, 512                      # 196 x 4 tiles: a tail of 16 tiles when tails are allowed
    a = torch.randn((M, K), generator=gen, device="cuda")
    w = torch.randn((N, K), generator=gen, device="cuda") * 0.05
    am_a, am_b = gp.amax(a), gp.amax(w)
    options = [dict(plan_tail=1), dict(plan_tail=2), dict(plan_nsplit=2, splitk_reduce=1), dict(plan_nsplit=2, splitk_reduce=2)]

    def run_once(opts, stream, out, ws):
        d = _desc(gp, a, w, out, M, N, K, am_a, am_b, **opts)
        rc = lib.grappa_gemm_f32(stream.cuda_stream, C.byref(d), ws.data_ptr(), ws.numel())
        assert rc == 0, rc

    def ws_for(opts):
        d = _desc(gp, a, w, a, M, N, K, am_a, am_b, **opts)
        return torch.empty(max(lib.grappa_gemm_f32_workspace_bytes_desc(C.byref(d)), 16), dtype=torch.uint8, device="cuda")

    # what each option set gives when it runs alone
    want = []
    for opts in options:
        out = torch.empty((M, N), device="cuda")
        run_once(opts, torch.cuda.current_stream(), out, ws_for(opts))
        torch.cuda.synchronize()
        want.append(out)
    assert not torch.equal(want[0], want[1]) and not torch.equal(want[0], want[2])      # the options DO change the summation order
    assert torch.equal(want[2], want[3])                                                # (where the K slices are summed does not)

    errors = []

    def worker(i, j):
        try:
            st = torch.cuda.Stream()
            outs = [torch.empty((M, N), device="cuda") for _ in range(2)]
            wss = [ws_for(options[i]), ws_for(options[j])]
            for rep in range(20):
                for k, o in enumerate((i, j)):
                    run_once(options[o], st, outs[k], wss[k])
                st.synchronize()
                for k, o in enumerate((i, j)):
                    if not torch.equal(outs[k], want[o]):
                        errors.append((i, j, rep, o))
                        return
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(0, 2)), threading.Thread(target=worker, args=(1, 3))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_dropout_salt_travels_with_the_call():
    """the same masked product with and without a salt word, interleaved: each call draws the masks of ITS argument"""
    import gemm_pairs_check as gp
    lib = gp.lib
    gen = torch.Generator(device="cuda").manual_seed(4)
    M, N, K = 2000, 512, 512
    a = torch.randn((M, K), generator=gen, device="cuda")
    w = torch.randn((N, K), generator=gen, device="cuda") * 0.05
    am_a, am_b = gp.amax(a), gp.amax(w)
    salt = torch.full((1,), 5, dtype=torch.int64, device="cuda")
    ws = gp.ws_for(M, N, K)
    outs = []
    for s in (None, salt.data_ptr(), None, salt.data_ptr()):
        out = torch.empty((M, N), device="cuda")
        d = _desc(gp, a, w, out, M, N, K, am_a, am_b, drop_p=0.5, drop_seed=99, drop_salt=s)
        assert lib.grappa_gemm_f32(gp.stream(), C.byref(d), ws.data_ptr(), ws.numel()) == 0
        outs.append(out)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[2]) and torch.equal(outs[1], outs[3]) and not torch.equal(outs[0], outs[1])
    zeros = [(o == 0).float().mean().item() for o in outs[:2]]
    assert all(0.45 < z < 0.55 for z in zeros)
