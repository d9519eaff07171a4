"""FastEvaluator against the metrics the reference's own FastEvaluator produced on two batches
(tests/golden/ref_eval.npz, written by oracle/make_goldens.py eval): one batch carries padded dummy conformations, three
dataset names interleave.  CPU: host logic through the test-only backend; GPU (-m gpu): the grappa_eval_se_f32 kernel."""
import types

import numpy as np
import pytest
import torch

import golden_utils as gu
from grappa_amd.evaluation import FastEvaluator, early_stopping_loss

RTOL = 1e-5   # fp32 sums in a different order than the reference's per-molecule torch.sum


class _Graph:
    """the slice of MolBatch the evaluator touches: plan() (B, N, atom_molptr) and the 'g' / 'n1' data dicts"""

    def __init__(self, fx, bi, device):
        t = lambda k: torch.from_numpy(fx[f"b{bi}::{k}"].copy()).to(device)   # noqa: E731
        counts = fx[f"b{bi}::atoms_per_mol"]
        ptr = torch.from_numpy(np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)).to(device)
        self._plan = types.SimpleNamespace(B=len(counts), N=int(counts.sum()), atom_molptr=ptr, device=device)
        self.nodes = {"g": types.SimpleNamespace(data={"energy": t("energy"), "energy_ref": t("energy_ref"), "is_dummy": t("is_dummy"),
                                                       "energy_classical_ff": t("energy_classical_ff")}),
                      "n1": types.SimpleNamespace(data={"gradient": t("gradient"), "gradient_ref": t("gradient_ref"),
                                                        "gradient_classical_ff": t("gradient_classical_ff")})}

    def plan(self):
        return self._plan


def _check(fx, device):
    for tag, ev in (("full", FastEvaluator()), ("nograd", FastEvaluator(gradients=False)), ("classical", FastEvaluator(log_classical_values=True))):
        for bi in range(int(fx["n_batches"][0])):
            ev.step(_Graph(fx, bi, device), [str(x) for x in fx[f"b{bi}::dsnames"]])
        m = ev.pool()
        keys = [k for k in fx.files if k.startswith(f"metrics::{tag}::")]
        assert len(keys) >= 8
        for k in keys:
            _, _, ds, name = k.split("::")
            want = float(fx[k][0])
            got = m[ds][name]
            if np.isnan(want):
                assert got is None, k
            else:
                assert abs(got - want) <= RTOL * abs(want), (k, got, want)
        assert set(m) == {"dsA", "dsB", "dsC", "avg"}
        assert ev._acc is None and ev._ds_index == {}          # pool() resets the storage (evaluation.py:155)
        if tag == "full":
            assert abs(early_stopping_loss(m, 2.0) - (2.0 * m["avg"]["rmse_energies"] + m["avg"]["rmse_gradients"])) < 1e-9


def test_fast_evaluator_matches_reference_cpu(ref_backend):
    _check(gu.load("ref_eval.npz"), "cpu")


@pytest.mark.gpu
def test_fast_evaluator_matches_reference_gpu():
    _check(gu.load("ref_eval.npz"), "cuda")


@pytest.mark.gpu
def test_eval_kernel_matches_oracle_on_a_large_batch():
    """256 molecules x 32 conformations with random dummy padding: kernel vs the oracle's restatement, per molecule"""
    from grappa_amd.backend import HipBackend
    from oracle.ops_ref import RefBackend
    g = torch.Generator().manual_seed(5)
    B, Cc = 256, 32
    counts = torch.randint(3, 90, (B,), generator=g)
    ptr = torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(0)]).int()
    N = int(counts.sum())
    e, er = torch.randn(B, Cc, generator=g) * 30, torch.randn(B, Cc, generator=g) * 30
    gr, grr = torch.randn(N, Cc, 3, generator=g) * 10, torch.randn(N, Cc, 3, generator=g) * 10
    nreal = torch.randint(1, Cc + 1, (B,), generator=g)
    dummy = (torch.arange(Cc)[None, :] >= nreal[:, None]).float()
    plan = types.SimpleNamespace(B=B, N=N, atom_molptr=ptr)
    want = torch.empty(B, 4)
    RefBackend().eval_se(plan, e, er, dummy, gr, grr, want)
    plan_d = types.SimpleNamespace(B=B, N=N, atom_molptr=ptr.cuda())
    got = torch.empty(B, 4, device="cuda")
    HipBackend().eval_se(plan_d, e.cuda(), er.cuda(), dummy.cuda(), gr.cuda(), grr.cuda(), got)
    torch.cuda.synchronize()
    assert torch.equal(got[:, 1].cpu(), want[:, 1]) and torch.equal(got[:, 3].cpu(), want[:, 3])      # counts: exact
    assert torch.allclose(got[:, 0].cpu(), want[:, 0], rtol=2e-5, atol=1e-3)
    assert torch.allclose(got[:, 2].cpu(), want[:, 2], rtol=2e-5, atol=1e-3)
