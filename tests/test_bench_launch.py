"""CPU: `python bench.py --gpus 2` starts its own ranks (one process per GPU in production; here two gloo ranks on the CPU) and prints ONE
JSON line carrying the all-reduce time, the same-workload single-rank reference and the scaling factor.  bench.py has no CPU compute path:
the test-only backend (oracle/ops_ref.py) is injected into the child interpreters from HERE, through a sitecustomize module on PYTHONPATH."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SITE = '''
import os, sys
if os.environ.get("GRAPPA_TEST_REF_BACKEND") == "1":
    sys.path.insert(0, {root!r})
    from grappa_amd import backend
    from oracle.ops_ref import RefBackend
    backend.set_backend(RefBackend())
'''


def test_bench_launches_its_own_ranks_and_reports_the_scaling_factor(tmp_path):
    (tmp_path / "sitecustomize.py").write_text(SITE.format(root=ROOT))
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([str(tmp_path), ROOT, os.environ.get("PYTHONPATH", "")]), GRAPPA_TEST_REF_BACKEND="1",
               OMP_NUM_THREADS="2", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--device", "cpu", "--tiny-model", "--steps", "2",
           "--warmup", "1", "--strong-global-batch", "8", "--chunk", "3", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "strong" and rec["steps"] == 2 and rec["warmup"] == 1
    assert rec["config"]["global_batch"] == 8 and rec["config"]["world_size"] == 2 and rec["config"]["dist_backend"] == "gloo"
    assert rec["config"]["molecules_rank0"] == 4 and rec["config"]["chunks_rank0"] == 2          # 8 molecules dealt to 2 ranks, chunks of <= 3
    assert rec["config"]["allreduce_ms_per_step"] is not None and rec["config"]["allreduce_bytes"] > 0
    ref = rec["strong_scaling_reference"]
    assert ref["n_gpus"] == 1 and ref["config"]["molecules_rank0"] == 8 and ref["value"] > 0
    assert abs(rec["scaling_factor"] - rec["value"] / ref["value"]) < 1e-9
    assert rec["value"] > 0 and rec["ms_per_step"] > 0


def test_bench_eight_gloo_ranks_check_themselves(tmp_path):
    """the N = 8 launch the driver performs on an 8-GPU node, rehearsed here with eight gloo ranks on the CPU (tiny model): the line carries what the
    collective layer saw (world size, backend, one entry per rank) and the bit-for-bit comparison of the overlapped and the post-backward
    gradient reduction"""
    (tmp_path / "sitecustomize.py").write_text(SITE.format(root=ROOT))
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([str(tmp_path), ROOT, os.environ.get("PYTHONPATH", "")]), GRAPPA_TEST_REF_BACKEND="1",
               OMP_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dist-backend", "gloo", "--device", "cpu", "--tiny-model", "--steps", "2",
           "--warmup", "1", "--strong-global-batch", "16", "--chunk", "2", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["scaling"] == "strong"
    assert rec["config"]["world_size"] == 8 and rec["config"]["dist_backend"] == "gloo" and rec["config"]["molecules_rank0"] == 2
    d = rec["dist"]
    assert d["world_size"] == 8 and d["backend"] == "gloo" and sorted(r["rank"] for r in d["ranks"]) == list(range(8))
    assert len({r["pid"] for r in d["ranks"]}) == 8 and "distinct_devices" in d
    assert d["allreduce_bit_check"]["mismatching_steps_summed_over_ranks"] == 0
    assert rec["strong_scaling_reference"]["config"]["molecules_rank0"] == 16 and rec["scaling_factor"] > 0


def test_bench_refuses_a_scaling_line_when_ranks_share_a_device():
    """N ranks over RCCL on fewer than N distinct devices must not produce a `scaling` line"""
    sys.path.insert(0, ROOT)
    import bench
    two_on_one = [dict(rank=0, device_index=0, device_uuid="a"), dict(rank=1, device_index=0, device_uuid="a")]
    assert bench.distinct_devices(two_on_one, 2, enforce=False) == 1          # (gloo rehearsals on the CPU: reported, not enforced)
    with pytest.raises(SystemExit):
        bench.distinct_devices(two_on_one, 2, enforce=True)
    assert bench.distinct_devices([dict(rank=0, device_index=0, device_uuid="a"), dict(rank=1, device_index=1, device_uuid="b")], 2, enforce=True) == 2


def test_bench_refuses_a_step_that_ends_in_a_non_finite_loss():
    """a NaN step is not a measurement (and runs faster than a real one on this chip): bench.py dies instead of printing a line"""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.finite_loss("C2", 90.75) == 90.75
    for bad in (float("nan"), float("inf"), -float("inf")):
        with pytest.raises(SystemExit):
            bench.finite_loss("C2", bad)
