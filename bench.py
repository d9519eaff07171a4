"""bench.py -- molecules/sec of one full Grappa train step (GrappaModel -> Energy with forces -> MolwiseLoss ->
backward -> all-reduce -> fused Adam with grad clipping) on N MI355X of one node.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[1]): 256 molecules of 20-40 atoms PER GPU drawn from the Espaloma
molecule pool, 32 conformations, production model (40.8 M parameters, random init), fp32, train mode (dropout on),
synthetic charges / coordinates / reference energies+forces.  Weak scaling: every rank gets its own 256 molecules,
gradients are summed with a two-bucket RCCL all-reduce of the flat gradient buffer after backward() (GRAPPA_OVERLAP_ALLREDUCE=1 sends
the writer-head bucket from inside the backward pass).
The JSON line also carries `roofline` (the GEMM family -- by default fp32 products as six bf16 MFMAs, `gemm_bf16x_kernel`:
algorithmic 2MNK FLOPs / HIP-event time per call, measured in an instrumented single-stream repetition of the same steps right
after the timed region; plus the GAT kernels vs HBM), `gemm_arithmetic` (the same steps with the native fp32 MFMA, and with the
optional reduced backward arithmetic -- neither is ever `value`) and `cpu_baseline` (the oracle's CPU restatement of the same
train step on a bounded sample, rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 16 * 157.3   # same table: the bf16 MFMA runs 16x the fp32 one (~2.5 PFLOP/s dense)
# dominant kernel per GEMM arithmetic: (kernel, bf16 MFMAs issued per fp32 multiply-add column, i.e. partial products)
GEMM_KERNELS = {"f32": ("gemm_f32_kernel<*> (v_mfma_f32_32x32x2_f32)", 0),
                "f32_bf16x9": ("gemm_bf16x_kernel<9,*> (fp32 operands as 3 bf16 pieces, 9 partial products, v_mfma_f32_32x32x16_bf16)", 9),
                "f32_bf16x6": ("gemm_bf16x_kernel<6,*> (fp32 operands as 3 bf16 pieces, 6 partial products, v_mfma_f32_32x32x16_bf16)", 6),
                "bf16x3": ("gemm_bf16x_kernel<3,*>", 3), "bf16": ("gemm_bf16x_kernel<1,*>", 1)}
PEAK_HBM_GBS = 8000.0             # HBM3E spec


def keyed_init(model):
    """deterministic non-trivial weights (same scheme as the parity tests)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_utils as gu
    model.load_state_dict(gu.keyed_state_dict(model))


def cpu_baseline_child(workload: str, n_mols: int, steps: int, threads: int):
    """the oracle (CPU restatement, kind 'port') timed on the host cores on a bounded sample of the same workload; runs in
    a child process of its own (`bench.py --cpu-baseline-child ...`) that never touches the GPU"""
    from grappa_amd import get_default_model_config
    from grappa_amd.datasets import build_batch_from_pool, workload_molecule_ids
    from oracle import cpu_ref
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_utils as gu
    torch.set_num_threads(threads)
    ids = workload_molecule_ids(workload, seed=0)[:n_mols]
    model = cpu_ref.RefGrappaModel(**get_default_model_config())
    model.load_state_dict(gu.keyed_state_dict(model))
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1.5e-5)
    loss_fn = cpu_ref.RefMolwiseLoss(gradient_weight=0.8, energy_weight=1.0, param_weight=0.0, proper_regularisation=1e-3)
    energy = cpu_ref.RefEnergy()
    times = []
    for it in range(steps + 1):
        g = build_batch_from_pool(ids, n_confs=32, seed=0)
        t0 = time.perf_counter()
        opt.zero_grad()
        loss = loss_fn(energy(model(g)))
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
        opt.step()
        if it > 0:
            times.append(time.perf_counter() - t0)
    t = sorted(times)[len(times) // 2]
    print(json.dumps({"value": n_mols / t, "unit": "molecules/s", "cores": int(torch.get_num_threads()), "kind": "port",
                      "sample": f"{n_mols} molecules of {workload} x 32 conformations, production model fp32, median of {steps} train "
                                f"step(s) after one warm-up (oracle/cpu_ref.py, torch {torch.__version__} CPU, {threads} threads)"}), flush=True)


def cpu_baseline(workload: str, n_mols: int, steps: int, limit_s: float):
    """run the bounded CPU sample in a child process (started before this process touches the GPU) under a wall-clock limit"""
    import subprocess
    threads = max(1, min(16, os.cpu_count() or 1))       # more threads than this only oversubscribe the oracle's small per-op work
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", workload, str(n_mols), str(steps), str(threads)]
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(threads))
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=limit_s, env=env)
        for line in reversed(out.stdout.splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"value": None, "unit": "molecules/s", "cores": threads, "kind": "port", "sample": f"child failed: {out.stderr[-300:]}"}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "molecules/s", "cores": threads, "kind": "port",
                "sample": f"{n_mols} molecules x {steps + 1} train steps did not finish within {limit_s:.0f} s on {threads} threads"}


def log(*a):
    print(f"[bench {time.strftime('%H:%M:%S')}]", *a, file=sys.stderr, flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-baseline-child":
        cpu_baseline_child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="C2-pubchem-b256")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gemm-precision", default=None, help="arithmetic of the dense products (default: the backend's, f32_bf16x6)")
    ap.add_argument("--alt-precision", default="f32", help="also time K steps with this GEMM arithmetic (reported beside the default); '' to skip")
    ap.add_argument("--bwd-precision", default="bf16x3", help="also time K steps with the backward-pass products in this arithmetic (reported beside the default, never as `value`); '' to skip")
    ap.add_argument("--strong-global-batch", type=int, default=0, help="strong-scaling run: this many molecules in total, dealt to the ranks (default 0 = weak scaling, the workload's batch per GPU)")
    ap.add_argument("--cpu-sample", type=int, default=8, help="molecules in the CPU baseline sample")
    ap.add_argument("--cpu-limit", type=float, default=150.0, help="wall-clock limit of the CPU baseline child, seconds")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL over xGMI; default) | gloo (functional test of the N>1 path on one GPU)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # CPU baseline first, in a child process, before this process initialises the GPU (rank 0 at N = 1 only)
    cpu_base = None
    if world == 1 and args.gpus == 1 and not args.no_cpu_baseline:
        log(f"cpu baseline (oracle, {args.cpu_sample} molecules, limit {args.cpu_limit:.0f} s) ...")
        cpu_base = cpu_baseline(args.workload, args.cpu_sample, 1, args.cpu_limit)
        log(f"cpu baseline: {cpu_base}")
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    if args.dist_backend == "gloo":          # test mode: all ranks share the visible device(s)
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=args.dist_backend)

    from grappa_amd import Energy, MolwiseLoss, get_default_model_config, model_from_config, ops
    from grappa_amd.backend import get_backend
    from grappa_amd.datasets import WORKLOADS, build_workload
    from grappa_amd.dist import BucketedGradReducer
    from grappa_amd.optim import FlatParams, FusedAdam

    log("imports done; building model")
    be = get_backend()
    if args.gemm_precision:
        be.set_gemm_precision(args.gemm_precision)
    model = model_from_config(get_default_model_config())
    keyed_init(model)
    model = model.to(dev).train()
    flat = FlatParams(model)
    opt = FusedAdam(flat, lr=1.5e-5, max_grad_norm=10.0)
    reducer = BucketedGradReducer(model, flat)
    energy = Energy()
    loss_fn = MolwiseLoss(gradient_weight=0.8, energy_weight=1.0, param_weight=0.0, proper_regularisation=1e-3)
    per_gpu = WORKLOADS[args.workload][0]
    mol_ids = None
    if args.strong_global_batch:
        # strong scaling (not the default contract line): ONE global batch of the workload's molecule range, dealt to the ranks by
        # size (dist.shard_indices); per-GPU work shrinks as N grows
        from grappa_amd.datasets import pool_atom_counts, select_molecules
        from grappa_amd.dist import shard_indices
        _, lo, hi, _ = WORKLOADS[args.workload]
        all_ids = select_molecules(args.strong_global_batch, seed=0, min_atoms=lo, max_atoms=hi)
        sizes = [int(pool_atom_counts()[i]) for i in all_ids]
        mol_ids = [all_ids[j] for j in shard_indices(sizes, world, rank)]
        per_gpu = args.strong_global_batch / world
    loss_fn.global_batch_size = int(round(per_gpu * world))
    ops.manual_seed(1234 + rank)
    log("model ready; building workload")
    g = build_workload(args.workload, seed=rank, mol_ids=mol_ids).to(dev)
    plan = g.plan()
    log(f"workload ready: atoms {plan.N} tuples {plan.T}; warmup")

    def step():
        opt.zero_grad()
        for lvl in ("n2", "n3", "n4", "n4_improper"):         # drop last step's outputs
            for k in ("k", "eq"):
                g.nodes[lvl].data.pop(k, None)
        loss = loss_fn(energy(model(g)))
        loss.backward()
        reducer.finish()               # all-reduce of the flat gradient buffer (both buckets here unless the overlap is switched on)
        opt.step()
        return loss

    for _ in range(args.warmup):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    final_loss = float(loss.detach())
    log(f"timed region done: {1e3 * dt / args.steps:.1f} ms/step; instrumented pass")

    # instrumented repetition of the same steps: HIP events around every GEMM / GAT launch on the launch stream.  The writer
    # heads run on ONE stream here (GRAPPA_HEAD_STREAMS=1 semantics): with the heads on four streams a kernel shares the chip with
    # another head's kernels and an event pair measures the sharing, not the kernel; profiles/ rocprof runs use the same setting.
    head_streams = model.parameter_writer.head_streams
    model.parameter_writer.head_streams = 1
    step()
    be.start_profile()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
    prof = be.stop_profile()
    dt_prof = time.perf_counter() - t1
    model.parameter_writer.head_streams = head_streams

    log("instrumented pass done")
    # the same K steps with the dense products on the native fp32 matrix instruction, for reference next to the default
    alt = None
    if args.alt_precision and args.alt_precision != be.gemm_precision_name:
        default_precision = be.gemm_precision_name
        be.set_gemm_precision(args.alt_precision)
        step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt_alt = time.perf_counter() - t2
        if world > 1:
            t = torch.tensor([dt_alt], device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_alt = float(t)
        alt = {"gemm_precision": args.alt_precision, "value": per_gpu * world * args.steps / dt_alt, "ms_per_step": 1e3 * dt_alt / args.steps}
        be.set_gemm_precision(default_precision)
        log(f"alt precision {args.alt_precision}: {alt['ms_per_step']:.1f} ms/step")
    # optional third timing: forward products unchanged (fp32-grade), backward products (dgrad / wgrad) in --bwd-precision
    bwd = None
    if args.bwd_precision:
        be.set_gemm_precision_bwd(args.bwd_precision)
        step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t3 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt_bwd = time.perf_counter() - t3
        if world > 1:
            t = torch.tensor([dt_bwd], device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_bwd = float(t)
        bwd = {"backward_gemm_precision": args.bwd_precision, "value": per_gpu * world * args.steps / dt_bwd, "ms_per_step": 1e3 * dt_bwd / args.steps,
               "note": "NOT the headline configuration: forward products as in `default`, dgrad/wgrad products in the named arithmetic "
                       "(GRAPPA_GEMM_PRECISION_BWD); parameters / energies / forces / loss are bit-identical to the default"}
        be.set_gemm_precision_bwd(None)
        log(f"backward precision {args.bwd_precision}: {bwd['ms_per_step']:.1f} ms/step")
    if rank == 0:
        n, ms, fl, by = prof.get("gemm_f32", (0, 0.0, 0.0, 0.0))
        achieved = (fl / (ms * 1e-3)) / 1e12 if ms > 0 else 0.0
        # HBM traffic per launch of the dominant kernel: PMC counters cannot be read from inside the process; the committed
        # rocprofv3 --pmc summary of this same command (tools/pmc_traffic.py -> profiles/pmc_traffic_c2.json) is reported
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic_c2.json")
        if args.workload == "C2-pubchem-b256" and os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath))["families"]["gemm_f32"]["hbm_bytes_per_launch"]
                traffic_src = "profiles/pmc_traffic_c2.json (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, bytes per launch)"
            except Exception:
                traffic = None
        kname, nprod = GEMM_KERNELS[be.gemm_precision_name]
        # `achieved` counts the ALGORITHMIC fp32 FLOPs (2MNK); the split kernel issues `nprod` bf16 MFMAs per fp32 product, so
        # its ceiling in the same unit is the bf16 dense peak / nprod
        peak = PEAK_F32_MFMA_TFLOPS if nprod == 0 else PEAK_BF16_MFMA_TFLOPS / nprod
        roof = {"bound": "mfma", "kernel": kname, "achieved": achieved, "peak": peak,
                "unit": "TFLOP/s", "frac": achieved / peak, "frac_of_native_f32_mfma_peak": achieved / PEAK_F32_MFMA_TFLOPS,
                "gemm_precision": be.gemm_precision_name, "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": by / max(n, 1),
                "launches_per_step": n / args.steps, "avg_launch_us": 1e3 * ms / max(n, 1), "gflop_per_launch": fl / max(n, 1) / 1e9,
                "kernel_ms_per_step": ms / args.steps}
        gat = {}
        for fam in ("gat_fwd", "gat_bwd"):
            n_, ms_, fl_, by_ = prof.get(fam, (0, 0.0, 0.0, 0.0))
            a = (by_ / (ms_ * 1e-3)) / 1e9 if ms_ > 0 else 0.0
            gat[fam] = {"bound": "hbm", "achieved": a, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": a / PEAK_HBM_GBS,
                        "avg_launch_us": 1e3 * ms_ / max(n_, 1), "mb_per_launch": by_ / max(n_, 1) / 1e6}
        out = {
            "metric": "molecules/sec (train step, energy+force loss)", "value": per_gpu * world * args.steps / dt, "unit": "molecules/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "strong" if args.strong_global_batch else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {per_gpu} molecules/GPU (20-40 atoms, Espaloma pool), 32 conformations, "
                                   f"production GrappaModel 40.8M params random-init, train mode (dropout on), Adam + clip 10",
                       "molecules_per_gpu": per_gpu, "global_batch": per_gpu * world, "conformations": 32, "atoms_rank0": plan.N,
                       "tuples_rank0": {k: int(v) for k, v in plan.T.items()}, "parallelism": f"dp{world}",
                       "writer_head_streams": head_streams},
            "gemm_arithmetic": {"default": be.gemm_precision_name,
                                "note": "inputs, outputs, accumulation and every non-GEMM kernel are fp32; f32_bf16x6 splits each fp32 operand "
                                        "exactly into 3 bf16 pieces and sums the 6 largest partial products on the bf16 matrix cores "
                                        "(error vs a float64 product <= that of the native fp32 MFMA: tests/test_gpu_ops.py::"
                                        "test_gemm_precision_modes; end-to-end parity: tests/test_gpu_e2e.py)",
                                "native_f32_mfma": alt, "backward_reduced": bwd},
            "roofline": roof, "roofline_gat": gat, "ms_per_step_instrumented": 1e3 * dt_prof / args.steps, "final_loss": final_loss,
        }
        out["cpu_baseline"] = cpu_base
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
