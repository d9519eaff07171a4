"""GPU (-m gpu): the whole hot path on the HIP kernels -- GrappaModel -> Energy -> MolwiseLoss -> backward ->
fused Adam -- against (1) the golden fixtures produced by the reference's own modules and (2) the oracle
(oracle/cpu_ref.py) on larger seeded batches; plus size-independent properties at BASELINE.json's C2 size.
Tolerance (north star): 1e-4 relative, |a-b| <= 1e-4*max(|b|, floor), floors written below."""
import numpy as np
import pytest
import torch

import golden_utils as gu

pytestmark = pytest.mark.gpu

FLOORS = {"k": 1e-3, "kt": 5e-2, "eq": 1e-4}   # see tests/test_host_model.py for the reasoning behind the floors
TOL = 1e-4


def _assert_loaded():
    import ctypes
    from grappa_amd import _lib
    assert isinstance(_lib.load(), ctypes.CDLL)


def _check_graph(g, out, loss):
    for lvl in ["n2", "n3", "n4", "n4_improper"]:
        assert np.array_equal(out[f"{lvl}_idxs"], g.nodes[lvl].data["idxs"].cpu().numpy())
        assert gu.rel_err(g.nodes[lvl].data["k"].detach().cpu(), out[f"{lvl}_k"], FLOORS["kt" if lvl.startswith("n4") else "k"]) < TOL, lvl
        if lvl in ("n2", "n3"):
            assert gu.rel_err(g.nodes[lvl].data["eq"].detach().cpu(), out[f"{lvl}_eq"], FLOORS["eq"]) < TOL, lvl
        assert gu.rel_err_scaled(g.nodes["g"].data[f"energy_{lvl}"].cpu(), out[f"energy_{lvl}"], 1e-2, 1e-3) < TOL, lvl   # floor: 1 % of the largest term (sums of ~100 signed terms)
    assert gu.rel_err(g.nodes["n1"].data["h"].detach().cpu(), out["h"], 1e-1) < TOL
    assert gu.rel_err_scaled(g.nodes["g"].data["energy"].detach().cpu(), out["energy"], 1e-3, 1e-3) < TOL
    assert gu.rel_err_scaled(g.nodes["n1"].data["gradient"].detach().cpu(), out["gradient"], 1e-2, 1e-2) < TOL
    assert gu.rel_err(loss.detach().cpu(), out["loss"], 1e-6) < TOL


@pytest.mark.parametrize("name,n_confs,refs", [("ref_small_att.npz", 4, True), ("ref_small_conv.npz", 5, False),
                                                 ("ref_small_nonorm.npz", 5, False), ("ref_small_nosi.npz", 5, False),
                                                 ("ref_small_learnstats.npz", 5, False),     # last three: layer_norm=False / self_interaction=False / learnable_statistics=True
                                                 ("ref_tiny_wrongsym.npz", 4, True), ("ref_tiny_harmonic_gate.npz", 4, True),
                                                 ("ref_tiny_nper3.npz", 4, True), ("ref_tiny_offset_torsion.npz", 4, True), ("ref_tiny_nopos.npz", 4, True)])     # wrong_symmetry / harmonic_gate / n_periodicity_proper=3 / Energy(offset_torsion=True)
def test_small_configs_against_reference_goldens(name, n_confs, refs):
    from grappa_amd import Energy, GrappaModel, MolwiseLoss
    _assert_loaded()
    fx = gu.load(name)
    cfg = gu.config_of(fx)
    g = gu.build_batch(gu.molecules_of(fx), n_confs, refs, (cfg["n_periodicity_proper"], cfg["n_periodicity_improper"])).to("cuda")
    model = GrappaModel(**cfg)
    model.load_state_dict(gu.weights_for(fx, model))
    model = model.to("cuda").eval()
    g = Energy(**gu.energy_kwargs_of(fx))(model(g))
    loss = MolwiseLoss(**gu.loss_kwargs_of(fx))(g)
    loss.backward()
    out = gu.outputs_of(fx)
    _check_graph(g, out, loss)
    n = 0
    for k, p in model.named_parameters():
        ref = out.get("grad::" + k)
        if ref is None:
            continue
        scale = max(float(np.abs(ref).max()), 1e-8)
        assert float(np.abs(p.grad.cpu().numpy() - ref).max()) / scale < TOL, k      # 1e-4 of the tensor's scale
        n += 1
    assert n > 50


def test_production_config_against_reference_golden():
    from grappa_amd import Energy, MolwiseLoss, get_default_model_config, model_from_config
    fx = gu.load("ref_prod.npz")
    cfg = gu.config_of(fx)
    assert cfg == get_default_model_config()
    model = model_from_config(cfg)
    model.load_state_dict(gu.keyed_state_dict(model))
    model = model.to("cuda").eval()
    g = gu.build_batch(gu.molecules_of(fx), 4, True).to("cuda")
    g = Energy()(model(g))
    loss = MolwiseLoss(**gu.loss_kwargs_of(fx))(g)
    loss.backward()
    out = gu.outputs_of(fx)
    _check_graph(g, out, loss)
    norms = dict(zip(out["grad_norm_keys"].tolist(), out["grad_norm_vals"].tolist()))
    checked = 0
    for k, p in model.named_parameters():
        if k in norms and norms[k] > 1e-6:
            assert abs(float(p.grad.norm()) - norms[k]) / norms[k] < TOL, k
            checked += 1
    assert checked > 150


def test_production_widths_in_the_shipped_experiment_setting_against_reference_golden():
    """ref_prod_nper3.npz (oracle/make_goldens.py prod_nper3: the imported reference at production widths with n_periodicity_proper =
    n_periodicity_improper = 3, experiments/train-grappa-1.2.1/grappa_config.yaml:98-99): outputs, every gradient norm, and the full gradients of
    the four heads' last symmetriser layer (`symmetriser.mlp.2.*`, the layers whose width follows n_periodicity) on the HIP path"""
    from grappa_amd import Energy, MolwiseLoss, model_from_config
    fx = gu.load("ref_prod_nper3.npz")
    cfg = gu.config_of(fx)
    assert cfg["n_periodicity_proper"] == 3 and cfg["n_periodicity_improper"] == 3
    model = model_from_config(cfg)
    model.load_state_dict(gu.keyed_state_dict(model))
    model = model.to("cuda").eval()
    g = gu.build_batch(gu.molecules_of(fx), 4, True).to("cuda")
    g = Energy()(model(g))
    loss = MolwiseLoss(**gu.loss_kwargs_of(fx))(g)
    loss.backward()
    out = gu.outputs_of(fx)
    assert tuple(g.nodes["n4"].data["k"].shape[1:]) == (3,) and tuple(g.nodes["n4_improper"].data["k"].shape[1:]) == (3,)
    _check_graph(g, out, loss)
    full = {k[6:]: v for k, v in out.items() if k.startswith("grad::")}
    assert len(full) >= 16
    named = dict(model.named_parameters())
    for k, ref in full.items():
        scale = max(float(np.abs(ref).max()), 1e-8)
        assert float(np.abs(named[k].grad.cpu().numpy() - ref).max()) / scale < TOL, k      # 1e-4 of the tensor's scale
    norms = dict(zip(out["grad_norm_keys"].tolist(), out["grad_norm_vals"].tolist()))
    checked = 0
    for k, p in model.named_parameters():
        if k in norms and norms[k] > 1e-6:
            assert abs(float(p.grad.norm()) - norms[k]) / norms[k] < TOL, k
            checked += 1
    assert checked > 150


def _oracle_step(cfg, sd, g_cpu, loss_kwargs, double=False):
    """double: the oracle in float64 -- ground truth instead of a second fp32 evaluation (whose own rounding, which moves with the
    host's thread count, is of the size of the tolerance for the torsion energies: sums of ~100 signed terms of small k)"""
    from oracle import cpu_ref
    model = cpu_ref.RefGrappaModel(**cfg)
    model.load_state_dict(sd)
    model.eval()
    if double:
        model = model.double()
        for nt in g_cpu.ntypes:
            for k, v in list(g_cpu.nodes[nt].data.items()):
                if torch.is_tensor(v) and v.dtype == torch.float32:
                    g_cpu.nodes[nt].data[k] = v.double()
    g = cpu_ref.RefEnergy()(model(g_cpu))
    loss = cpu_ref.RefMolwiseLoss(**loss_kwargs)(g)
    loss.backward()
    return model, g, loss


def test_production_config_against_oracle_on_a_larger_batch():
    """32 molecules x 8 conformations, production widths: HIP vs the oracle on the same seeded inputs, incl. every parameter gradient."""
    from grappa_amd import Energy, MolwiseLoss, get_default_model_config, model_from_config
    from grappa_amd.datasets import build_batch_from_pool
    cfg = get_default_model_config()
    ids = list(range(500, 532))
    lk = dict(gradient_weight=0.8, energy_weight=1.0, param_weight=0.0, proper_regularisation=1e-3)
    model = model_from_config(cfg)
    sd = gu.keyed_state_dict(model)
    model.load_state_dict(sd)
    model = model.to("cuda").eval()
    g = build_batch_from_pool(ids, n_confs=8, seed=3).to("cuda")
    g = Energy()(model(g))
    loss = MolwiseLoss(**lk)(g)
    loss.backward()
    ref_model, rg, rloss = _oracle_step(cfg, sd, build_batch_from_pool(ids, n_confs=8, seed=3), lk, double=True)      # float64: ground truth
    out = {"h": rg.nodes["n1"].data["h"].detach().numpy(), "loss": rloss.detach().numpy().reshape(1),
           "energy": rg.nodes["g"].data["energy"].detach().numpy(), "gradient": rg.nodes["n1"].data["gradient"].detach().numpy()}
    for lvl in ["n2", "n3", "n4", "n4_improper"]:
        out[f"{lvl}_idxs"] = rg.nodes[lvl].data["idxs"].numpy()
        out[f"{lvl}_k"] = rg.nodes[lvl].data["k"].detach().numpy()
        out[f"energy_{lvl}"] = rg.nodes["g"].data[f"energy_{lvl}"].numpy()
        if lvl in ("n2", "n3"):
            out[f"{lvl}_eq"] = rg.nodes[lvl].data["eq"].detach().numpy()
    _check_graph(g, out, loss)
    ref_grads = dict(ref_model.named_parameters())
    worst = 0.0
    for k, p in model.named_parameters():
        r = ref_grads[k].grad
        if r is None:
            continue
        scale = max(float(r.abs().max()), 1e-8)
        worst = max(worst, float((p.grad.cpu() - r).abs().max()) / scale)
        assert float((p.grad.cpu() - r).abs().max()) / scale < TOL, k
    print("worst relative parameter-gradient error vs oracle:", worst)
    # the gate above is against float64 (on the torsion energies the fp32 oracle is itself 7e-5 from float64 and moves
    # with the host's thread count).  The distance to the fp32 oracle -- what north_star's 1e-4 literally names -- is measured and
    # recorded beside it (a drift must be seen), with an alarm at 3x the contract
    _, rg32, rloss32 = _oracle_step(cfg, sd, build_batch_from_pool(ids, n_confs=8, seed=3), lk, double=False)
    dist32 = {}
    for label, ref_g in (("fp32", rg32), ("fp64", rg)):
        gd, rd = g.nodes["g"].data, ref_g.nodes["g"].data
        for lvl in ["n2", "n3", "n4", "n4_improper"]:          # the measures of _check_graph, floors included
            dist32[(f"energy_{lvl}", label)] = gu.rel_err_scaled(gd[f"energy_{lvl}"].cpu(), rd[f"energy_{lvl}"].double().numpy(), 1e-2, 1e-3)
            dist32[(f"{lvl}_k", label)] = gu.rel_err(g.nodes[lvl].data["k"].detach().cpu(), ref_g.nodes[lvl].data["k"].detach().double().numpy(),
                                                     FLOORS["kt" if lvl.startswith("n4") else "k"])
        dist32[("energy", label)] = gu.rel_err_scaled(gd["energy"].detach().cpu(), rd["energy"].detach().double().numpy(), 1e-3, 1e-3)
        dist32[("gradient", label)] = gu.rel_err_scaled(g.nodes["n1"].data["gradient"].detach().cpu(),
                                                        ref_g.nodes["n1"].data["gradient"].detach().double().numpy(), 1e-2, 1e-2)
    keys = sorted({k for k, _ in dist32})
    report = "; ".join(f"{k}: vs fp32 oracle {dist32[(k, 'fp32')]:.2e} / vs fp64 oracle {dist32[(k, 'fp64')]:.2e}" for k in keys)
    print("GPU distance to the oracle in the measures of _check_graph --", report)
    import os
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_distance_fp32_vs_fp64_oracle.txt", "w") as fh:
        fh.write(report.replace("; ", "\n") + "\n")
    assert all(v < 3e-4 for (k, lab), v in dist32.items() if lab == "fp32"), report
    # The same quantities at SURVEY 8(d)'s own floors (k 1e-3 incl. the torsion constants, eq 1e-4, E 1e-3 kcal/mol, G 1e-2 kcal/mol/A --
    # absolute, not scaled with the tensor): three distances each -- GPU to float64, the fp32 ORACLE to float64 (what the reference's own
    # arithmetic can hold at these floors), GPU to the fp32 oracle.  Where the fp32 oracle itself is beyond 1e-4 of float64 the floors of
    # this file are fp32's, not this engine's: gate = the GPU's distance from float64 is of the order of the fp32 oracle's own (within 2x since round 5 -- measured <= 1.8x over the boxes and split-K plans of rounds 3 and 4; at these floors the measure sits on
    # a few elements that cancel to ~0, and moves by 2x with the summation order of a product -- split-K plan, thread count of the oracle)
    # (or inside the contract).
    def contract(got_g, ref_g):
        gd, rd = got_g.nodes["g"].data, ref_g.nodes["g"].data
        m = {}
        for lvl in ["n2", "n3", "n4", "n4_improper"]:
            m[f"{lvl}_k"] = gu.rel_err(got_g.nodes[lvl].data["k"].detach().cpu().double(), ref_g.nodes[lvl].data["k"].detach().double().numpy(), 1e-3)
            if lvl in ("n2", "n3"):
                m[f"{lvl}_eq"] = gu.rel_err(got_g.nodes[lvl].data["eq"].detach().cpu().double(), ref_g.nodes[lvl].data["eq"].detach().double().numpy(), 1e-4)
            m[f"energy_{lvl}"] = gu.rel_err(gd[f"energy_{lvl}"].detach().cpu().double(), rd[f"energy_{lvl}"].detach().double().numpy(), 1e-3)
        m["energy"] = gu.rel_err(gd["energy"].detach().cpu().double(), rd["energy"].detach().double().numpy(), 1e-3)
        m["gradient"] = gu.rel_err(got_g.nodes["n1"].data["gradient"].detach().cpu().double(), ref_g.nodes["n1"].data["gradient"].detach().double().numpy(), 1e-2)
        return m
    gpu64, ora64, gpu32 = contract(g, rg), contract(rg32, rg), contract(g, rg32)
    lines = [f"{k}: GPU vs fp64 {gpu64[k]:.2e} | fp32 oracle vs fp64 {ora64[k]:.2e} | GPU vs fp32 oracle {gpu32[k]:.2e}" for k in sorted(gpu64)]
    print("at SURVEY 8(d)'s floors --", "; ".join(lines))
    with open("gpurun_out/parity_distance_fp32_vs_fp64_oracle.txt", "a") as fh:
        fh.write("# at SURVEY 8(d)'s floors (k 1e-3, eq 1e-4, E 1e-3, G 1e-2, absolute)\n" + "\n".join(lines) + "\n")
    ratios = {k: (gpu64[k] / ora64[k] if ora64[k] > 0 else float("inf")) for k in gpu64}
    with open("gpurun_out/parity_distance_fp32_vs_fp64_oracle.txt", "a") as fh:
        fh.write("# GPU's distance from float64 over the fp32 oracle's own, per quantity (gate: < 2, or the GPU inside 1e-4)\n" +
                 "\n".join(f"{k}: {ratios[k]:.2f}" for k in sorted(ratios)) + "\n")
    # gate 1 (was 3 x): the GPU is no further from float64 than twice what the reference's own fp32 arithmetic is, or inside the contract
    for k in gpu64:
        assert gpu64[k] < max(TOL, 2.0 * ora64[k]), (k, gpu64[k], ora64[k])
    # gate 2: the LITERAL north-star contract -- GPU against the fp32 CPU path within 1e-4 at SURVEY 8(d)'s absolute floors -- for
    # the quantities fp32 itself holds there (bond / angle parameters, their energies, the total energy); the torsion constants and the forces
    # are where the fp32 oracle is itself 2e-4 .. 2.5e-3 from float64 at these floors and stay under gate 1
    for k in ("n2_k", "n2_eq", "n3_k", "n3_eq", "energy", "energy_n2", "energy_n3"):
        assert ora64[k] < TOL, (k, "the fp32 oracle itself leaves the contract here", ora64[k])
        assert gpu32[k] < TOL, (k, gpu32[k])


def test_train_step_decreases_loss_and_matches_oracle_adam():
    """two fused-Adam steps (flat buffers, device-side clip) track torch.optim.Adam + clip_grad_norm_ on the oracle."""
    from grappa_amd import Energy, GrappaModel, MolwiseLoss
    from grappa_amd.optim import FlatParams, FusedAdam
    from oracle import cpu_ref
    fx = gu.load("ref_small_att.npz")
    cfg = gu.config_of(fx)
    lk = dict(gradient_weight=0.8, energy_weight=1.0, param_weight=0.0)
    mols = gu.molecules_of(fx)
    model = GrappaModel(**cfg)
    model.load_state_dict(gu.state_dict_of(fx))
    model = model.to("cuda").eval()
    flat = FlatParams(model)
    opt = FusedAdam(flat, lr=1e-3, max_grad_norm=10.0)
    ref = cpu_ref.RefGrappaModel(**cfg)
    ref.load_state_dict(gu.state_dict_of(fx))
    ref.eval()
    ropt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    losses, rlosses = [], []
    for step in range(3):
        g = gu.build_batch(mols, 4, False).to("cuda")
        opt.zero_grad()
        loss = MolwiseLoss(**lk)(Energy()(model(g)))
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
        rg = gu.build_batch(mols, 4, False)
        ropt.zero_grad()
        rl = cpu_ref.RefMolwiseLoss(**lk)(cpu_ref.RefEnergy()(ref(rg)))
        rl.backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 10.0)
        ropt.step()
        rlosses.append(float(rl.detach()))
    assert losses[-1] < losses[0]
    for a, b in zip(losses, rlosses):
        assert abs(a - b) / abs(b) < 1e-3, (losses, rlosses)


def test_properties_at_c2_size():
    """BASELINE.json configs[1] (256 molecules of 20-40 atoms, 32 conformations, production model, train-mode dropout):
    size-independent properties instead of a CPU recomputation."""
    from grappa_amd import Energy, get_default_model_config, model_from_config, ops
    from grappa_amd.datasets import build_batch_from_pool, build_workload, workload_molecule_ids
    model = model_from_config(get_default_model_config())
    model.load_state_dict(gu.keyed_state_dict(model))
    model = model.to("cuda")
    ids = workload_molecule_ids("C2-pubchem-b256", seed=0)
    g = build_workload("C2-pubchem-b256", seed=0).to("cuda")
    model.eval()
    with torch.no_grad():
        g = Energy()(model(g))
    E, G = g.nodes["g"].data["energy"], g.nodes["n1"].data["gradient"]
    assert torch.isfinite(E).all() and torch.isfinite(G).all()
    plan = g.plan()
    # (1) zero net force per molecule and conformation (translation invariance of the MM energy)
    ptr = plan.atom_molptr.long()
    seg = torch.repeat_interleave(torch.arange(plan.B, device="cuda"), ptr[1:] - ptr[:-1])
    net = torch.zeros(plan.B, E.shape[1], 3, device="cuda").index_add(0, seg, G)
    assert float(net.abs().max()) < 1e-4 * float(G.abs().max()) * 40
    # (2) batching invariance (reference tests/unbatch.py): molecule 17 alone == inside the batch
    with torch.no_grad():
        g1 = Energy()(model(build_batch_from_pool([ids[17]], n_confs=32, seed=0).to("cuda")))
    e1 = g1.nodes["g"].data["energy"][0]
    assert gu.rel_err_scaled(E[17].cpu(), e1.cpu().numpy(), 1e-3, 1e-3) < 1e-4
    a0, a1 = int(ptr[17]), int(ptr[18])
    assert gu.rel_err_scaled(G[a0:a1].cpu(), g1.nodes["n1"].data["gradient"].cpu().numpy(), 1e-2, 1e-2) < 1e-4
    # (3) permutation symmetry of the heads: reversing every proper torsion leaves k unchanged
    k4 = g.nodes["n4"].data["k"].clone()
    g2 = build_workload("C2-pubchem-b256", seed=0)
    g2.nodes["n4"].data["idxs"] = g2.nodes["n4"].data["idxs"].flip(1).contiguous()
    g2 = g2.to("cuda")
    with torch.no_grad():
        g2 = model(g2)
    assert gu.rel_err(g2.nodes["n4"].data["k"].cpu(), k4.cpu().numpy(), 1e-2) < 1e-4
    # (4) train mode: dropout is active, finite, and reproducible for a fixed seed
    model.train()
    ops.manual_seed(123)
    with torch.no_grad():
        ka = model(build_workload("C2-pubchem-b256", seed=0).to("cuda")).nodes["n2"].data["k"].clone()
    ops.manual_seed(123)
    with torch.no_grad():
        kb = model(build_workload("C2-pubchem-b256", seed=0).to("cuda")).nodes["n2"].data["k"].clone()
    assert torch.equal(ka, kb) and torch.isfinite(ka).all()
    assert not torch.allclose(ka, g.nodes["n2"].data["k"])


def test_first_layer_on_atom_position_rows_equals_the_token_formulation():
    """ops.ProjFirstLayerFn (LayerNorm + QKV product of a writer's first layer once per (atom, position) row, gathered to the tokens
    behind the product) against ProjGatherFn + TransformerLayerFn on the tokens: same parameters, loss and parameter gradients up to
    fp32 summation order (the token gradients are summed into table rows before the products instead of inside them), train mode
    with the same dropout masks"""
    from grappa_amd import Energy, MolwiseLoss, get_default_model_config, model_from_config, ops
    from grappa_amd.datasets import build_batch_from_pool
    from grappa_amd.optim import FlatParams
    model = model_from_config(get_default_model_config())
    model.load_state_dict(gu.keyed_state_dict(model))
    model = model.to("cuda").train()
    flat = FlatParams(model)
    g0 = build_batch_from_pool(list(range(300, 364)), n_confs=8, seed=5).to("cuda")
    plan = g0.plan()
    assert 4 * plan.N <= 3 * plan.T["n4"] and 4 * plan.N <= 3 * plan.T["n3"] and 4 * plan.N > 3 * plan.T["n2"]      # propers + angles take the new path
    res = []
    old = ops.FIRST_LAYER_ON_ATOM_ROWS
    try:
        for rows in (False, True):
            ops.FIRST_LAYER_ON_ATOM_ROWS = rows
            ops.manual_seed(21)
            flat.zero_grad()
            g = Energy()(model(build_batch_from_pool(list(range(300, 364)), n_confs=8, seed=5).to("cuda")))
            loss = MolwiseLoss(gradient_weight=0.8, energy_weight=1.0, param_weight=0.0, proper_regularisation=1e-3)(g)
            loss.backward()
            torch.cuda.synchronize()
            res.append((loss.detach().clone(), flat.grad.clone(), {lvl: g.nodes[lvl].data["k"].detach().clone() for lvl in ("n2", "n3", "n4", "n4_improper")}))
    finally:
        ops.FIRST_LAYER_ON_ATOM_ROWS = old
    (l0, g0_, k0), (l1, g1_, k1) = res
    assert abs(float(l0) - float(l1)) <= 1e-5 * abs(float(l0))
    for lvl in k0:
        assert float((k0[lvl] - k1[lvl]).abs().max()) <= 2e-5 * float(k0[lvl].abs().max()), lvl
    assert torch.equal(k0["n2"], k1["n2"]) and torch.equal(k0["n4_improper"], k1["n4_improper"])      # untouched writers: the same bits
    names = {id(p): k for k, p in model.named_parameters()}
    for p in flat.params:                           # per tensor, against that tensor's own scale
        lo, hi = flat._offsets[id(p)]
        a, b = g0_[lo:hi], g1_[lo:hi]
        scale = float(a.abs().max())
        if scale > 0:
            assert float((a - b).abs().max()) <= 1e-4 * scale, names[id(p)]


def test_c3_batch_1024_bf16_arithmetic():
    """BASELINE.json configs[2]: 1024 molecules drawn from the whole Espaloma pool (~39 k atoms, ~0.75 M tokens), 32 conformations,
    production model, dense products in bf16 arithmetic on the matrix cores (fp32 accumulate; LayerNorm / softmax / energy in
    fp32).  No CPU recomputation at this size: the bf16 modes are checked against the fp32-grade default ON THE SAME BATCH with
    the tolerance SURVEY section 8(d) states for the bf16 configuration (2e-2 relative on parameters, stated floors), bf16x3
    (two bf16 pieces per operand, ~2^-16) an order of magnitude tighter; one train step in bf16x3 must give a finite loss and
    finite, non-zero gradients."""
    from grappa_amd import Energy, MolwiseLoss, get_default_model_config, model_from_config
    from grappa_amd.backend import get_backend
    from grappa_amd.datasets import build_workload
    from grappa_amd.optim import FlatParams
    be = get_backend()
    model = model_from_config(get_default_model_config())
    model.load_state_dict(gu.keyed_state_dict(model))
    model = model.to("cuda").eval()
    g_cpu = build_workload("C3-espaloma-b1024", seed=0)
    assert g_cpu.plan().B == 1024
    default = be.gemm_precision_name
    outs = {}
    try:
        for mode in (default, "bf16x3", "bf16"):
            be.set_gemm_precision(mode)
            with torch.no_grad():
                g = Energy()(model(g_cpu.to("cuda")))
            outs[mode] = {(lvl, k): g.nodes[lvl].data[k].float().cpu() for lvl in ("n2", "n3", "n4", "n4_improper") for k in ("k", "eq")
                          if k in g.nodes[lvl].data}
            outs[mode]["E"] = g.nodes["g"].data["energy"].cpu()
            assert all(torch.isfinite(v).all() for v in outs[mode].values()), mode
        for mode, tol in (("bf16x3", 2e-3), ("bf16", 2e-2)):
            for key, ref in outs[default].items():
                if key == "E":
                    continue
                lvl, k = key
                # floors = half the output scale of the head (k = c * k_std + k_mean: std 0.5 / 1.2 proper, 4.1 improper; bond/angle k ~ 1e2)
                floor = FLOORS["eq"] if k == "eq" else {"n4": 0.5, "n4_improper": 2.0}.get(lvl, 1.0)
                assert gu.rel_err(outs[mode][key], ref.numpy(), floor) < tol, (mode, key)
        # one bf16x3 train step at this size
        be.set_gemm_precision("bf16x3")
        model.train()
        flat = FlatParams(model)
        flat.zero_grad()
        loss = MolwiseLoss(gradient_weight=0.8, energy_weight=1.0, param_weight=0.0, proper_regularisation=1e-3)(Energy()(model(g_cpu.to("cuda"))))
        loss.backward()
        torch.cuda.synchronize()
        assert torch.isfinite(loss) and torch.isfinite(flat.grad).all() and float(flat.grad.abs().max()) > 0
    finally:
        be.set_gemm_precision(default)


def test_backward_products_in_bf16x3_leave_the_forward_untouched():
    """optional GRAPPA_GEMM_PRECISION_BWD=bf16x3: loss / parameters bit-identical (forward GEMMs keep the default arithmetic), every
    parameter gradient within 1e-3 of its tensor's max of the default's (2^-16 per product, averaged over 10^4..10^5 summands)"""
    from grappa_amd import Energy, MolwiseLoss, get_default_model_config, model_from_config, ops
    from grappa_amd.backend import get_backend
    from grappa_amd.datasets import build_batch_from_pool
    from grappa_amd.optim import FlatParams
    be = get_backend()
    model = model_from_config(get_default_model_config())
    model.load_state_dict(gu.keyed_state_dict(model))
    model = model.to("cuda").eval()
    flat = FlatParams(model)
    g_cpu = build_batch_from_pool(list(range(100, 132)), n_confs=8, seed=3)
    res = []
    try:
        for bwd in (None, "bf16x3"):
            be.set_gemm_precision_bwd(bwd)
            flat.zero_grad()
            g = Energy()(model(g_cpu.to("cuda")))
            loss = MolwiseLoss(gradient_weight=0.8, energy_weight=1.0, param_weight=0.0, proper_regularisation=1e-3)(g)
            loss.backward()
            torch.cuda.synchronize()
            res.append((loss.detach().clone(), g.nodes["n3"].data["k"].detach().clone(), [p.grad.detach().clone() for p in flat.params]))
    finally:
        be.set_gemm_precision_bwd(None)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    worst = max(float((a - b).abs().max() / a.abs().max().clamp_min(1e-20)) for a, b in zip(res[0][2], res[1][2]) if float(a.abs().max()) > 0)
    assert 0 < worst < 1e-3, worst


def test_writer_heads_on_streams_match_single_stream():
    """the writer heads on four HIP streams (the default) against GRAPPA_HEAD_STREAMS=1: loss, parameters and gradients are
    bit-identical when both plan the same K cuts (grouped launches off, split-K tails pinned); single-stream runs must be
    bit-reproducible.  (History: a since-removed LayerNorm kernel deviated under concurrent queues)"""
    from grappa_amd import Energy, MolwiseLoss, get_default_model_config, model_from_config, ops
    from grappa_amd.datasets import build_batch_from_pool
    from grappa_amd.optim import FlatParams
    model = model_from_config(get_default_model_config())
    model.load_state_dict(gu.keyed_state_dict(model))
    model = model.to("cuda").train()
    flat = FlatParams(model)
    g_cpu = build_batch_from_pool(list(range(100, 148)), n_confs=8, seed=3)
    res = []
    from grappa_amd.backend import get_backend
    be = get_backend()
    defer, be.defer_wgrads = be.defer_wgrads, False        # grouped weight gradients exist on the caller's stream only: another summation order
    be.pin_tail_launches(False)                            # (the model plans without split-K tails on four streams, with them on one: same K cuts here)
    for streams in (1, 1, 4):
        model.parameter_writer.head_streams = streams
        ops.manual_seed(77)
        flat.zero_grad()
        g = Energy()(model(g_cpu.to("cuda")))
        loss = MolwiseLoss(gradient_weight=0.8, energy_weight=1.0, param_weight=0.0, proper_regularisation=1e-3)(g)
        loss.backward()
        torch.cuda.synchronize()
        res.append((loss.detach().clone(), flat.grad.clone(), g.nodes["n4"].data["k"].detach().clone(), g.nodes["n2"].data["eq"].detach().clone()))
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)                                   # one stream: bit-reproducible
    loss1, grad1, k41, eq1 = res[0]
    loss4, grad4, k44, eq4 = res[2]
    model.parameter_writer.head_streams = 1
    be.defer_wgrads = defer
    be.pin_tail_launches(None)
    # with the shipped kernels the four-stream step has always been bit-identical; anything else is the multi-queue deviation of
    # DESIGN.md section 6 coming back and must be seen, not absorbed by a tolerance
    assert torch.equal(loss4, loss1) and torch.equal(k44, k41) and torch.equal(eq4, eq1) and torch.equal(grad4, grad1)


def test_train_steps_on_four_streams_match_single_stream():
    """SEVERAL optimiser steps with the heads on four streams against one stream: same losses, same parameters.  One forward / backward
    (the test above) cannot see a race that needs a parameter update between two uses -- a refresh of per-weight data (operand scales,
    pairs, planes, packed weights) issued from one head's stream while another head reads it.  (History: DESIGN.md section 6, the NaN
    'gain' of a per-head weight copy.)"""
    from grappa_amd import Energy, MolwiseLoss, get_default_model_config, model_from_config, ops
    from grappa_amd.backend import get_backend
    from grappa_amd.datasets import build_batch_from_pool
    from grappa_amd.optim import FlatParams, FusedAdam
    be = get_backend()
    g_cpu = build_batch_from_pool(list(range(100, 164)), n_confs=8, seed=5)
    loss_fn = MolwiseLoss(gradient_weight=0.8, energy_weight=1.0, param_weight=0.0, proper_regularisation=1e-3)
    defer, be.defer_wgrads = be.defer_wgrads, False        # (as above: one summation order for both runs)
    be.pin_tail_launches(False)
    out = {}
    try:
        for streams in (1, 4):
            model = model_from_config(get_default_model_config())
            model.load_state_dict(gu.keyed_state_dict(model))
            model = model.to("cuda").train()
            model.parameter_writer.head_streams = streams
            flat = FlatParams(model)
            opt = FusedAdam(flat, lr=1e-4)
            ops.manual_seed(123)
            g = g_cpu.to("cuda")
            losses = []
            for _ in range(4):
                opt.zero_grad()
                for lvl in ("n2", "n3", "n4", "n4_improper"):
                    for k in ("k", "eq"):
                        g.nodes[lvl].data.pop(k, None)
                loss = loss_fn(Energy()(model(g)))
                loss.backward()
                opt.step()
                losses.append(loss.detach().clone())
            torch.cuda.synchronize()
            out[streams] = (torch.stack(losses), flat.data.clone())
    finally:
        be.defer_wgrads = defer
        be.pin_tail_launches(None)
    l1, p1 = out[1]
    l4, p4 = out[4]
    assert torch.isfinite(l1).all() and torch.isfinite(l4).all() and torch.isfinite(p4).all()
    assert float(l1[-1]) != float(l1[0])                          # the steps did move the parameters
    assert torch.equal(l4, l1) and torch.equal(p4, p1)


def test_predict_drop_in():
    from grappa_amd import Grappa, Molecule, get_default_model_config, model_from_config
    from grappa_amd.datasets import molecule_from_pool
    model = model_from_config(get_default_model_config())
    model.load_state_dict(gu.keyed_state_dict(model))
    params = Grappa(model, device="cuda").predict(molecule_from_pool(42))
    assert params.bond_k.shape[0] == params.bonds.shape[0] and params.proper_ks.shape[1] == 6 and params.improper_ks.shape[1] == 3
    assert (params.proper_ks >= 0).all() and np.isin(params.proper_phases, [0.0, np.float32(np.pi)]).all()


def test_c5_protein_size_inference_matches_oracle():
    """BASELINE.json configs[4]: inference-only parametrisation of ONE >= 50k-atom protein graph (19 disjoint copies of all-atom T4
    lysozyme = 50,046 atoms, amber99 template charges), single forward on the GPU vs the oracle on the CPU, parameters within 1e-4."""
    import time
    from grappa_amd import get_default_model_config, model_from_config
    from grappa_amd.datasets import protein_graph_t4
    from oracle import cpu_ref
    cfg = get_default_model_config()
    model = model_from_config(cfg)
    sd = gu.keyed_state_dict(model)
    model.load_state_dict(sd)
    model = model.to("cuda").eval()
    g_cpu = protein_graph_t4(19)
    assert g_cpu.num_nodes("n1") == 50046 and g_cpu.batch_size == 1
    g = g_cpu.to("cuda")
    with torch.no_grad():
        model(g)                      # warm-up (plan build, workspace)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g = model(g)
        torch.cuda.synchronize()
        t_gpu = time.perf_counter() - t0
    ref = cpu_ref.RefGrappaModel(**cfg)
    ref.load_state_dict(sd)
    ref.eval()
    t0 = time.perf_counter()
    with torch.no_grad():
        rg = ref(g_cpu)
    t_cpu = time.perf_counter() - t0
    print(f"C5: {g_cpu.num_nodes('n1')} atoms, tuples { {l: g_cpu.num_nodes(l) for l in ['n2', 'n3', 'n4', 'n4_improper']} }: "
          f"GPU forward {1e3 * t_gpu:.1f} ms, oracle CPU forward {t_cpu:.1f} s ({torch.get_num_threads()} threads)")
    for lvl in ["n2", "n3", "n4", "n4_improper"]:
        floor = FLOORS["kt" if lvl.startswith("n4") else "k"]
        assert gu.rel_err(g.nodes[lvl].data["k"].cpu(), rg.nodes[lvl].data["k"].numpy(), floor) < TOL, lvl
        if lvl in ("n2", "n3"):
            assert gu.rel_err(g.nodes[lvl].data["eq"].cpu(), rg.nodes[lvl].data["eq"].numpy(), FLOORS["eq"]) < TOL, lvl
    try:
        import os
        with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "c5_inference.txt"), "w") as f:
            f.write(f"atoms {g_cpu.num_nodes('n1')} gpu_forward_ms {1e3 * t_gpu:.2f} oracle_cpu_forward_s {t_cpu:.2f} cpu_threads {torch.get_num_threads()}\n")
    except OSError:
        pass
