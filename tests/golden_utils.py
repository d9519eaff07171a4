"""Helpers shared by the parity tests: rebuild graphs from the committed golden fixtures
(tests/golden/*.npz, produced by oracle/make_goldens.py from the reference's own modules)
through the PRODUCT's host code (grappa_amd.Molecule -> MolBatch), so that the fixtures also pin
the batch construction (tuple row order, idx shifting, conformation padding)."""
import ast
import hashlib
import os

import numpy as np
import torch

from grappa_amd.batch import batch, set_number_confs
from grappa_amd.molecule import Molecule

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def config_of(fx):
    return {k: ast.literal_eval(v) for k, v in zip(fx["cfg_keys"].tolist(), fx["cfg_vals"].tolist())}


def loss_kwargs_of(fx):
    return {k: float(v) for k, v in zip(fx["loss_kwargs_keys"].tolist(), fx["loss_kwargs_vals"].tolist())}


def state_dict_of(fx):
    return {k[4:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("sd::")}


def weights_for(fx, model):
    """the weights the reference run used: stored in the fixture (sd::*), or -- fixtures that ship none -- derived from the state-dict keys"""
    sd = state_dict_of(fx)
    return sd if sd else keyed_state_dict(model)


def energy_kwargs_of(fx):
    if "energy_kwargs_keys" not in fx.files:
        return {}
    return {k: ast.literal_eval(v) for k, v in zip(fx["energy_kwargs_keys"].tolist(), fx["energy_kwargs_vals"].tolist())}


def outputs_of(fx):
    return {k[5:]: fx[k] for k in fx.files if k.startswith("out::")}


def molecules_of(fx):
    n = int(fx["n_mols"][0])
    mols = []
    for i in range(n):
        pre = f"mol{i}::"
        mols.append({k[len(pre):]: fx[k] for k in fx.files if k.startswith(pre)})
    return mols


def param_refs(m, g, n_per=(6, 3), nan_refs=True):
    """the synthetic classical parameters make_goldens.py attached (same generator, same order)."""
    rng = np.random.default_rng(int(m["seed"]))
    for lvl, name, mean, std, shape1 in [("n2", "k", 700., 150., None), ("n2", "eq", 1.2, 0.15, None), ("n3", "k", 100., 25., None),
                                          ("n3", "eq", 1.95, 0.1, None), ("n4", "k", 0., 0.8, n_per[0]),
                                          ("n4_improper", "k", 0., 2.0, n_per[1])]:
        T = g.num_nodes(lvl)
        shape = (T,) if shape1 is None else (T, shape1)
        v = rng.normal(mean, std, size=shape).astype(np.float32)
        if name == "k" and shape1 is None:
            v = np.abs(v)
        g.nodes[lvl].data[name + "_ref"] = torch.from_numpy(v)
    if nan_refs and int(m["seed"]) % 2 == 1 and g.num_nodes("n3") > 2:
        g.nodes["n3"].data["k_ref"][:2] = float("nan")


def molecule_of(m) -> Molecule:
    """the product's Molecule for one fixture record (the reference run was fed the same graph, charges and features)."""
    mol = Molecule(atoms=list(range(len(m["z"]))), bonds=[tuple(int(x) for x in b) for b in m["bonds"]],
                   impropers=[tuple(int(x) for x in r) for r in m["impropers"]],
                   atomic_numbers=[int(x) for x in m["z"]], partial_charges=[float(x) for x in m["q"]],
                   charge_model=str(m["charge_model"]))
    # the featuriser must reproduce what the reference run was fed
    assert np.array_equal(mol.additional_features["ring_encoding"], m["ring_encoding"])
    assert np.array_equal(mol.additional_features["degree"], m["degree"])
    return mol


def build_batch(mols, n_confs, with_param_refs=True, n_per=(6, 3), nan_refs=True):
    graphs = []
    for m in mols:
        mol = molecule_of(m)
        g = mol.to_dgl()
        g.nodes["n1"].data["xyz"] = torch.from_numpy(m["xyz"].copy())
        g.nodes["g"].data["energy_ref"] = torch.from_numpy(m["energy_ref"].copy())
        g.nodes["n1"].data["gradient_ref"] = torch.from_numpy(m["gradient_ref"].copy())
        if with_param_refs:
            param_refs(m, g, n_per, nan_refs)
        g = set_number_confs(g, n_confs)
        graphs.append(g)
    return batch(graphs)


def rel_err(a, b, floor):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def rel_err_scaled(a, b, floor_frac=1e-3, floor_abs=0.0):
    """max |a-b| / max(|b|, floor) with floor = max(floor_abs, floor_frac * max|b|): element-wise relative error whose
    floor follows the scale of the tensor (forces/energies are sums of O(20) large terms that cancel; an element that
    cancels to ~0 keeps the absolute rounding error of its summands)."""
    b64 = np.asarray(b, dtype=np.float64)
    if b64.size == 0:
        return 0.0
    return rel_err(a, b, max(floor_abs, floor_frac * float(np.abs(b64).max())))


def keyed_tensor(key: str, shape, scale: float) -> torch.Tensor:
    seed = int.from_bytes(hashlib.sha256(key.encode()).digest()[:4], "little")
    gen = torch.Generator().manual_seed(seed)
    return (torch.rand(tuple(shape), generator=gen) * 2 - 1) * scale


def keyed_state_dict(model) -> dict:
    """Deterministic weights derived from the state-dict KEY (so the production-size weights never need to be
    shipped): U(+-1/sqrt(fan_in)) matrices, small biases, LayerNorm gamma near 1; buffers are kept.  The aliased
    gnn.blocks.* entries are made identical to their gnn.att_blocks.* / gnn.conv_blocks.* twins."""
    sd = model.state_dict()
    new = {}
    for k, v in sd.items():
        if not v.dtype.is_floating_point or k.endswith(("positional_encoding", "permutation_prefactors", "k_mean", "k_std",
                                                           "mean_over_std", ".std", "min_", "std_over_max", ".max")):
            new[k] = v.clone()
        elif v.dim() == 2:
            new[k] = keyed_tensor(k, v.shape, 1.0 / np.sqrt(v.shape[1]))
        elif "norm" in k and k.endswith("weight"):
            new[k] = 1.0 + keyed_tensor(k, v.shape, 0.1)
        else:
            new[k] = keyed_tensor(k, v.shape, 0.05)
    n_conv = len(model.gnn.conv_blocks) if hasattr(model.gnn, "conv_blocks") else 0
    for k in list(new.keys()):
        if k.startswith("gnn.blocks."):
            parts = k.split(".")
            i = int(parts[2])
            alias = (f"gnn.conv_blocks.{i}." if i < n_conv else f"gnn.att_blocks.{i - n_conv}.") + ".".join(parts[3:])
            new[k] = new[alias]
    return new
