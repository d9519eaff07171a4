"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*.npz by running the REFERENCE's own
modules (imported verbatim from /root/reference/src, with the pure-torch DGL shim of
oracle/dgl_shim on sys.path).  Runs only in the build container; the fixtures it writes are data
(inputs + expected outputs +, for the small config, the weights that produced them).

    python oracle/make_goldens.py            # writes tests/golden/ref_small_*.npz, ref_prod.npz, ref_energy.npz

What is run, per fixture:  Sequential(GrappaModel(**cfg), Energy()) in eval mode on a batch built
by the reference's Molecule.to_dgl -> set_number_confs -> dgl_utils.batch, then MolwiseLoss and
loss.backward().  The dihedral noise (models/internal_coordinates.py:194-196) is patched
to zero while the goldens are produced; a noise-on run is recorded beside it to bound its effect.
Batches avoid exactly-3 angles / exactly-3 conformations (torch.cross axis quirk).
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "dgl_shim"))
sys.path.insert(0, "/root/reference/src")
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")

import grappa  # noqa: E402  (the reference)
from grappa.data import Molecule as RefMolecule  # noqa: E402
from grappa.models import Energy as RefEnergy, GrappaModel as RefGrappaModel, get_default_model_config  # noqa: E402
from grappa.training.loss import MolwiseLoss as RefMolwiseLoss  # noqa: E402
from grappa.utils import dgl_utils as ref_dgl_utils  # noqa: E402
from grappa.utils.graph_utils import get_default_statistics  # noqa: E402

from grappa_amd import featurize, tuple_indices  # noqa: E402  (graph featuriser only: rdkit is absent)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_utils import keyed_state_dict  # noqa: E402  (weights derived from state-dict keys; shared with the tests)

OUT = os.path.join(ROOT, "tests", "golden")
POOL = np.load(os.path.join(ROOT, "grappa_amd", "data", "espaloma_pool.npz"))


def pool_molecule(i):
    a0, a1 = POOL["atom_ptr"][i], POOL["atom_ptr"][i + 1]
    b0, b1 = POOL["bond_ptr"][i], POOL["bond_ptr"][i + 1]
    return POOL["z"][a0:a1].astype(np.int64), POOL["bonds"][b0:b1].astype(np.int64), POOL["xyz"][a0:a1].astype(np.float32)


def build_inputs(mol_ids, n_confs, seed, charge_model="am1BCC", with_param_refs=True, pad_confs_of=None):
    """-> list of dicts of numpy arrays (one per molecule): everything needed to rebuild the graphs."""
    rng = np.random.default_rng(seed)
    mols = []
    for j, mid in enumerate(mol_ids):
        z, bonds, xyz0 = pool_molecule(mid)
        n = len(z)
        q = rng.normal(0, 0.3, size=n).astype(np.float32)
        q = (q - q.mean()).astype(np.float32)
        nd = tuple_indices.get_neighbor_dict([tuple(b) for b in bonds.tolist()], sort=True)
        imps = tuple_indices.improper_centres_from_bonds(bonds.tolist(), nd, z)
        c = n_confs if (pad_confs_of is None or j not in pad_confs_of) else pad_confs_of[j]
        xyz = xyz0[:, None, :] + rng.normal(0, 0.08, size=(n, c, 3)).astype(np.float32)
        m = dict(z=z, bonds=bonds, impropers=np.asarray(imps, dtype=np.int64).reshape(-1, 4), q=q, xyz=xyz.astype(np.float32),
                 energy_ref=rng.normal(0, 3, size=(1, c)).astype(np.float32),
                 gradient_ref=rng.normal(0, 10, size=(n, c, 3)).astype(np.float32),
                 ring_encoding=featurize.ring_encoding(n, bonds), degree=featurize.degree_encoding(n, bonds),
                 charge_model=charge_model, seed=seed + j)
        mols.append(m)
    return mols


def ref_graph(m, n_confs, with_param_refs, n_per_ref=(6, 3), nan_refs=True):
    bonds = [tuple(int(x) for x in b) for b in m["bonds"]]
    mol = RefMolecule(atoms=list(range(len(m["z"]))), bonds=bonds, impropers=[tuple(int(x) for x in r) for r in m["impropers"]],
                      atomic_numbers=[int(x) for x in m["z"]], partial_charges=[float(x) for x in m["q"]],
                      additional_features={"ring_encoding": m["ring_encoding"], "degree": m["degree"]},
                      ring_encoding=False, degree=False, charge_model=m["charge_model"])
    g = mol.to_dgl()
    g.nodes["n1"].data["xyz"] = torch.from_numpy(m["xyz"])
    g.nodes["g"].data["energy_ref"] = torch.from_numpy(m["energy_ref"])
    g.nodes["n1"].data["gradient_ref"] = torch.from_numpy(m["gradient_ref"])
    if with_param_refs:
        rng = np.random.default_rng(m["seed"])
        for lvl, name, mean, std, shape1 in [("n2", "k", 700., 150., None), ("n2", "eq", 1.2, 0.15, None), ("n3", "k", 100., 25., None),
                                              ("n3", "eq", 1.95, 0.1, None), ("n4", "k", 0., 0.8, n_per_ref[0]),
                                              ("n4_improper", "k", 0., 2.0, n_per_ref[1])]:
            T = g.num_nodes(lvl)
            shape = (T,) if shape1 is None else (T, shape1)
            v = rng.normal(mean, std, size=shape).astype(np.float32)
            if name == "k" and shape1 is None:
                v = np.abs(v)
            g.nodes[lvl].data[name + "_ref"] = torch.from_numpy(v)
        # a few NaN references (molecules without classical parameters are stored like this)
        if nan_refs and m["seed"] % 2 == 1 and g.num_nodes("n3") > 2:
            g.nodes["n3"].data["k_ref"][:2] = float("nan")
    g = ref_dgl_utils.set_number_confs(g, n_confs)
    return g, mol


def to_np(t):
    return t.detach().cpu().numpy()


class zero_dihedral_noise:
    def __enter__(self):
        self._orig = torch.randn_like
        torch.randn_like = lambda x, *a, **k: torch.zeros_like(x)

    def __exit__(self, *a):
        torch.randn_like = self._orig


def run_reference(cfg, mols, n_confs, state_dict=None, loss_kwargs=None, with_param_refs=True, noise=False, grads="all", energy_kwargs=None):
    torch.manual_seed(0)
    model = RefGrappaModel(**cfg)
    sd = keyed_state_dict(model) if state_dict is None else state_dict
    model.load_state_dict(sd)
    full = torch.nn.Sequential(model, RefEnergy(suffix="", gradients=True, **(energy_kwargs or {})))
    full.eval()
    graphs = [ref_graph(m, n_confs, with_param_refs, (cfg["n_periodicity_proper"], cfg["n_periodicity_improper"]))[0] for m in mols]
    g = ref_dgl_utils.batch(graphs)
    loss_fn = RefMolwiseLoss(**(loss_kwargs or {}))
    ctx = zero_dihedral_noise() if not noise else torch.no_grad().__class__()  # noqa
    if not noise:
        with zero_dihedral_noise():
            g = full(g)
            loss = loss_fn(g)
            loss.backward()
    else:
        g = full(g)
        loss = loss_fn(g)
        loss.backward()
    out = {"h": to_np(g.nodes["n1"].data["h"]), "loss": to_np(loss).reshape(1),
           "energy": to_np(g.nodes["g"].data["energy"]), "gradient": to_np(g.nodes["n1"].data["gradient"])}
    for lvl in ["n2", "n3", "n4", "n4_improper"]:
        out[f"{lvl}_idxs"] = to_np(g.nodes[lvl].data["idxs"])
        out[f"{lvl}_k"] = to_np(g.nodes[lvl].data["k"])
        out[f"{lvl}_x"] = to_np(g.nodes[lvl].data["x"])
        out[f"energy_{lvl}"] = to_np(g.nodes["g"].data[f"energy_{lvl}"])
        if lvl in ("n2", "n3"):
            out[f"{lvl}_eq"] = to_np(g.nodes[lvl].data["eq"])
    seen = set()
    gn = {}
    for k, p in model.named_parameters():          # named_parameters de-duplicates the gnn.blocks alias
        if p.grad is None:
            continue
        if grads == "all" or (callable(grads) and grads(k)):
            out["grad::" + k] = to_np(p.grad)
        gn[k] = float(p.grad.norm())
    out["grad_norm_keys"] = np.array(list(gn.keys()))
    out["grad_norm_vals"] = np.array(list(gn.values()), dtype=np.float64)
    return out, sd, g


def pack_inputs(mols):
    d = {"n_mols": np.array([len(mols)])}
    for i, m in enumerate(mols):
        for k, v in m.items():
            d[f"mol{i}::{k}"] = np.asarray(v)
    return d


def small_config(n_conv=0, gated=True, n_att=2):
    return dict(graph_node_features=32, in_feats=None,
                in_feat_name=["atomic_number", "partial_charge", "ring_encoding", "degree", "charge_model"], in_feat_dims={},
                gnn_width=64, gnn_attentional_layers=n_att, gnn_convolutions=n_conv, gnn_attention_heads=4,
                gnn_dropout_attention=0.3, gnn_dropout_initial=0.0, gnn_dropout_conv=0.1, gnn_dropout_final=0.1, parameter_dropout=0.5,
                bond_transformer_depth=2, bond_n_heads=4, bond_transformer_width=64, bond_symmetriser_depth=3, bond_symmetriser_width=32,
                angle_transformer_depth=2, angle_n_heads=4, angle_transformer_width=64, angle_symmetriser_depth=3, angle_symmetriser_width=32,
                proper_transformer_depth=2, proper_n_heads=4, proper_transformer_width=64, proper_symmetriser_depth=2, proper_symmetriser_width=32,
                improper_transformer_depth=1, improper_n_heads=4, improper_transformer_width=64, improper_symmetriser_depth=1,
                improper_symmetriser_width=32, n_periodicity_proper=6, n_periodicity_improper=3, gated_torsion=gated, wrong_symmetry=False,
                positional_encoding=True, layer_norm=True, self_interaction=True, learnable_statistics=False, torsion_cutoff=1e-4)


def save(name, cfg, mols, out, sd=None, extra=None):
    d = pack_inputs(mols)
    d.update({"out::" + k: v for k, v in out.items()})
    d["cfg_keys"] = np.array(list(cfg.keys()))
    d["cfg_vals"] = np.array([repr(v) for v in cfg.values()])
    if sd is not None:
        for k, v in sd.items():
            d["sd::" + k] = to_np(v)
    if extra:
        d.update(extra)
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **d)
    print("wrote", path, os.path.getsize(path))


def n_improper_centres(i):
    z, bonds, _ = pool_molecule(i)
    nd = tuple_indices.get_neighbor_dict([tuple(b) for b in bonds.tolist()], sort=True)
    return len(tuple_indices.improper_centres_from_bonds(bonds.tolist(), nd, z))


def pick_small(n, lo, hi, start=0, impropers=True):
    """n pool molecules with lo..hi atoms; impropers=True: only molecules WITH improper centres, False: only without."""
    ids = []
    for i in range(start, len(POOL["atom_ptr"]) - 1):
        na = POOL["atom_ptr"][i + 1] - POOL["atom_ptr"][i]
        if lo <= na <= hi and (n_improper_centres(i) > 0) == impropers:
            ids.append(i)
        if len(ids) == n:
            break
    return ids


def main():
    os.makedirs(OUT, exist_ok=True)
    loss_kwargs = dict(gradient_weight=0.8, energy_weight=1.0, param_weight=1e-3, proper_regularisation=1e-3,
                       improper_regularisation=1e-3)
    # ---- small config A: 2 attention blocks, gated torsions; 5 molecules, 4 conformations
    mols = build_inputs(pick_small(5, 8, 24), n_confs=4, seed=11)
    assert all(len(m["impropers"]) > 0 for m in mols)     # Q4: the reference loss is NaN otherwise
    cfg = small_config()
    out, sd, _ = run_reference(cfg, mols, 4, loss_kwargs=loss_kwargs)
    out_noise, _, _ = run_reference(cfg, mols, 4, state_dict=sd, loss_kwargs=loss_kwargs, noise=True, grads="none")
    # with improper_regularisation > 0 the reference's loss is NaN as soon as one molecule has no impropers
    mols_q4 = mols[:2] + build_inputs(pick_small(1, 5, 30, impropers=False), n_confs=4, seed=3)
    out_q4, _, _ = run_reference(cfg, mols_q4, 4, state_dict=sd, loss_kwargs=loss_kwargs, grads="none")
    extra = {"q4::reference_loss_with_improper_free_molecule": out_q4["loss"],
             "noise::energy_maxabs": np.array([np.abs(out_noise["energy"] - out["energy"]).max()]),
             "noise::gradient_maxabs": np.array([np.abs(out_noise["gradient"] - out["gradient"]).max()]),
             "loss_kwargs_keys": np.array(list(loss_kwargs.keys())), "loss_kwargs_vals": np.array(list(loss_kwargs.values()))}
    save("ref_small_att.npz", cfg, mols, out, sd, extra)

    # ---- small config B: 1 SAGE conv block + 1 attention block, ungated torsions, padded (dummy) conformations
    mols = build_inputs(pick_small(4, 10, 30, start=40), n_confs=5, seed=23, charge_model="amber99", pad_confs_of={1: 2, 3: 4})
    cfg = small_config(n_conv=1, gated=False, n_att=1)
    lk = dict(gradient_weight=0.5, energy_weight=1.0, param_weight=0.0)
    out, sd, g = run_reference(cfg, mols, 5, loss_kwargs=lk, with_param_refs=False)
    extra = {"is_dummy": to_np(g.nodes["g"].data["is_dummy"]),
             "loss_kwargs_keys": np.array(list(lk.keys())), "loss_kwargs_vals": np.array(list(lk.values()))}
    save("ref_small_conv.npz", cfg, mols, out, sd, extra)

    # ---- production config: weights regenerated from the state-dict keys on both sides; only outputs are stored
    mols = build_inputs(pick_small(3, 12, 26, start=100), n_confs=4, seed=5)
    cfg = get_default_model_config()
    out, sd, _ = run_reference(cfg, mols, 4, loss_kwargs=loss_kwargs, grads="norms")
    extra = {"loss_kwargs_keys": np.array(list(loss_kwargs.keys())), "loss_kwargs_vals": np.array(list(loss_kwargs.values()))}
    save("ref_prod.npz", cfg, mols, out, None, extra)

    # ---- collate (N1): the reference's collate_fn on molecules with different conformation counts, seeded sub-sampling
    from grappa.data.GraphDataLoader import get_collate_fn as ref_get_collate_fn
    mols = build_inputs(pick_small(4, 8, 20, start=300), n_confs=6, seed=91, pad_confs_of={0: 9, 2: 3, 3: 7})
    col = {}
    for strategy in (4, "min", "max", "mean"):
        graphs = [ref_graph(m, m["xyz"].shape[1], False)[0] for m in mols]
        torch.manual_seed(1234)
        gb, names = ref_get_collate_fn(conf_strategy=strategy)([(g, f"ds{i % 2}") for i, g in enumerate(graphs)])
        col[f"{strategy}::xyz"] = to_np(gb.nodes["n1"].data["xyz"])
        col[f"{strategy}::is_dummy"] = to_np(gb.nodes["g"].data["is_dummy"])
        col[f"{strategy}::energy_ref"] = to_np(gb.nodes["g"].data["energy_ref"])
        col[f"{strategy}::gradient_ref"] = to_np(gb.nodes["n1"].data["gradient_ref"])
        col[f"{strategy}::n4_idxs"] = to_np(gb.nodes["n4"].data["idxs"])
    save("ref_collate.npz", {}, mols, col)

    # ---- MolData .npz schema (N2): the reference's MolData.from_dict(...).to_dgl() on a synthetic record
    from grappa.data import MolData as RefMolData
    m = build_inputs(pick_small(1, 10, 18, start=400), n_confs=5, seed=17)[0]
    gref, mol = ref_graph(m, 5, False)
    rng = np.random.default_rng(4)
    n, C = len(m["z"]), 5
    nb, na, npr, ni = [gref.num_nodes(l) for l in ("n2", "n3", "n4", "n4_improper")]
    record = dict(mol.to_dict())
    record.update(xyz=rng.normal(size=(C, n, 3)).astype(np.float32), energy=rng.normal(size=C).astype(np.float32),
                  gradient=rng.normal(size=(C, n, 3)).astype(np.float32), energy_ref=rng.normal(size=C).astype(np.float32),
                  gradient_ref=rng.normal(size=(C, n, 3)).astype(np.float32), mol_id=np.array("mol-0"), smiles=np.array("C"),
                  bond_k=rng.uniform(300, 900, nb).astype(np.float32), bond_eq=rng.uniform(1, 1.5, nb).astype(np.float32),
                  angle_k=rng.uniform(50, 150, na).astype(np.float32), angle_eq=rng.uniform(1.7, 2.2, na).astype(np.float32),
                  proper_ks=rng.uniform(0, 2, (npr, 4)).astype(np.float32), proper_phases=(rng.integers(0, 2, (npr, 4)) * np.pi).astype(np.float32),
                  improper_ks=rng.uniform(0, 5, (ni, 6)).astype(np.float32), improper_phases=(rng.integers(0, 2, (ni, 6)) * np.pi).astype(np.float32),
                  energy_reference_ff=rng.normal(size=C).astype(np.float32), gradient_reference_ff=rng.normal(size=(C, n, 3)).astype(np.float32))
    # the reference's Molecule.from_dict re-derives ring_encoding / degree through RDKit (absent offline); the record already
    # carries both features, so the RDKit call is skipped for features that are present (the only runtime patch besides the noise)
    _orig_add = RefMolecule.add_features

    def _add_features_without_rdkit(self, feat_names=("ring_encoding", "degree", "mass"), **kw):
        names = [feat_names] if isinstance(feat_names, str) else list(feat_names)
        names = [f for f in names if not (f in ("ring_encoding", "degree") and f in self.additional_features)]
        return _orig_add(self, names, **kw) if names else None

    RefMolecule.add_features = _add_features_without_rdkit
    try:
        gd = RefMolData.from_dict(record).to_dgl()
    finally:
        RefMolecule.add_features = _orig_add
    rec_out = {"record::" + k: np.asarray(v) for k, v in record.items()}
    for nt in ("g", "n1", "n2", "n3", "n4", "n4_improper"):
        for k, v in gd.nodes[nt].data.items():
            rec_out[f"graph::{nt}::{k}"] = to_np(v)
    np.savez_compressed(os.path.join(OUT, "ref_moldata.npz"), **rec_out)
    print("wrote ref_moldata.npz")

    # ---- Energy only, on "classical" parameters (suffix _ref), incl. the torsion offset option
    mols = build_inputs(pick_small(4, 8, 40, start=200), n_confs=6, seed=77)
    graphs = [ref_graph(m, 6, True, nan_refs=False)[0] for m in mols]
    g = ref_dgl_utils.batch(graphs)
    with zero_dihedral_noise():
        g = RefEnergy(suffix="_ref", write_suffix="_classical", gradients=True)(g)
        g = RefEnergy(suffix="_ref", write_suffix="_offs", gradients=True, offset_torsion=True)(g)
    out = {"energy": to_np(g.nodes["g"].data["energy_classical"]), "gradient": to_np(g.nodes["n1"].data["gradient_classical"]),
           "energy_offs": to_np(g.nodes["g"].data["energy_offs"])}
    for lvl in ["n2", "n3", "n4", "n4_improper"]:
        out[f"{lvl}_x"] = to_np(g.nodes[lvl].data["x"])
        out[f"{lvl}_tuple_energy"] = to_np(g.nodes[lvl].data["energy_classical"])
        out[f"{lvl}_k_ref"] = to_np(g.nodes[lvl].data["k_ref"])
        if lvl in ("n2", "n3"):
            out[f"{lvl}_eq_ref"] = to_np(g.nodes[lvl].data["eq_ref"])
    save("ref_energy.npz", {}, mols, out)


def make_eval_golden():
    """FastEvaluator (N4): the reference's own evaluator on two batches (one with padded dummy conformations), three dataset
    names.  matplotlib (plot helpers only, never called here) is absent offline: an empty module object satisfies the import."""
    import types
    for name in ("matplotlib", "matplotlib.pyplot"):
        sys.modules.setdefault(name, types.ModuleType(name))
    from grappa.training.evaluation import FastEvaluator as RefFastEvaluator
    cfg = small_config(n_conv=1, gated=False, n_att=1)
    ev, ev_nograd, ev_cl = RefFastEvaluator(), RefFastEvaluator(gradients=False), RefFastEvaluator(log_classical_values=True)
    d, batches = {}, []
    specs = [dict(ids=pick_small(4, 10, 30, start=40), n_confs=5, seed=23, pad={1: 2, 3: 4}, names=["dsA", "dsB", "dsA", "dsC"]),
             dict(ids=pick_small(3, 8, 24, start=500), n_confs=7, seed=29, pad=None, names=["dsB", "dsB", "dsA"])]
    sd = None
    for bi, sp in enumerate(specs):
        mols = build_inputs(sp["ids"], n_confs=sp["n_confs"], seed=sp["seed"], charge_model="amber99", pad_confs_of=sp["pad"])
        out, sd, g = run_reference(cfg, mols, sp["n_confs"], state_dict=sd, loss_kwargs=dict(gradient_weight=0.5, energy_weight=1.0, param_weight=0.0),
                                   with_param_refs=False, grads="none")
        # classical force-field values: an Energy pass on seeded 'classical' parameters (suffix _ref), written with the
        # suffix the evaluator reads (_classical_ff)
        rng = np.random.default_rng(100 + bi)
        with torch.no_grad(), zero_dihedral_noise():
            for lvl, scale in (("n2", 0.05), ("n3", 0.05)):
                for kk in ("k", "eq"):
                    v = g.nodes[lvl].data[kk].detach()
                    g.nodes[lvl].data[kk + "_ref"] = v * torch.from_numpy(1.0 + scale * rng.normal(size=tuple(v.shape))).float()
            for lvl in ("n4", "n4_improper"):
                v = g.nodes[lvl].data["k"].detach()
                g.nodes[lvl].data["k_ref"] = v + torch.from_numpy(0.1 * rng.normal(size=tuple(v.shape))).float()
        g.nodes["n1"].data["xyz"] = g.nodes["n1"].data["xyz"].detach().clone()       # a fresh leaf: the first pass's graph is freed
        with zero_dihedral_noise():
            g = RefEnergy(suffix="_ref", write_suffix="_classical_ff", gradients=True)(g)
        with torch.no_grad():
            ev.step(g, sp["names"])
            ev_nograd.step(g, sp["names"])
            ev_cl.step(g, sp["names"])
        d[f"b{bi}::energy_classical_ff"] = to_np(g.nodes["g"].data["energy_classical_ff"])
        d[f"b{bi}::gradient_classical_ff"] = to_np(g.nodes["n1"].data["gradient_classical_ff"])
        d[f"b{bi}::energy"] = to_np(g.nodes["g"].data["energy"])
        d[f"b{bi}::energy_ref"] = to_np(g.nodes["g"].data["energy_ref"])
        d[f"b{bi}::is_dummy"] = to_np(g.nodes["g"].data["is_dummy"])
        d[f"b{bi}::gradient"] = to_np(g.nodes["n1"].data["gradient"])
        d[f"b{bi}::gradient_ref"] = to_np(g.nodes["n1"].data["gradient_ref"])
        d[f"b{bi}::atoms_per_mol"] = np.array([len(m["z"]) for m in mols])
        d[f"b{bi}::dsnames"] = np.array(sp["names"])
    for tag, e in (("full", ev), ("nograd", ev_nograd), ("classical", ev_cl)):
        m = e.pool()
        for ds, mm in m.items():
            for k, v in mm.items():
                d[f"metrics::{tag}::{ds}::{k}"] = np.array([np.nan if v is None else float(v)])
    d["n_batches"] = np.array([len(specs)])
    np.savez_compressed(os.path.join(OUT, "ref_eval.npz"), **d)
    print("wrote ref_eval.npz", {k: float(v[0]) for k, v in d.items() if k.startswith("metrics::full")})


def make_tuple_golden():
    """angles / propers of 36 pool molecules from the reference's own enumeration (utils/tuple_indices.py:7-63), incl. scrambled
    bond order and orientation (first-appearance order != index order) and sparse atom ids: pins the ROW ORDER of the native
    enumerator (include/grappa_host.h grappa_topo_enumerate)."""
    from grappa.utils import tuple_indices as ref_ti
    rng = np.random.default_rng(0)
    d = {}
    n_pool = len(POOL["atom_ptr"]) - 1
    ids = list(range(0, n_pool, 97))[:36]
    for j, i in enumerate(ids):
        _, bonds, _ = pool_molecule(i)
        bonds = bonds.copy()
        if j % 3 == 1:
            bonds = bonds[rng.permutation(len(bonds))]
            flip = rng.random(len(bonds)) < 0.5
            bonds[flip] = bonds[flip][:, ::-1]
        if j % 3 == 2:
            bonds = bonds * 3 + 5
        out = ref_ti.get_idx_tuples([tuple(int(x) for x in b) for b in bonds])
        d[f"m{j}::bonds"] = bonds.astype(np.int64)
        d[f"m{j}::angles"] = np.asarray(out["angles"], dtype=np.int64).reshape(-1, 3)
        d[f"m{j}::propers"] = np.asarray(out["propers"], dtype=np.int64).reshape(-1, 4)
    d["n"] = np.array([len(ids)])
    np.savez_compressed(os.path.join(OUT, "ref_tuples.npz"), **d)
    print("wrote ref_tuples.npz")


def make_predict_golden():
    """P1: the reference's own `Grappa.predict` (grappa.py:36-57 -> data/Parameters.py:62-140) on four pool molecules with the
    production config (weights from the state-dict keys), plus `Parameters.from_dgl` on a hand-filled graph that holds exact
    zeros and both signs in the torsion tables (the >= / > asymmetry) and the two raise conditions."""
    from grappa.grappa import Grappa as RefGrappa
    from grappa.data import Parameters as RefParameters
    cfg = get_default_model_config()
    torch.manual_seed(0)
    model = RefGrappaModel(**cfg)
    model.load_state_dict(keyed_state_dict(model))
    wrapper = RefGrappa(model, device="cpu")
    mols = build_inputs(pick_small(2, 10, 22, start=700) + pick_small(1, 30, 45, start=900) + pick_small(1, 6, 12, start=1500, impropers=False),
                        n_confs=1, seed=41)
    d = pack_inputs(mols)
    fields = ["atoms", "bonds", "bond_k", "bond_eq", "angles", "angle_k", "angle_eq", "propers", "proper_ks", "proper_phases",
              "impropers", "improper_ks", "improper_phases"]
    for i, m in enumerate(mols):
        _, mol = ref_graph(m, 1, False)
        params = wrapper.predict(mol)
        for f in fields:
            d[f"pred{i}::{f}"] = np.asarray(getattr(params, f))
    # from_dgl alone, on a graph whose parameter tables are set by hand
    g, _ = ref_graph(mols[0], 1, False)
    rng = np.random.default_rng(5)
    for lvl, n in (("n4", 6), ("n4_improper", 3)):
        k = rng.normal(0, 1, size=(g.num_nodes(lvl), n)).astype(np.float32)
        k[::2, 0] = 0.0
        k[1::3, 1] = -0.0
        g.nodes[lvl].data["k"] = torch.from_numpy(k)
        d[f"hand::{lvl}_k"] = k
    for lvl, lo, hi in (("n2", 0.9, 1.6), ("n3", 1.6, 2.2)):
        g.nodes[lvl].data["k"] = torch.from_numpy(rng.uniform(50, 900, g.num_nodes(lvl)).astype(np.float32))
        g.nodes[lvl].data["eq"] = torch.from_numpy(rng.uniform(lo, hi, g.num_nodes(lvl)).astype(np.float32))
        d[f"hand::{lvl}_k"], d[f"hand::{lvl}_eq"] = to_np(g.nodes[lvl].data["k"]), to_np(g.nodes[lvl].data["eq"])
    params = RefParameters.from_dgl(g)
    for f in fields:
        d[f"hand::out::{f}"] = np.asarray(getattr(params, f))
    raised = []
    for lvl, bad in (("n3", np.pi / 180 * 44.9), ("n2", 0.499)):
        keep = g.nodes[lvl].data["eq"].clone()
        g.nodes[lvl].data["eq"] = keep.clone()
        g.nodes[lvl].data["eq"][1] = bad
        try:
            RefParameters.from_dgl(g)
            raised.append("none")
        except Exception as e:  # noqa: BLE001
            raised.append(type(e).__name__)
        g.nodes[lvl].data["eq"][1] = bad * 1.01          # just above the limit: accepted
        RefParameters.from_dgl(g)
        g.nodes[lvl].data["eq"] = keep
    d["hand::raises"] = np.array(raised)
    np.savez_compressed(os.path.join(OUT, "ref_predict.npz"), **d)
    print("wrote ref_predict.npz; raise conditions ->", raised)


def make_options_golden():
    """the reference with its optional sub-modules switched off (layer_norm=False, self_interaction=False; models/grappa.py:51,
    graph_attention.py:255-310, :366-412, network_utils.py:36-48, :98-114) and with learnable statistics (final_layer.py:21-44, :64-88,
    interaction_parameters.py:463-470): 1 SAGE + 1 attention block, ungated torsions"""
    mols = build_inputs(pick_small(4, 10, 30, start=40), n_confs=5, seed=23, charge_model="amber99")
    lk = dict(gradient_weight=0.5, energy_weight=1.0, param_weight=0.0)
    for name, opts in (("ref_small_nonorm.npz", dict(layer_norm=False)), ("ref_small_nosi.npz", dict(self_interaction=False)),
                       ("ref_small_learnstats.npz", dict(learnable_statistics=True))):
        cfg = small_config(n_conv=1, gated=False, n_att=1)
        cfg.update(opts)
        out, sd, g = run_reference(cfg, mols, 5, loss_kwargs=lk, with_param_refs=False)
        extra = {"is_dummy": to_np(g.nodes["g"].data["is_dummy"]),
                 "loss_kwargs_keys": np.array(list(lk.keys())), "loss_kwargs_vals": np.array(list(lk.values()))}
        save(name, cfg, mols, out, sd, extra)


def tiny_config(**opts):
    """a model small enough that a fixture with its weights and every parameter gradient stays well under 1 MB"""
    cfg = small_config(n_conv=0, gated=False, n_att=1)
    cfg.update(graph_node_features=16, gnn_width=32, gnn_attention_heads=2)
    for lvl in ("bond", "angle", "proper", "improper"):
        cfg.update({f"{lvl}_transformer_depth": 1, f"{lvl}_n_heads": 2, f"{lvl}_transformer_width": 32, f"{lvl}_symmetriser_depth": 2,
                    f"{lvl}_symmetriser_width": 16})
    cfg.update(opts)
    return cfg


def make_variants_golden(only=None):
    """constructor / Energy variants of the reference that had no reference-generated fixture (VERDICT r3 missing #2):
    wrong_symmetry=True (models/interaction_parameters.py:502-507: six permutations of the improper tokens, positional code [0,0,1,0]),
    harmonic_gate=True (:257-264, :352-360: a third output column per bond / angle that the written k ignores), n_periodicity_proper=3
    (experiments/train-grappa-1.2.1/grappa_config.yaml:98-99) and Energy(offset_torsion=True) (models/energy.py:79) end to end,
    i.e. with the loss and every parameter gradient behind it.  Round 4, second batch: positional_encoding=False (:166-176, no position code on
    the tokens) with two input features only and a non-default torsion_cutoff -- three more options without a fixture of their own.
    `only`: write that file alone."""
    mols = build_inputs(pick_small(4, 9, 22, start=60), n_confs=4, seed=31, charge_model="amber99")
    lk = dict(gradient_weight=0.8, energy_weight=1.0, param_weight=1e-3, proper_regularisation=1e-3, improper_regularisation=1e-3)
    for name, opts, ek in (("ref_tiny_wrongsym.npz", dict(wrong_symmetry=True), None),
                           ("ref_tiny_harmonic_gate.npz", dict(harmonic_gate=True), None),
                           ("ref_tiny_nper3.npz", dict(n_periodicity_proper=3, gated_torsion=True), None),
                           ("ref_tiny_offset_torsion.npz", dict(), dict(offset_torsion=True)),
                           ("ref_tiny_nopos.npz", dict(positional_encoding=False, in_feat_name=["atomic_number", "partial_charge"], torsion_cutoff=1e-2), None)):
        if only and name != only:
            continue
        cfg = tiny_config(**opts)
        out, sd, g = run_reference(cfg, mols, 4, loss_kwargs=lk, energy_kwargs=ek)
        extra = {"loss_kwargs_keys": np.array(list(lk.keys())), "loss_kwargs_vals": np.array(list(lk.values())),
                 "energy_kwargs_keys": np.array(list((ek or {}).keys())), "energy_kwargs_vals": np.array([repr(v) for v in (ek or {}).values()])}
        save(name, cfg, mols, out, None, extra)         # weights: golden_utils.keyed_state_dict on both sides (derived from the state-dict keys)


def make_prod_nper3_golden():
    """production widths in the SHIPPED experiment's setting (reference experiments/train-grappa-1.2.1/grappa_config.yaml:98-99:
    n_periodicity_proper = 3, n_periodicity_improper = 3): outputs, every gradient norm, and the FULL gradients of the last symmetriser
    layer of the four writer heads (`symmetriser.mlp.2.*`: the layers whose width follows n_periodicity) -- VERDICT r4 item 6c"""
    loss_kwargs = dict(gradient_weight=0.8, energy_weight=1.0, param_weight=1e-3, proper_regularisation=1e-3, improper_regularisation=1e-3)
    mols = build_inputs(pick_small(3, 12, 26, start=140), n_confs=4, seed=9)
    cfg = get_default_model_config()
    cfg["n_periodicity_proper"], cfg["n_periodicity_improper"] = 3, 3
    out, sd, _ = run_reference(cfg, mols, 4, loss_kwargs=loss_kwargs, grads=lambda k: "symmetriser.mlp.2." in k)
    assert sum(k.startswith("grad::") for k in out) >= 16, [k for k in out if k.startswith("grad::")]
    extra = {"loss_kwargs_keys": np.array(list(loss_kwargs.keys())), "loss_kwargs_vals": np.array(list(loss_kwargs.values()))}
    save("ref_prod_nper3.npz", cfg, mols, out, None, extra)


def make_dataset_golden():
    """the reference's dataset-level logic (data/Dataset.py:80-112 `split`, :236-258 k-fold / partition splits through
    utils/torch_utils.py:11-135, :141-345, :259-294 `where` / `shuffle` / `subsampled`) on synthetic (mol_id, subdataset) lists with
    duplicated ids across subdatasets: the graphs are placeholders (the index in the list), so what is pinned is which item goes where."""
    import json
    from grappa.data import Dataset as RefDataset
    rng = np.random.default_rng(9)
    names, ids = [], []
    for ds, n in (("spice-dipeptide", 37), ("spice-pubchem", 61), ("rna-diverse", 9), ("gen2", 23)):
        for i in range(n):
            names.append(ds)
            ids.append(f"{ds}-{i:03d}")
    # duplicated molecules: the same id in two subdatasets (gen2 / spice-pubchem) and twice inside one (conformation sets)
    for i in range(7):
        names.append("gen2")
        ids.append(f"spice-pubchem-{3 * i:03d}")
    for i in range(3):
        names.append("spice-dipeptide")
        ids.append(f"spice-dipeptide-{5 * i:03d}")
    order = rng.permutation(len(ids))
    ids, names = [ids[i] for i in order], [names[i] for i in order]
    ds = RefDataset(graphs=list(range(len(ids))), mol_ids=list(ids), subdataset=list(names))
    out = {"mol_ids": np.array(ids), "subdataset": np.array(names)}
    cases = {}
    cases["partition_tuple_seed0"] = ds.calc_split_ids(partition=(0.8, 0.1, 0.1), seed=0)
    cases["partition_tuple_seed3"] = ds.calc_split_ids(partition=(0.6, 0.2, 0.2), seed=3)
    cases["partition_dict"] = ds.calc_split_ids(partition=((0.8, 0.1, 0.1), {"rna-diverse": (1.0, 0.0, 0.0), "spice-dipeptide": (0.8, 0.1, 0.1)}), seed=1)
    existing = {"train": [i for i in ids if i.startswith("gen2-00")], "val": ["spice-pubchem-000"], "test": ["rna-diverse-001", "not-in-this-dataset"]}
    cases["existing_split"] = ds.calc_split_ids(partition=(0.8, 0.1, 0.1), seed=2, existing_split={k: list(v) for k, v in existing.items()})
    out["existing_split_input"] = np.array(json.dumps(existing))
    for k, v in cases.items():
        out["calc::" + k] = np.array(json.dumps(v))
    folds = ds.get_k_fold_split_ids(k=5, seed=4)
    out["kfold::k5_seed4"] = np.array(json.dumps(folds))
    out["kfold::k4_seed1_folds2"] = np.array(json.dumps(ds.get_k_fold_split_ids(k=4, seed=1, num_folds=2)))
    # Dataset.split: items by membership of their id; ids in no list go to the test set
    sp = cases["partition_tuple_seed0"]
    tr, vl, te = ds.split(sp["train"], sp["val"][: len(sp["val"]) // 2], [])
    out["split::train"], out["split::val"], out["split::test"] = np.array(tr.graphs), np.array(vl.graphs), np.array(te.graphs)
    out["shuffle::seed5"] = np.array(RefDataset(list(range(len(ids))), list(ids), list(names)).shuffle(seed=5).graphs)
    out["subsampled::0.3_seed2"] = np.array(RefDataset(list(range(len(ids))), list(ids), list(names)).subsampled(0.3, seed=2).graphs)
    out["slice::10_20"] = np.array(ds.slice(10, 20).graphs)
    np.savez_compressed(os.path.join(OUT, "ref_dataset.npz"), **out)
    print("wrote ref_dataset.npz:", {k: {s: len(v[s]) for s in v} for k, v in cases.items()})


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "dataset":
        make_dataset_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "options":
        make_options_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "variants":
        make_variants_golden(sys.argv[2] if len(sys.argv) > 2 else None)
    elif len(sys.argv) > 1 and sys.argv[1] == "prod_nper3":
        make_prod_nper3_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "eval":
        make_eval_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "tuples":
        make_tuple_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "predict":
        make_predict_golden()
    else:
        main()
        make_eval_golden()
        make_tuple_golden()
        make_predict_golden()
        make_options_golden()
        make_variants_golden()
        make_dataset_golden()
