"""CPU: the native index-plan builder (libgrappa_host.so grappa_plan_build, include/grappa_host.h) against its numpy definition
(grappa_amd/batch.py _plan_arrays_numpy), element for element, on pool molecules, batches, empty levels and malformed graphs."""
import numpy as np
import pytest
import torch

import importlib

B = importlib.import_module("grappa_amd.batch")          # (grappa_amd.batch is also the name of the batching function)
from grappa_amd.datasets import build_batch_from_pool, molecule_from_pool


def _both(g):
    N = g.num_nodes("n1")
    src, dst = g._src.numpy().astype(np.int64), g._dst.numpy().astype(np.int64)
    idx = [g._data[lvl]["idxs"].numpy().astype(np.int32).reshape(-1, B.LEVEL_ARITY[lvl]) for lvl in B.TUPLE_LEVELS]
    return B._plan_arrays_native(N, src, dst, idx), B._plan_arrays_numpy(N, src, dst, idx)


@pytest.mark.parametrize("ids", [[0], [17], [3, 50, 900], list(range(200, 232))])
def test_native_plan_equals_the_numpy_definition(ids):
    g = build_batch_from_pool(ids, n_confs=1, seed=0) if len(ids) > 1 else molecule_from_pool(ids[0]).to_dgl()
    nat, ref = _both(g)
    assert nat is not None and set(nat) == set(ref)
    for k in ref:
        if k == "max_degree":
            assert nat[k] == ref[k]
        else:
            assert nat[k].dtype == np.int32 and np.array_equal(nat[k], ref[k]), k


def test_plan_views_share_one_buffer_and_serve_the_model_paths():
    g = build_batch_from_pool([5, 6, 7], n_confs=1, seed=0)
    plan = g.plan()
    base = plan.indptr.untyped_storage().data_ptr()
    for t in (plan.indices, plan.rev, plan.inc_code, plan.idx32["n4"], plan.inv_rows["n3"], plan.mol_ptr["n2"], plan.atom_molptr):
        assert t.untyped_storage().data_ptr() == base and t.dtype == torch.int32 and t.is_contiguous()
        assert t.data_ptr() % 16 == 0 or t.numel() == 0
    assert plan.idx32["n4"].shape == (plan.T["n4"], 4) and int(plan.indptr[-1]) == plan.E


def test_malformed_graphs_raise_as_before():
    m = molecule_from_pool(3).to_dgl()
    one_way = B.MolBatch(m._src[: m._src.shape[0] // 2], m._dst[: m._dst.shape[0] // 2], m._data, m._bnn)
    with pytest.raises(ValueError, match="both directions"):
        one_way.plan()
    bad = B.MolBatch(m._src, m._dst, {nt: dict(d) for nt, d in m._data.items()}, m._bnn)
    bad._data["n3"]["idxs"] = bad._data["n3"]["idxs"].clone()
    bad._data["n3"]["idxs"][0, 0] = 10 ** 6
    with pytest.raises(AssertionError, match="Encountered idxs"):
        bad.plan()


def test_native_connected_components_against_union_find():
    """grappa_components (the water guard of Grappa.predict): label = smallest atom index of the component, on random forests with extra
    edges, isolated atoms, one direction or both; out-of-range atoms are refused"""
    from grappa_amd import _hostlib
    rng = np.random.default_rng(0)
    for n, e in [(1, 0), (5, 0), (40, 39), (200, 150), (1000, 1400)]:
        src = rng.integers(0, n, size=e).astype(np.int64)
        dst = rng.integers(0, n, size=e).astype(np.int64)
        if e and e % 2 == 0:                                  # both directions, as MolBatch stores them
            src, dst = np.concatenate([src, dst]), np.concatenate([dst, src])
        parent = list(range(n))

        def find(a):
            while parent[a] != a:
                parent[a] = parent[parent[a]]
                a = parent[a]
            return a
        for a, b in zip(src.tolist(), dst.tolist()):
            ra, rb = find(a), find(b)
            if ra != rb:
                parent[max(ra, rb)] = min(ra, rb)
        want = np.array([find(i) for i in range(n)], dtype=np.int32)
        # (the smallest index of a component is its root under "the smaller index stays the root")
        got = _hostlib.components(n, src, dst)
        assert got.dtype == np.int32 and np.array_equal(got, want), (n, e)
    with pytest.raises(RuntimeError):
        _hostlib.components(3, np.array([0, 5]), np.array([1, 2]))


def test_water_guard_still_fires_through_the_native_components():
    from grappa_amd.molecule import Molecule
    # two waters in one graph: disconnected, every component {H, O} only
    mol = Molecule(atoms=[1, 2, 3, 4, 5, 6], bonds=[(1, 2), (1, 3), (4, 5), (4, 6)], impropers=[], atomic_numbers=[8, 1, 1, 8, 1, 1],
                   partial_charges=[-0.8, 0.4, 0.4, -0.8, 0.4, 0.4])
    g = mol.to_dgl(max_element=53, exclude_feats=[])
    with pytest.raises(Exception):
        B.check_disconnected_graphs(g)
    # the reference's literal check (Z - 1 compared with {1, 8}, utils/dgl_utils.py:231-234) never fires on water: the switch reproduces that
    B.check_disconnected_graphs(g, print_information=False, reference_water_guard=True)
    # ... and fires where the reference's would: a three-atom component of He and F only
    hef = Molecule(atoms=[1, 2, 3], bonds=[(1, 2), (1, 3)], impropers=[], atomic_numbers=[9, 2, 2], partial_charges=[0., 0., 0.])
    with pytest.raises(ValueError):
        B.check_disconnected_graphs(hef.to_dgl(max_element=53, exclude_feats=[]), print_information=False, reference_water_guard=True)


@pytest.mark.parametrize("ids", [[0], [17], [3, 50, 900]])
def test_native_position_tables_equal_the_numpy_definition(ids, monkeypatch):
    """grappa_position_tables against batch._position_tables' numpy expressions, level by level (incl. an empty level)"""
    g = build_batch_from_pool(ids, n_confs=1, seed=0) if len(ids) > 1 else molecule_from_pool(ids[0]).to_dgl()
    got = {lvl: g.plan().position_tables(lvl) for lvl in B.TUPLE_LEVELS}
    monkeypatch.setenv("GRAPPA_HOST_PLAN", "numpy")
    g2 = build_batch_from_pool(ids, n_confs=1, seed=0) if len(ids) > 1 else molecule_from_pool(ids[0]).to_dgl()
    want = {lvl: g2.plan().position_tables(lvl) for lvl in B.TUPLE_LEVELS}
    for lvl in B.TUPLE_LEVELS:
        assert len(got[lvl]) == len(want[lvl]) == 6
        for a, b in zip(got[lvl], want[lvl]):
            assert a.dtype == b.dtype == torch.int32 and a.shape == b.shape and torch.equal(a, b), lvl


def test_native_position_tables_refuse_bad_indices():
    from grappa_amd import _hostlib
    with pytest.raises(RuntimeError):
        _hostlib.position_tables(3, np.array([[0, 7]], dtype=np.int32))
    flat, parts = _hostlib.position_tables(4, np.zeros((0, 3), dtype=np.int32))         # an empty level
    assert [n for _o, n in parts] == [12, 5, 12, 0, 13, 0] and flat.dtype == np.int32
