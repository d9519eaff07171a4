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
