"""CPU: the index tables of the (atom, position) formulation of a writer's first layer (batch.BatchPlan.position_tables, used by
ops.ProjFirstLayerFn): every token row appears exactly once in the inverse incidence of the table, under the table row it reads."""
import numpy as np
import torch

import golden_utils as gu
from grappa_amd.constants import LEVEL_ARITY


def test_position_tables_are_consistent_inverses():
    fx = gu.load("ref_small_att.npz")
    cfg = gu.config_of(fx)
    g = gu.build_batch(gu.molecules_of(fx), 3, False, (cfg["n_periodicity_proper"], cfg["n_periodicity_improper"]))
    plan = g.plan()
    N = plan.N
    for lvl in ("n2", "n3", "n4", "n4_improper"):
        s, T = LEVEL_ARITY[lvl], plan.T[lvl]
        idx_id, invid_ptr, invid_rows, idx_tab, invtab_ptr, invtab_rows = (t.numpy() for t in plan.position_tables(lvl))
        idx = plan.idx32[lvl].numpy()
        # the table layout: row pos*N + n holds atom n at position pos
        assert idx_id.shape == (N, s) and np.array_equal(idx_id, np.repeat(np.arange(N)[:, None], s, 1))
        assert np.array_equal(invid_ptr, np.arange(N + 1) * s)
        assert np.array_equal(invid_rows.reshape(N, s), np.arange(s)[None, :] * N + np.arange(N)[:, None])
        # tokens read table rows
        assert idx_tab.shape == (T, s) and np.array_equal(idx_tab, idx + np.arange(s)[None, :] * N)
        # inverse: table row -> token rows pos*T + t, ascending, every token exactly once
        assert invtab_ptr.shape == (s * N + 1,) and invtab_ptr[0] == 0 and invtab_ptr[-1] == s * T
        assert np.array_equal(np.sort(invtab_rows), np.arange(s * T))
        token_table_row = idx_tab.T.reshape(-1)                         # token row r = pos*T + t reads this table row
        for row in range(s * N):
            toks = invtab_rows[invtab_ptr[row]:invtab_ptr[row + 1]]
            assert np.all(token_table_row[toks] == row)
            assert np.all(np.diff(toks) > 0)                            # ascending: a fixed summation order in the backward gather
        assert plan.position_tables(lvl)[0] is plan.position_tables(lvl)[0]      # cached
