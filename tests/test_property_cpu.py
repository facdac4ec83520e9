"""Property tests (hypothesis) of the host-side plumbing on random small graphs: whatever the edge rows,
the CSR is the binarised symmetrised adjacency, every plan covers every (kept) entry exactly once within
the segment / slice limits, and the C-ABI host builders agree with the torch builders."""
import ctypes as C

import numpy as np
import torch
from hypothesis import given, settings, strategies as st

from test_host_cpu import _check_plan, _host_arr


@st.composite
def edge_rows(draw):
    n = draw(st.integers(2, 40))
    e = draw(st.integers(0, 4 * n))
    src = draw(st.lists(st.integers(0, n - 1), min_size=e, max_size=e))
    dst = draw(st.lists(st.integers(0, n - 1), min_size=e, max_size=e))
    return n, np.array(src, dtype=np.int64), np.array(dst, dtype=np.int64)


@settings(max_examples=60, deadline=None)
@given(edge_rows(), st.integers(1, 9), st.sampled_from([1, 8, 16]))
def test_graph_and_plans_on_random_edge_rows(rows, seg_len, n_slices):
    from disenlink_amd import _lib
    from disenlink_amd.graph import CsrPlan, Graph
    n, src, dst = rows
    G = Graph.from_edge_rows(torch.from_numpy(src), torch.from_numpy(dst), n, seg_len=seg_len)
    dense = np.zeros((n, n), bool)
    dense[src, dst] = True
    dense |= dense.T
    rowptr, col = G.rowptr.numpy(), G.col.numpy()
    r, c = np.nonzero(dense)
    assert np.array_equal(col, c) and np.array_equal(np.diff(rowptr), np.bincount(r, minlength=n))
    if col.size:
        row_of = np.repeat(np.arange(n), np.diff(rowptr))
        rev = G.rev.numpy()
        assert np.array_equal(row_of[rev], col) and np.array_equal(col[rev], row_of)
    _check_plan(G.plan, rowptr, seg_len)
    sliced = CsrPlan.build(G.rowptr, G.col, n, seg_len=seg_len, n_slices=n_slices)
    _check_plan(sliced, rowptr, seg_len, col_slices=n_slices)
    # C ABI host builders: identical arrays
    lib = _lib.load()
    hc = _lib.DlHostCsr()
    assert lib.dl_host_csr_from_edges(src.ctypes.data, dst.ctypes.data, src.size, n, 1, C.byref(hc)) == 0
    try:
        assert np.array_equal(_host_arr(hc.rowptr, n + 1), rowptr) and np.array_equal(_host_arr(hc.col, hc.n_entries), col)
        hp = _lib.DlHostPlan()
        rp32, c32 = rowptr.astype(np.int32), col.astype(np.int32)
        assert lib.dl_host_plan_build(n, n, rp32.ctypes.data, c32.ctypes.data if col.size else None, seg_len, n_slices,
                                      None, C.byref(hp)) == 0, lib.dl_last_error()
        try:
            for name, cnt in (("seg_row", hp.n_seg), ("seg_beg", hp.n_seg), ("seg_end", hp.n_seg), ("seg_slot", hp.n_seg)):
                assert np.array_equal(_host_arr(getattr(hp, name), cnt), getattr(sliced, name).numpy()), name
        finally:
            lib.dl_host_plan_free(C.byref(hp))
    finally:
        lib.dl_host_csr_free(C.byref(hc))


@settings(max_examples=40, deadline=None)
@given(st.integers(2, 30), st.integers(0, 120), st.integers(0, 2 ** 31 - 1))
def test_pair_plans_on_random_pairs(n, P, seed):
    from disenlink_amd.graph import PairList
    rng = np.random.default_rng(seed)
    pu, pv = rng.integers(0, n, P), rng.integers(0, n, P)
    pl = PairList.build(torch.from_numpy(pu), torch.from_numpy(pv), n, seg_len=5, run_len=7, n_slices=8)
    assert pl.by_u.n_entries == P and pl.inc.n_entries == 2 * P
    _check_plan(pl.by_u, pl.by_u.rowptr.numpy(), 7)
    _check_plan(pl.inc, pl.inc.rowptr.numpy(), 5)
    ids = pl.by_u_pair.numpy()
    assert sorted(ids.tolist()) == list(range(P))
    row_of = np.repeat(np.arange(n), np.diff(pl.by_u.rowptr.numpy()))
    assert np.array_equal(pu[ids], row_of) and np.array_equal(pv[ids], pl.by_u.col.numpy())
