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
                                      None, 4, 0, C.byref(hp)) == 0, lib.dl_last_error()
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
    _check_plan(pl.by_u, pl.by_u.rowptr.numpy(), 7, unit_segs=1)
    _check_plan(pl.inc, pl.inc.rowptr.numpy(), 5)
    ids = pl.by_u_pair.numpy()
    assert sorted(ids.tolist()) == list(range(P))
    row_of = np.repeat(np.arange(n), np.diff(pl.by_u.rowptr.numpy()))
    assert np.array_equal(pu[ids], row_of) and np.array_equal(pv[ids], pl.by_u.col.numpy())


@st.composite
def tiny_problem(draw):
    n = draw(st.integers(2, 14))
    k = draw(st.sampled_from([1, 2, 3, 5]))
    d = draw(st.sampled_from([1, 2, 4]))
    seed = draw(st.integers(0, 2 ** 31 - 1))
    dens = draw(st.sampled_from([0.0, 0.15, 0.5, 1.0]))
    return n, k, d, seed, dens, draw(st.sampled_from([0.5, 0.7, 0.9])), draw(st.sampled_from([1.0, 2.0]))


@settings(max_examples=40, deadline=None)
@given(tiny_problem())
def test_sparse_oracle_equals_autograd_of_the_dense_oracle(prob):
    """The CSR + pair-list restatement (forward and the analytic backward of SURVEY.md Appendix A.3) against
    torch autograd of the dense restatement, on random tiny graphs: empty graphs, full graphs, self-loops, K = 1."""
    from oracle import dense_ref, sparse_ref
    n, k, d, seed, dens, beta, t = prob
    rng = np.random.default_rng(seed)
    adj = (rng.random((n, n)) < dens).astype(np.float32)
    adj = ((adj + adj.T) > 0).astype(np.float32)                   # symmetric, diagonal may be 1
    Z = (rng.standard_normal((n, k, d)) * 0.7).astype(np.float32)
    Zt = torch.from_numpy(Z).permute(1, 0, 2).contiguous().requires_grad_(True)      # dense oracle wants [K,N,d]
    H_d, e_d, _att, _p, _s = dense_ref.route_aggregate(Zt, torch.from_numpy(adj), beta, t)
    P_d = dense_ref.score_allpairs(H_d, e_d)
    w = torch.from_numpy(rng.standard_normal((n, n)).astype(np.float32))
    (P_d * w).sum().backward()
    rowptr, col, rev = sparse_ref.csr_from_dense(adj)
    alpha = (e_d / e_d.sum(0)).detach().numpy()
    if k > 1:                                                      # skip problems with a routing near-tie on an edge
        top = np.sort(alpha, axis=0)
        if ((top[-1] - top[-2])[adj > 0] < 1e-4).any():
            return
    H, p, a, s_raw = sparse_ref.forward(Z, rowptr, col, beta, t)
    np.testing.assert_allclose(H, H_d.detach().permute(1, 0, 2).numpy(), rtol=2e-5, atol=2e-6)
    uu, vv = np.divmod(np.arange(n * n), n)
    P = sparse_ref.score_pairs(Z, H, uu, vv, t)
    np.testing.assert_allclose(P.reshape(n, n), P_d.detach().numpy(), rtol=2e-5, atol=2e-6)
    dZ_s, dH = sparse_ref.score_pairs_bwd(Z, H, uu, vv, t, w.numpy().reshape(-1))
    dZ = dZ_s + sparse_ref.route_aggregate_bwd(Z, rowptr, col, rev, p, a, s_raw, beta, t, dH)
    want = Zt.grad.permute(1, 0, 2).numpy()
    assert np.abs(dZ - want).max() <= 5e-4 * max(np.abs(want).max(), 1e-6)
