"""CPU-only checks: the C-ABI library loads and exports what the header declares, the host-side
graph/pair plumbing matches the oracle, and the product path refuses to run without a GPU."""
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, golden_case_names, load_golden
from oracle import dense_ref, sparse_ref


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "disenlink_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dl_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from disenlink_amd import _lib, build
    build.build()
    lib = _lib.load()
    names = _declared_symbols()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/disenlink_hip.h but not exported"
        assert n in _lib.EXPORTS, f"{n} has no ctypes signature in _lib.EXPORTS"
    assert set(_lib.EXPORTS) == set(names)
    assert b"gfx950" in lib.dl_version()


def test_argument_errors_are_reported_without_a_gpu():
    from disenlink_amd import _lib
    lib = _lib.load()
    g = _lib.DlGraph()
    F32, BF16 = _lib.DL_F32, _lib.DL_BF16
    rc = lib.dl_route_fwd(None, None, 4, 8, F32, 1.0, None, None, None, None, 0, None)
    assert rc == -1 and b"NULL" in lib.dl_last_error()
    import ctypes as C
    rc = lib.dl_route_fwd(C.byref(g), None, 0, 8, F32, 1.0, None, None, None, None, 0, None)
    assert rc == -1 and b"K=0" in lib.dl_last_error()
    rc = lib.dl_route_fwd(C.byref(g), None, 4, 8, F32, 0.0, None, None, None, None, 0, None)
    assert rc == -1 and b"temperature" in lib.dl_last_error()
    rc = lib.dl_route_fwd(C.byref(g), None, 3, 5, BF16, 1.0, None, None, None, None, 0, None)
    assert rc == -1 and b"bf16" in lib.dl_last_error()                # no generic bf16 path
    rc = lib.dl_route_fwd(C.byref(g), None, 4, 8, 7, 1.0, None, None, None, None, 0, None)
    assert rc == -1 and b"dtype" in lib.dl_last_error()
    g.csr.n_rows, g.csr.row_offset, g.csr.n_total = 5, 3, 6          # shard sticking out of the node range
    rc = lib.dl_route_fwd(C.byref(g), None, 4, 8, F32, 1.0, None, None, None, None, 0, None)
    assert rc == -1 and b"exceed n_total" in lib.dl_last_error()
    assert lib.dl_has_fast_path(8, 64) == 1 and lib.dl_has_fast_path(3, 5) == 0
    assert lib.dl_has_fast_path_dtype(16, 128, BF16) == 1 and lib.dl_has_fast_path_dtype(3, 8, BF16) == 0


def test_ops_fail_loudly_on_cpu_tensors():
    from disenlink_amd import _lib, ops
    from disenlink_amd.graph import Graph
    g = Graph.from_edge_rows(torch.tensor([0, 1]), torch.tensor([1, 2]), 3)
    Z = torch.randn(3, 2, 4)
    with pytest.raises(_lib.DisenlinkHipError, match="no CPU fallback"):
        ops.route_fwd(g, Z, 1.0)


def _check_plan(plan, rowptr, seg_len, col_slices=None, unit_segs=None):
    """segment plan (include/disenlink_hip.h): every row covered exactly once, in order, by segments of <= seg_len
    entries that never span two column slices; positions are stored slice-major, a multiple of UNIT_SEGS per stream,
    padding positions have row -1; the segments of a (row, slice) group sit, in entry order, in units of <= UNIT_SEGS
    consecutive positions that never straddle a group of UNIT_SEGS positions, full units first; rows with more than one
    unit own consecutive partial slots, one per unit, in entry order"""
    from disenlink_amd.graph import UNIT_SEGS
    unit_segs = unit_segs or UNIT_SEGS                 # 1: every segment its own unit (plans of kernels that sum nothing over a row)
    deg = np.diff(rowptr)
    col = plan.col.numpy()
    seg_row, seg_beg, seg_end = plan.seg_row.numpy(), plan.seg_beg.numpy(), plan.seg_end.numpy()
    seg_slot, sl0 = plan.seg_slot.numpy(), plan.slice_seg0.numpy()
    col_slices = col_slices or plan.n_slices          # plan.n_slices = XCD streams; col_slices = column slices
    real = seg_row >= 0
    assert (seg_beg[~real] == 0).all() and (seg_end[~real] == 0).all() and (seg_slot[~real] == -1).all()
    # column slices hold equal numbers of the plan's entries: boundaries at the quantiles of the covered columns
    covered = np.concatenate([col[b:e] for b, e in zip(seg_beg[real], seg_end[real])]) if real.any() else np.zeros(0, col.dtype)
    srt = np.sort(covered)
    bounds = srt[(np.arange(1, col_slices) * srt.size) // col_slices] if srt.size and col_slices > 1 else np.zeros(0, col.dtype)
    slice_of = lambda c: np.searchsorted(bounds, c, side="right")
    if srt.size and col_slices > 1:
        counts = np.bincount(slice_of(covered), minlength=col_slices)
        assert counts.max() - srt.size / col_slices <= np.unique(srt, return_counts=True)[1].max()   # balanced up to ties
    assert sl0[0] == 0 and sl0[-1] == plan.n_seg and plan.slice_max_seg == np.diff(sl0).max()
    assert (sl0 % UNIT_SEGS == 0).all()
    by_row = {i: [] for i in range(plan.n_rows)}
    units = []                                          # (row, [positions]) in storage order
    for x in range(plan.n_slices):
        last_q = -1
        for g0 in range(sl0[x], sl0[x + 1], UNIT_SEGS):
            prev = None
            for sgi in range(g0, g0 + UNIT_SEGS):
                if seg_row[sgi] < 0:
                    prev = None
                    continue
                b, e = seg_beg[sgi], seg_end[sgi]
                assert 0 <= e - b <= seg_len
                if e > b and col_slices > 1:
                    q = slice_of(col[b])
                    assert (slice_of(col[b:e]) == q).all()            # inside one column slice ...
                    assert q % plan.n_slices == x and q >= last_q    # ... of this stream, slices in time order
                    last_q = q
                by_row[seg_row[sgi]].append((b, e, seg_slot[sgi], sgi))
                if prev == seg_row[sgi] and unit_segs > 1:
                    units[-1][1].append(sgi)
                else:
                    units.append((seg_row[sgi], [sgi]))
                prev = seg_row[sgi]
    # a unit: consecutive positions, consecutive entries in order, one slot
    for row, poss in units:
        assert len(poss) <= unit_segs and poss == list(range(poss[0], poss[0] + len(poss)))
        assert all(seg_end[a] == seg_beg[b] for a, b in zip(poss[:-1], poss[1:]))
        assert len({seg_slot[q] for q in poss}) == 1
    multi = []
    n_units_row = {i: 0 for i in range(plan.n_rows)}
    for row, _ in units:
        n_units_row[row] += 1
    for i in range(plan.n_rows):
        segs = sorted(by_row[i])
        assert len(segs) >= 1 and segs[0][0] == rowptr[i] and segs[-1][1] == rowptr[i + 1]
        assert all(a[1] == b[0] for a, b in zip(segs[:-1], segs[1:]))          # contiguous cover
        if plan.n_slices == 1:
            assert len(segs) == max(1, -(-deg[i] // seg_len))
            assert n_units_row[i] == -(-len(segs) // unit_segs)                 # units counted from the row's first segment
        slots = [sg[2] for sg in segs]
        if n_units_row[i] > 1:
            multi.append(i)
            assert slots[0] >= 0 and all(b - a in (0, 1) for a, b in zip(slots[:-1], slots[1:]))   # entry order
            assert slots[-1] - slots[0] + 1 == n_units_row[i]
        else:
            assert set(slots) == {-1}
    assert np.array_equal(plan.multi_row.numpy(), np.array(multi, dtype=np.int32))
    slot0 = plan.multi_slot0.numpy()
    assert slot0[0] == 0 and slot0[-1] == plan.n_slots
    for m, i in enumerate(multi):
        assert sorted(by_row[i])[0][2] == slot0[m] and slot0[m + 1] - slot0[m] == n_units_row[i]


@pytest.mark.parametrize("name", golden_case_names())
def test_graph_builder_matches_oracle_csr(name):
    from disenlink_amd.graph import Graph
    g = load_golden(name)
    rowptr, col, rev = sparse_ref.csr_from_dense(g["adj"])
    for seg_len in (1, 4, 32):
        G = Graph.from_dense(torch.from_numpy(g["adj"]), seg_len=seg_len)
        assert np.array_equal(G.rowptr.numpy(), rowptr)
        assert np.array_equal(G.col.numpy(), col)
        assert np.array_equal(G.rev.numpy(), rev)
        _check_plan(G.plan, rowptr, seg_len)
        # upper plan: exactly the entries with col >= row, each once, in segments of <= seg_len
        up = G.route
        assert G.route_mirror and up.rowptr.data_ptr() == G.plan.rowptr.data_ptr() and up.col.data_ptr() == G.plan.col.data_ptr()
        src = np.repeat(np.arange(G.n_nodes), np.diff(rowptr))
        cover = np.zeros(col.size, int)
        for r_, b_, e_ in zip(up.seg_row.numpy(), up.seg_beg.numpy(), up.seg_end.numpy()):
            if r_ < 0:
                continue                                                            # padding position
            assert 0 <= e_ - b_ <= seg_len and (src[b_:e_] == r_).all()
            cover[b_:e_] += 1
        assert np.array_equal(cover, (col >= src).astype(int))
        assert set(up.seg_row.numpy().tolist()) - {-1} == set(range(G.n_nodes))    # every row owns >= 1 segment
        # a shard keeps global column ids and re-bases rowptr
        lo, hi = G.n_nodes // 3, G.n_nodes - 2
        nz = np.nonzero(g["adj"])
        S = Graph.from_edge_rows(torch.from_numpy(nz[0]), torch.from_numpy(nz[1]), G.n_nodes, symmetrise=False,
                                 seg_len=seg_len, row_range=(lo, hi))
        assert S.n_rows == hi - lo and S.row_offset == lo and S.n_nodes == G.n_nodes
        assert not S.route_mirror and S.rev is None and S.route.n_entries == S.n_edges
        assert np.array_equal(S.rowptr.numpy(), rowptr[lo:hi + 1] - rowptr[lo])
        assert np.array_equal(S.col.numpy(), col[rowptr[lo]:rowptr[hi]])
        _check_plan(S.plan, S.rowptr.numpy(), seg_len)
        from disenlink_amd.graph import CsrPlan
        sliced = CsrPlan.build(G.rowptr, G.col, G.n_nodes, seg_len=seg_len, n_slices=8)
        _check_plan(sliced, rowptr, seg_len)
        multi = CsrPlan.build(G.rowptr, G.col, G.n_nodes, seg_len=seg_len, n_slices=24)   # 3 slices per XCD stream
        assert multi.n_slices == 8
        _check_plan(multi, rowptr, seg_len, col_slices=24)


def test_graph_from_edge_rows_symmetrises_and_collapses_duplicates():
    from disenlink_amd.graph import Graph
    src = torch.tensor([0, 0, 2, 2, 3, 0])
    dst = torch.tensor([1, 1, 0, 2, 1, 1])          # duplicates, a self-loop, one-directional rows
    G = Graph.from_edge_rows(src, dst, 5)
    dense = np.zeros((5, 5), np.float32)
    dense[src.numpy(), dst.numpy()] = 1
    dense = ((dense + dense.T) != 0).astype(np.float32)
    rowptr, col, rev = sparse_ref.csr_from_dense(dense)
    assert np.array_equal(G.rowptr.numpy(), rowptr) and np.array_equal(G.col.numpy(), col)
    assert np.array_equal(G.rev.numpy(), rev)
    with pytest.raises(ValueError, match="not symmetric"):
        Graph.from_edge_rows(src, dst, 5, symmetrise=False)
    with pytest.raises(ValueError, match="outside"):
        Graph.from_edge_rows(torch.tensor([0]), torch.tensor([7]), 5)
    empty = Graph.from_edge_rows(torch.zeros(0, dtype=torch.long), torch.zeros(0, dtype=torch.long), 4)
    assert empty.n_edges == 0 and empty.n_seg == 4 and empty.rowptr.tolist() == [0] * 5     # four empty segments: one workgroup
    assert empty.plan.n_slots == 0 and empty.plan.multi_row.numel() == 0


def test_pair_incidence_lists_every_slot_once():
    from disenlink_amd.graph import PairList
    rng = np.random.default_rng(0)
    n, P = 11, 60
    pu, pv = rng.integers(0, n, P), rng.integers(0, n, P)
    pu[:3] = pv[:3]                                   # self pairs
    pl = PairList.build(torch.from_numpy(pu), torch.from_numpy(pv), n, seg_len=4, run_len=4, n_slices=8)
    ptr, other, pair = pl.inc.rowptr.numpy(), pl.inc.col.numpy(), pl.inc_pair.numpy()
    assert ptr[-1] == 2 * P
    seen = np.zeros(P, int)
    for u in range(n):
        ids = pair[ptr[u]:ptr[u + 1]]
        oth = other[ptr[u]:ptr[u + 1]]
        assert (np.diff(oth) >= 0).all()                                  # sorted by the other endpoint
        for q, o in zip(ids, oth):
            assert (pu[q] == u and pv[q] == o) or (pv[q] == u and pu[q] == o)
            seen[q] += 1
    assert (seen == 2).all()
    _check_plan(pl.inc, ptr, 4)
    # forward plan: every pair exactly once, in the row of its first endpoint
    uptr, ucol, uid = pl.by_u.rowptr.numpy(), pl.by_u.col.numpy(), pl.by_u_pair.numpy()
    assert uptr[-1] == P and sorted(uid.tolist()) == list(range(P))
    for u in range(n):
        for q, v in zip(uid[uptr[u]:uptr[u + 1]], ucol[uptr[u]:uptr[u + 1]]):
            assert pu[q] == u and pv[q] == v
    _check_plan(pl.by_u, uptr, 4, unit_segs=1)
    assert pl.by_u.n_slices == 8 and pl.inc.n_slices == 1          # the incidence plan has its own rule (graph.auto_inc_slices): tiny table, short rows -> unsliced
    pl4 = PairList.build(torch.from_numpy(pu), torch.from_numpy(pv), n, seg_len=4, run_len=4, n_slices=8, inc_slices=4)
    assert pl4.inc.n_slices == 4 and np.array_equal(pl4.inc.col.numpy(), other)       # slicing re-places segments only
    _check_plan(pl4.inc, ptr, 4)
    from disenlink_amd.graph import auto_inc_slices                # the measured shapes (profiles/r7b_train_scorer_ab.txt)
    assert [auto_inc_slices(5201, 2048, 388), auto_inc_slices(2277, 2048, 148), auto_inc_slices(41554, 2048, 331),
            auto_inc_slices(41554, 4096, 331), auto_inc_slices(2923922, 2048, 49)] == [8, 4, 16, 16, 1]
    both = PairList.build(torch.from_numpy(pu), torch.from_numpy(pv), n, seg_len=3, run_len=4, n_slices=8, inc_slices=8)
    assert both.inc.n_slices == 8
    _check_plan(both.inc, both.inc.rowptr.numpy(), 3)
    from disenlink_amd.graph import auto_slices
    assert auto_slices(5201, 2048, 200) == 8 and auto_slices(2_900_000, 2048, 24) == 1
    assert auto_slices(41608, 2048, 209) == 32 and auto_slices(41608, 2048, 10) == 1
    # a shard lists only its own nodes' slots
    sh = PairList.build(torch.from_numpy(pu), torch.from_numpy(pv), n, row_range=(3, 8))
    assert sh.inc.n_rows == 5 and sh.inc.row_offset == 3
    assert np.array_equal(sh.inc.rowptr.numpy(), ptr[3:9] - ptr[3])
    assert np.array_equal(sh.inc.col.numpy(), other[ptr[3]:ptr[8]])
    with pytest.raises(ValueError, match="by_u_range"):
        PairList.build(torch.from_numpy(pu), torch.from_numpy(pv), n, by_u_range=(0, 3))


@pytest.mark.parametrize("name", golden_case_names())
def test_module_state_dict_and_projection_match_reference(name):
    """Same state_dict keys/shapes as the reference, and the batched projection equals K separate MLPs."""
    from disenlink_amd.model import Disentangle
    g = load_golden(name)
    m = g["meta"]
    model = Disentangle(m["F"], m["nhid"], m["d"], nfactor=m["K"], beta=m["beta"], t=m["t"])
    sd = {k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd__")}
    assert list(model.state_dict().keys()) == list(sd.keys())
    model.load_state_dict(sd)
    x = torch.from_numpy(g["x"])
    Z = model.project(x)
    Zref = dense_ref.project(x, sd).permute(1, 0, 2)
    assert Z.shape == (m["N"], m["K"], m["d"]) and Z.is_contiguous()
    np.testing.assert_allclose(Z.detach().numpy(), Zref.numpy(), rtol=1e-5, atol=1e-6)
    model.train(); model.eval()                      # must stay no-ops (main_disentangled.py:193,201)


@pytest.mark.parametrize("nhid", [1, 16])
def test_stacked_buffers_follow_replaced_parameter_objects(nhid):
    """The shared [K, ...] buffers must never outlive the parameters they mirror (round-3 advisor finding): after
    load_state_dict(assign=True) or a directly replaced Parameter the fast path has to see the LIVE objects —
    snapshot_state() == state_dict(), project() computes with the new weights."""
    from disenlink_amd.model import Disentangle
    torch.manual_seed(3)
    K, F, d = 3, 5, 4
    model = Disentangle(F, nhid, d, nfactor=K, beta=0.5)
    x = torch.randn(7, F)
    donor = Disentangle(F, nhid, d, nfactor=K, beta=0.5)
    sd = {k: v.clone() for k, v in donor.state_dict().items()}
    assert model._stacked_params() is not None
    model.load_state_dict(sd, assign=True)
    flat = model._stacked_params()
    assert flat is not None, "assign=True must leave the module on its shared buffers again"
    live = dict(model.named_parameters())
    assert all(any(p is q for q in live.values()) for p in flat) and len(flat) == len(live)
    snap = model.snapshot_state()
    for k, v in model.state_dict().items():
        assert torch.equal(snap[k], v) and torch.equal(v, sd[k])
    np.testing.assert_allclose(model.project(x).detach().numpy(), donor.project(x).detach().numpy(), rtol=1e-6)
    # a Parameter replaced behind the module's back: the fast path must notice (no stale list), results follow the new one
    lin = model.factor_1.mlp if nhid == 1 else model.factor_1.mlp2
    lin.weight = torch.nn.Parameter(torch.full_like(lin.weight, 0.25))
    assert model._stacked_params() is None
    snap = model.snapshot_state()
    key = "factor_1.mlp.weight" if nhid == 1 else "factor_1.mlp2.weight"
    assert torch.equal(snap[key], torch.full_like(lin.weight, 0.25))
    Z = model.project(x)
    ref = torch.stack([f(x) for f in model.factors], dim=1)
    np.testing.assert_allclose(Z.detach().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-6)
    model.restack()
    assert model._stacked_params() is not None
    np.testing.assert_allclose(model.project(x).detach().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-6)


def _host_arr(ptr, n):
    return np.ctypeslib.as_array(ptr, shape=(max(n, 1),))[:n].copy()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_c_abi_host_graph_prep_matches_python_builders(seed):
    """dl_host_csr_from_edges / dl_host_plan_build (C++, for non-Python hosts) produce exactly the arrays
    disenlink_amd.graph builds with torch ops: CSR, reverse permutation, plain / sliced / mirrored plans."""
    import ctypes as C
    from disenlink_amd import _lib
    from disenlink_amd.graph import CsrPlan, Graph
    lib = _lib.load()
    rng = np.random.default_rng(seed)
    n = int(rng.integers(20, 120))
    E = int(rng.integers(n, 12 * n))
    src, dst = rng.integers(0, n, E), rng.integers(0, n, E)
    src[: n // 2] = 0                                                     # a hub
    keepn = (src != n - 1) & (dst != n - 1)                               # an isolated node
    src, dst = src[keepn].astype(np.int64), dst[keepn].astype(np.int64)
    G = Graph.from_edge_rows(torch.from_numpy(src), torch.from_numpy(dst), n, seg_len=8)
    hc = _lib.DlHostCsr()
    assert lib.dl_host_csr_from_edges(src.ctypes.data, dst.ctypes.data, src.size, n, 1, C.byref(hc)) == 0
    try:
        assert hc.n_nodes == n and hc.n_entries == G.n_edges
        rowptr, col = _host_arr(hc.rowptr, n + 1), _host_arr(hc.col, hc.n_entries)
        assert np.array_equal(rowptr, G.rowptr.numpy()) and np.array_equal(col, G.col.numpy())
        assert np.array_equal(_host_arr(hc.rev, hc.n_entries), G.rev.numpy())
        row_of = np.repeat(np.arange(n), np.diff(rowptr))
        # (a graph this small has cache-resident tables: Graph.from_edge_rows orders its plans by length)
        cases = [(8, 1, None, 4, 1, G.plan), (8, 8, None, 4, 0, CsrPlan.build(G.rowptr, G.col, n, seg_len=8, n_slices=8)),
                 (8, 1, None, 4, 0, CsrPlan.build(G.rowptr, G.col, n, seg_len=8)),
                 (5, 24, None, 4, 0, CsrPlan.build(G.rowptr, G.col, n, seg_len=5, n_slices=24)),
                 (5, 24, None, 4, 1, CsrPlan.build(G.rowptr, G.col, n, seg_len=5, n_slices=24, by_length=True)),
                 (3, 8, None, 1, 0, CsrPlan.build(G.rowptr, G.col, n, seg_len=3, n_slices=8, unit_segs=1)),
                 (3, 8, None, 1, 1, CsrPlan.build(G.rowptr, G.col, n, seg_len=3, n_slices=8, unit_segs=1, by_length=True)),
                 (8, 1, (col >= row_of).astype(np.uint8), 1, 1, G.route)]
        for seg_len, slices, keep, unit_segs, by_length, ref in cases:
            hp = _lib.DlHostPlan()
            kp = keep.ctypes.data if keep is not None else None
            rc = lib.dl_host_plan_build(n, n, rowptr.ctypes.data, col.ctypes.data, seg_len, slices, kp, unit_segs, by_length,
                                        C.byref(hp))
            assert rc == 0, lib.dl_last_error()
            try:
                got = (hp.n_seg, hp.n_slices, hp.slice_max_seg, hp.n_multi, hp.n_slots)
                assert got == (ref.n_seg, ref.n_slices, ref.slice_max_seg, int(ref.multi_row.numel()), ref.n_slots)
                for name, count in (("seg_row", hp.n_seg), ("seg_beg", hp.n_seg), ("seg_end", hp.n_seg),
                                    ("seg_slot", hp.n_seg), ("slice_seg0", hp.n_slices + 1), ("multi_row", hp.n_multi),
                                    ("multi_slot0", hp.n_multi + 1), ("slot_multi", hp.n_slots)):
                    assert np.array_equal(_host_arr(getattr(hp, name), count), getattr(ref, name).numpy()), name
            finally:
                lib.dl_host_plan_free(C.byref(hp))
    finally:
        lib.dl_host_csr_free(C.byref(hc))
    # asymmetric input without symmetrise is rejected, like Graph.from_edge_rows
    one = np.array([0], dtype=np.int64), np.array([1], dtype=np.int64)
    assert lib.dl_host_csr_from_edges(one[0].ctypes.data, one[1].ctypes.data, 1, 3, 0, C.byref(_lib.DlHostCsr())) == -1
    assert b"not symmetric" in lib.dl_last_error()


def test_module_copies_and_pickles_without_its_caches():
    import copy
    import io
    from disenlink_amd.model import Disentangle
    m = Disentangle(6, 5, 8, nfactor=3, beta=0.5)
    m._graph_cache = (lambda: None, 0, None)                     # stands in for a live cache entry
    c = copy.deepcopy(m)
    assert c._graph_cache is None
    for (k1, v1), (k2, v2) in zip(m.state_dict().items(), c.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    buf = io.BytesIO()
    torch.save(m, buf)
    x = torch.randn(4, 6)
    assert torch.equal(c.project(x), m.project(x))               # the copy re-stacks its own parameters


def test_bench_pmc_table_covers_every_phase_kernel():
    """bench.py prices each timed phase with the PMC traffic of its kernels (profiles/pmc_traffic_latest.json: one entry
    per workload, kernels keyed by the names rocprofv3 reports, tagged with the hash of the kernel sources it was
    collected for).  Every entry must cover every phase; a summary collected for OTHER kernel sources must come back
    as None with the reason, never as numbers."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    table = json.load(open(bench.PMC_SUMMARY))
    assert table, "no PMC passes committed"
    for key, entry in table.items():
        assert len(entry["kernel_source_hash"]) == 16 and "rocprofv3" in entry["source"], key
        bases = {k.split("<")[0].split("::")[-1]: v for k, v in entry["kernels"].items()}
        # `<workload>_train` entries (round 4) are passes over `--sections fwd_bwd`: the training step's kernels
        phases = bench.TRAIN_PHASE_KERNELS if key.endswith("_train") else bench.PHASE_KERNELS
        for phase, names in phases.items():
            assert any(n in bases and bases[n]["traffic_bytes"] > 0 for n in names), (key, phase)
        got, why = bench.pmc_traffic(key, phases=phases)
        if entry["kernel_source_hash"] == bench.kernel_source_hash():
            assert set(got) == set(phases) and all(v > 0 for v in got.values())
        else:
            assert got is None and entry["kernel_source_hash"] in why
    assert bench.pmc_traffic("no_such_workload")[0] is None
    # the L2-side request passes (round 4): what the kernels ask of the L2s, next to what leaves them
    l2 = json.load(open(bench.PMC_L2_SUMMARY))
    assert l2, "no L2-request passes committed"
    for key, entry in l2.items():
        assert len(entry["kernel_source_hash"]) == 16 and "TCC_READ_sum" in entry["source"], key
        phases = bench.TRAIN_PHASE_KERNELS if key.endswith("_train") else bench.PHASE_KERNELS
        got = bench.pmc_l2(key, phases=phases)
        if entry["kernel_source_hash"] == bench.kernel_source_hash():
            assert set(got) == set(phases) and all(v["l2_bytes"] > 0 and 0 <= v["hit_rate"] <= 1 for v in got.values()), key
        else:
            assert got is None


def test_bench_prices_against_the_measured_roof():
    """bench.memory_bound: with PMC passes the roof follows the measured traffic (bytes leaving the L2s below half of the
    requested bytes: cache-served, else HBM), whatever the table size; without them the table size decides and says so."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    mb = dict(route=400e6, aggregate=130e6, score=4.2e9)
    b, peak, how = bench.memory_bound(21 << 20, mb, dict(route=100e6, aggregate=80e6, score=210e6))
    assert (b, peak) == ("l2", bench.L2_PEAK_GBS) and "measured" in how
    b, peak, how = bench.memory_bound(340 << 20, mb, dict(route=100e6, aggregate=80e6, score=210e6))
    assert b == "l2" and "measured" in how                          # a 340 MB table served by the caches is NOT HBM-bound
    b, peak, how = bench.memory_bound(21 << 20, mb, dict(route=390e6, aggregate=150e6, score=4.0e9))
    assert (b, peak) == ("hbm", bench.HBM_PEAK_GBS)
    b, _p, how = bench.memory_bound(340 << 20, mb, None)
    assert b == "hbm" and "no PMC" in how
    assert bench.memory_bound(21 << 20, mb, None)[0] == "l2"


def test_torch_library_operators_are_registered_with_schemas_and_shape_functions():
    """disenlink_amd/torch_ops.py: the C ABI as torch.ops.disenlink.* (schema, fake implementation, autograd).  Without a
    GPU: the operators exist, carry the documented schemas, and their fake implementations propagate shapes / dtypes for
    fake CUDA tensors — what torch.compile needs to trace through them."""
    from torch._subclasses.fake_tensor import FakeTensorMode
    import disenlink_amd.torch_ops as to
    from disenlink_amd.graph import Graph, PairList
    for name in ("route_fwd", "aggregate_fwd", "route_aggregate", "route_aggregate_bwd", "score_pairs_terms",
                 "score_pairs_bwd", "project_fwd", "project_bwd"):
        assert hasattr(torch.ops.disenlink, name), name
    assert str(torch.ops.disenlink.route_aggregate.default._schema) == \
        "disenlink::route_aggregate(Tensor Z, SymInt graph, float beta, float t) -> (Tensor, Tensor, Tensor, Tensor)"
    g = Graph.from_edge_rows(torch.tensor([0, 1, 2, 3]), torch.tensor([1, 2, 3, 0]), 5)
    pl = PairList.build(torch.tensor([0, 1, 2]), torch.tensor([3, 4, 0]), 5)
    hg, hp = to.register_graph(g), to.register_pairs(pl)
    assert to.register_graph(g) == hg and to.register_pairs(pl) == hp          # idempotent per object
    with FakeTensorMode():
        Z = torch.empty(5, 4, 32, device="cuda")
        H, p, a, s = torch.ops.disenlink.route_aggregate(Z, hg, 0.5, 1.0)
        prob, coef = torch.ops.disenlink.score_pairs_terms(Z, H, hp, 1.0)
        dZ, dH = torch.ops.disenlink.score_pairs_bwd(Z, H, prob, prob, coef, hp, 1.0)
        assert H.shape == Z.shape and p.shape == (8,) and p.dtype == torch.uint8 and s.shape == (5, 4)
        assert prob.shape == (3,) and coef.shape == (2, 3, 4) and dZ.shape == Z.shape and dH.dtype == torch.float32
        x = torch.empty(5, 7, device="cuda")
        Zp = torch.ops.disenlink.project_fwd(x, torch.empty(4, 16, 7, device="cuda"), torch.empty(4, 16, device="cuda"),
                                             torch.empty(4, 32, 16, device="cuda"), torch.empty(4, 32, device="cuda"))
        assert Zp.shape == (5, 4, 32)
    with pytest.raises(ValueError, match="no graph registered"):
        to._g(10 ** 9)
    to.release(hg)
    to.release(hp)
    # the registry holds its objects weakly (round-3 advisor finding: a caller that builds a new pair list per epoch
    # must not accumulate GPU plans): an unpinned handle vanishes with its object, a pinned one lives until release()
    import gc
    g2 = Graph.from_edge_rows(torch.tensor([0, 1]), torch.tensor([1, 2]), 3)
    h_weak, h_pin = to.register_graph(g2), to.register_pairs(pl, pin=True)
    assert to._g(h_weak) is g2
    del g2, pl
    gc.collect()
    with pytest.raises(ValueError, match="no graph registered"):
        to._g(h_weak)
    assert to._p(h_pin).n_pairs == 3
    to.release(h_pin)
    gc.collect()
    with pytest.raises(ValueError, match="no pair list registered"):
        to._p(h_pin)


def test_feature_rows_do_not_depend_on_who_generates_them():
    """SyntheticGraph.features(rows=...): every 16,384-row block has its own counter-based stream, so a rank of a
    sharded run that generates only its own rows gets exactly the rows of the whole matrix."""
    from disenlink_amd.data import synthetic_graph
    sg = synthetic_graph("snap_patents", seed=0, scale=0.02)          # 58k nodes: several blocks
    assert sg.n_nodes > 3 * sg.FEATURE_BLOCK
    full = sg.features()
    assert full.shape == (sg.n_nodes, sg.n_feat) and full.dtype == np.float32
    np.testing.assert_allclose(full.mean(axis=1), 0.0, atol=1e-5)
    np.testing.assert_allclose(full.std(axis=1, ddof=1), 1.0, rtol=1e-4)
    for r0, r1 in ((0, 1), (16383, 16385), (20000, 51234), (sg.n_nodes - 5, sg.n_nodes), (777, 777)):
        assert np.array_equal(sg.features(rows=(r0, r1)), full[r0:r1])
    with pytest.raises(ValueError):
        sg.features(rows=(5, sg.n_nodes + 1))


def test_split_on_a_torch_device_is_the_numpy_split():
    """make_link_split(device=...): the sorts / searches run through torch (on the GPU in the sharded bench), the random
    draws stay in numpy — the pair sets must come out identical, element for element."""
    from disenlink_amd.data import synthetic_graph
    from disenlink_amd.splits import make_link_split
    sg = synthetic_graph("chameleon", seed=3)
    a = make_link_split(sg.src, sg.dst, sg.n_nodes, m=5, seed=3)
    b = make_link_split(sg.src, sg.dst, sg.n_nodes, m=5, seed=3, device=torch.device("cpu"))
    assert np.array_equal(a.train_src, b.train_src) and np.array_equal(a.train_dst, b.train_dst)
    for name in ("pos_train", "neg_train", "val", "test"):
        x, y = getattr(a, name), getattr(b, name)
        assert np.array_equal(x.u, y.u) and np.array_equal(x.v, y.v) and np.array_equal(x.label, y.label), name


def test_bench_launches_its_own_ranks_when_started_plainly():
    """`python bench.py --gpus N` (N > 1) without a launcher — the way the driver starts it — must start its N ranks
    itself as child processes BEFORE any GPU call, relay rank 0's JSON line and exit 0; a failing rank must fail the
    whole launch.  DL_BENCH_LAUNCH_CHECK replaces the measurement by one gloo all-reduce, so this runs without a GPU (the
    parent never touches one: this container has none, and it gets as far as a successful rendezvous)."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["DL_BENCH_LAUNCH_CHECK"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                       # ONE JSON line on stdout, everything else on stderr
    line = json.loads(lines[0])
    assert line["launch_check"] and line["n_gpus"] == 2 and line["rank_sum"] == 3.0 and line["self_launched"]
    env["DL_BENCH_LAUNCH_CHECK"] = "fail"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and not r.stdout.strip()
    # under a launcher the rank count must agree with --gpus
    env2 = dict(env, RANK="0", WORLD_SIZE="1", DL_BENCH_LAUNCH_CHECK="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env2, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode != 0 and "disagree" in r.stderr


def test_bench_warmup_settles_and_parity_block_reports_both_sides():
    """bench.steady_warmup: blocks until >= min_s have passed AND three consecutive blocks agree within tol (a ramp keeps
    it going; max_s bounds it).  bench.parity_block: max |dprob| and the tie-aware AUC of both sides (the GPU side's AUC
    falls back to the torch form on CPU tensors), with the tolerances the metric names."""
    import time
    import bench
    ramp = iter([1.0, 0.8, 0.7, 0.65, 0.64, 0.64, 0.641, 0.64, 0.64])
    calls = []

    def block():
        calls.append(1)
        time.sleep(0.002)
        return next(ramp)
    n, el, settled = bench.steady_warmup(block, min_s=1e-6, tol=0.02, max_s=5.0)
    assert settled and n == 6, (n, settled)                     # 0.65, 0.64, 0.64 is the first triple within 2 %
    n, el, settled = bench.steady_warmup(lambda: (time.sleep(0.01), 1.0)[1], min_s=0.05, tol=0.02, max_s=5.0)
    assert settled and el >= 0.05 and n >= 5
    k = [0]
    def never():
        k[0] += 1
        time.sleep(0.005)
        return float(k[0])
    n, el, settled = bench.steady_warmup(never, min_s=1e-6, tol=0.02, max_s=0.05)
    assert bench.steady_warmup(never, min_s=0.0)[0] == 1             # --warm-s 0 (profiler passes): one block, no waiting
    assert not settled and el >= 0.05
    rng = np.random.default_rng(0)
    label = (rng.random(500) < 0.3).astype(np.float32)
    prob = rng.random(500).astype(np.float32)
    par = bench.parity_block(prob.copy(), label, torch.from_numpy(prob))
    assert par["ok"] and par["max_abs_dprob"] == 0.0 and abs(par["auc_gpu"] - par["auc_cpu"]) < 1e-12 and par["pairs"] == 500
    off = prob.copy()
    off[7] += 1e-3
    par = bench.parity_block(off, label, torch.from_numpy(prob))
    assert not par["ok"] and abs(par["max_abs_dprob"] - 1e-3) < 1e-6


def test_compiled_torch_binding_builds_loads_and_registers_its_operator():
    """libdisenlink_torch.so (csrc/torch/dl_torch.cpp, built by disenlink_amd.build with g++ against the installed torch):
    the TORCH_LIBRARY binding over the C ABI loads next to libdisenlink_hip.so and registers
    torch.ops.disenlink_native.hot_path_pairs_loss with the documented schema; it reaches the same library (abi_version is
    dl_version()).  No compute without a GPU."""
    from disenlink_amd import _lib, build, native
    path = build.build_torch_binding()
    assert os.path.exists(path) and path.endswith("libdisenlink_torch.so")
    assert native.available()
    op = torch.ops.disenlink_native.hot_path_pairs_loss
    assert str(op.default._schema) == (
        "disenlink_native::hot_path_pairs_loss(Tensor Z, int graph_ptr, int inc_ptr, int n_edges, float beta, float t, "
        "Tensor label, Tensor weight, Tensor ws_graph, Tensor ws_pairs, Tensor ws_bce, int table_bf16) -> (Tensor, Tensor, Tensor)")
    # round 5: the rest of the training step — projection node, Adam step, AUC counts
    schemas = {"project_stacked": "disenlink_native::project_stacked(Tensor x, Tensor W1, Tensor b1, Tensor W2, Tensor b2, "
                                  "Tensor[] params, bool keep_hid, Tensor? xplanes) -> Tensor",
               "adam_step": "disenlink_native::adam_step(Tensor[] bufs, Tensor[] params, Tensor[] exp_avg, Tensor[] exp_avg_sq, "
                            "Tensor state, float lr, float beta1, float beta2, float eps, float weight_decay, int host_step=0) -> ()",
               "auc_pair_counts": "disenlink_native::auc_pair_counts(Tensor score, Tensor pos_idx, Tensor neg_idx) -> Tensor",
               "epoch_finish": "disenlink_native::epoch_finish(Tensor score_val, Tensor pos_idx, Tensor neg_idx, Tensor u2, Tensor loss, "
                               "Tensor[] params, Tensor[] best, Tensor state, Tensor hist, int ring_ptr, int ring, float denom2, "
                               "int max_epochs, int patience) -> ()"}
    for name, want in schemas.items():
        assert str(getattr(torch.ops.disenlink_native, name).default._schema) == want, name
    assert torch.ops.disenlink_native.abi_version() == _lib.load().dl_version().decode()
    # the operator set / schema number the Python side expects: a library built for another one is refused (available() False
    # -> the ctypes binding carries on), instead of failing inside a training step
    assert int(torch.ops.disenlink_native.binding_abi()) == native.BINDING_ABI
    with pytest.raises(RuntimeError, match="CUDA fp32"):
        op(torch.zeros(3, 2, 8), 0, 0, 0, 0.5, 1.0, torch.zeros(1), torch.zeros(1), torch.zeros(1), torch.zeros(1), torch.zeros(1), 0)


def test_link_pred_behaves_like_a_tensor_for_copies_and_files():
    """ops.LinkPred (the dense link_pred of the drop-in forward) only adds a hook to indexing: every result is a plain
    tensor, deepcopy / pickle / torch.save of it are plain tensors too (no module cache inside, weights_only loads work),
    and autograd runs through a mask gather exactly as through a plain tensor."""
    import copy
    import io
    import pickle
    from disenlink_amd import ops

    class Cache:
        notes = 0

        def note_index(self, n, idx):
            Cache.notes += 1

    base = torch.rand(6, 6, requires_grad=True)
    t = ops.as_link_pred(base * 1.0, Cache())
    assert isinstance(t, ops.LinkPred) and t.requires_grad
    m = torch.rand(6, 6, generator=torch.Generator().manual_seed(1)) > 0.5
    for r in (t[m], t[2, 3], t[1:3], t + 1, torch.sigmoid(t), t.detach(), t.clone(), t.view(-1), t.t(), torch.masked_select(t, m),
              t[torch.tensor([0, 1]), torch.tensor([2, 3])]):
        assert type(r) is torch.Tensor
    assert Cache.notes >= 4                                          # every a_pred[...] was reported
    d = ops.as_link_pred(torch.rand(3, 3), Cache())
    c = copy.deepcopy(d)
    assert type(c) is torch.Tensor and torch.equal(c, d) and c.data_ptr() != d.data_ptr()
    assert type(pickle.loads(pickle.dumps(d))) is torch.Tensor
    buf = io.BytesIO()
    torch.save(d, buf)
    buf.seek(0)
    back = torch.load(buf, weights_only=True)
    assert type(back) is torch.Tensor and torch.equal(back, d)
    torch.nn.functional.binary_cross_entropy(t[m], torch.ones(int(m.sum()))).backward()
    ref = torch.rand(0)                                              # the same through a plain tensor
    b2 = base.detach().clone().requires_grad_(True)
    torch.nn.functional.binary_cross_entropy((b2 * 1.0)[m], torch.ones(int(m.sum()))).backward()
    assert torch.equal(base.grad, b2.grad)


def test_graph_from_dense_takes_sparse_layouts_and_refuses_weights():
    """Graph.from_dense: adj is the 0 / 1 matrix the reference's layer multiplies by (model.py:62) — dense of any dtype,
    sparse COO or CSR give the same plan; an entry that is neither 0 nor 1 raises instead of becoming an edge."""
    from disenlink_amd.graph import Graph
    g = torch.Generator().manual_seed(3)
    a = (torch.rand(40, 40, generator=g) < 0.1).float()
    a = ((a + a.t()) != 0).float()
    ref = Graph.from_dense(a)
    for other in (a.bool(), a.long(), a.double(), a.to_sparse(), a.to_sparse_csr()):
        got = Graph.from_dense(other)
        assert torch.equal(got.rowptr, ref.rowptr) and torch.equal(got.col, ref.col)
    sp = a.to_sparse()
    explicit_zero = torch.sparse_coo_tensor(torch.cat([sp.indices(), torch.tensor([[0], [0]])], 1),
                                            torch.cat([sp.values(), torch.zeros(1)]), a.shape)
    if a[0, 0] == 0:
        assert torch.equal(Graph.from_dense(explicit_zero).col, ref.col)          # a stored 0 is no edge
    w = a.clone()
    r, c = torch.nonzero(a)[0].tolist()
    w[r, c] = w[c, r] = 0.5
    for bad in (w, w.to_sparse()):
        with pytest.raises(ValueError, match="0 / 1"):
            Graph.from_dense(bad)


def test_label_stream_is_keyed_on_tensor_objects():
    """PairList.bind_labels: the per-entry (label, weight) stream follows the tensor OBJECTS and their version counters.  Two
    tensors over the same memory (torch.from_numpy twice: same address, version 0 both) whose contents changed in between
    are two different label vectors — the second must not get the first one's stream."""
    from disenlink_amd.graph import PairList
    rng = np.random.default_rng(1)
    N, P = 50, 400
    pu, pv = torch.from_numpy(rng.integers(0, N, P)), torch.from_numpy(rng.integers(0, N, P))
    pl = PairList.build(pu, pv, N, build_by_u=False)
    mem = rng.random(P).astype(np.float32)
    weight = torch.full((P,), 0.5)
    a = torch.from_numpy(mem)
    pl.bind_labels(a, weight, P)
    assert pl._yw is None                                            # first sight
    pl.bind_labels(a, weight, P)
    assert pl._yw is not None                                        # second sight of the same objects: bound
    first = pl._yw.clone()
    mem[:] = 1.0 - mem                                               # other contents, no version bump (written through numpy)
    b = torch.from_numpy(mem)
    assert b.data_ptr() == a.data_ptr() and b._version == a._version and b is not a
    pl.bind_labels(b, weight, P)
    assert pl._yw is None                                            # a NEW object: first sight again, nothing stale is used
    pl.bind_labels(b, weight, P)
    assert pl._yw is not None and not torch.equal(pl._yw[:, 0], first[:, 0])
    q = pl.inc_pair.long()
    assert torch.equal(pl._yw[:, 0], b[q]) and torch.equal(pl._yw[:, 1].abs(), weight[q])
    a.add_(0)                                                        # an in-place write bumps the version: rebinding `a` starts over
    pl.bind_labels(a, weight, P)
    assert pl._yw is None
