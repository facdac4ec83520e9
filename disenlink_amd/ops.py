"""Operators of the DisenLink hot path over libdisenlink_hip.so.

Raw wrappers (``route_fwd`` ...) take/return device tensors and launch on torch's current
stream; the ``autograd.Function``s stitch them into the graph so that the reference's
training loop (``loss.backward()``, main_disentangled.py:198) works unchanged.

There is no fallback: tensors must be CUDA(HIP) fp32 tensors and the library must be built.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref

import torch

from . import _lib
from .graph import Graph, PairList


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    """Handle of torch's current stream on the current device (the raw getter where torch has it: a fifth of the host
    time of building a Stream object, ~25 times per training epoch)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _need_cuda(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if not t.is_cuda:
            raise _lib.DisenlinkHipError(
                "disenlink_amd operators run only on the GPU through libdisenlink_hip.so "
                "(there is no CPU fallback); got a tensor on " + str(t.device))


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise TypeError(f"expected float32, got {t.dtype}")
    return t.contiguous()


def _tab(t: torch.Tensor):
    """A node table (Z or H): fp32 or bf16 storage -> (contiguous tensor, dl_dtype code)."""
    if t.dtype == torch.float32:
        return t.contiguous(), _lib.DL_F32
    if t.dtype == torch.bfloat16:
        return t.contiguous(), _lib.DL_BF16
    raise TypeError(f"node tables must be float32 or bfloat16, got {t.dtype}")


def _nkd(Z: torch.Tensor):
    if Z.dim() != 3:
        raise ValueError("Z/H must be [N, K, d]")
    return Z.shape[0], Z.shape[1], Z.shape[2]


# DL_POISON=1 (tests): every buffer the kernels are expected to fill — outputs and the workspace — starts as
# NaN bit patterns instead of whatever the allocator hands out, so a read of anything nobody wrote shows up as
# a NaN in the result instead of depending on what happened to be in that memory.
_POISON = os.environ.get("DL_POISON", "0") == "1"


def _empty(shape, dtype, device):
    t = torch.empty(shape, dtype=dtype, device=device)
    if _POISON:
        t.view(torch.uint8).fill_(0xFF) if t.numel() else None
    return t


def _empty_like(x):
    return _empty(x.shape, x.dtype, x.device)


class _Workspace:
    """Grow-only scratch per device; the C ABI never allocates.  ONE buffer per device, whatever the stream: launches on one
    stream use it one after the other — two streams of one process computing side by side would share it (INTEGRATION.md:
    one compute stream per device, or order the streams).  (Per-stream buffers were tried at the end of round 6 and taken
    back: every graph-replayed run warms up on a fresh side stream and captures on another, so the buffers — and the
    captured graphs' memory pools they were allocated from — accumulated run after run.)"""

    def __init__(self):
        self.buf: dict = {}

    def get(self, nbytes: int, device) -> torch.Tensor:
        cur = self.buf.get(device)
        if cur is None or cur.numel() < nbytes:
            cur = torch.empty(max(nbytes, 1024), dtype=torch.uint8, device=device)
            self.buf[device] = cur
        if _POISON:
            cur.fill_(0xFF)
        return cur


_ws = _Workspace()
_bce_ws: dict = {}


def _ws_bce(device) -> torch.Tensor:
    """dl_pair_bce's 8 KiB of scratch: a buffer of its own (the shared grow-only workspace is handed to the plans' kernels
    in the same step; the compiled binding passes all three scratch tensors into one call)."""
    t = _bce_ws.get(device)
    if t is None:
        t = _bce_ws[device] = torch.empty(8192, dtype=torch.uint8, device=device)
    return t


_ws_need: dict = {}


def _workspace(plan_ref, device, K: int, d: int):
    """Scratch for the plan behind `plan_ref` (a ctypes byref into the owner's cached struct: its address identifies the
    plan for as long as the owner lives; the size is asked once per (plan, K, d))."""
    pl = plan_ref._obj                                            # the sizes ride along: an address can be reused
    key = (C.addressof(pl), K, d, pl.n_entries, pl.n_total, pl.n_slots)
    need = _ws_need.get(key)
    if need is None:
        need = _ws_need[key] = int(_lib.load().dl_workspace_bytes(plan_ref, K, d))
        if len(_ws_need) > 4096:                                  # plans come and go (tests, sweeps): do not grow forever
            _ws_need.clear()
    return _ws.get(need, device)


def _check_rows(g: Graph, Z: torch.Tensor):
    N, K, d = _nkd(Z)
    if N != g.n_nodes:
        raise ValueError(f"Z has {N} rows, graph has {g.n_nodes} nodes")
    return N, K, d


# ---------------------------------------------------------------------- raw wrappers
def route_fwd(g: Graph, Z: torch.Tensor, t: float, s_out: torch.Tensor | None = None, p_out: torch.Tensor | None = None,
              a_out: torch.Tensor | None = None):
    """-> p uint8[E], a f32[E], s f32[N,K] (raw sums; only the graph's rows are written).
    model.py:56-72 on the edges of adj.  p_out / a_out: write into existing per-entry arrays (a graph whose routing plan
    covers only SOME entries — dist.Shard.route_by_peer — fills its part of them; the row sums are taken over the
    arrays as they stand, so they are final after the last part)."""
    lib = _lib.load()
    Z, dt = _tab(Z)
    _need_cuda(Z, g.rowptr)
    N, K, d = _check_rows(g, Z)
    if (p_out is None) != (a_out is None):
        raise ValueError("p_out and a_out come together")
    if p_out is not None and (p_out.dtype != torch.uint8 or a_out.dtype != torch.float32 or p_out.numel() != g.n_edges
                              or a_out.numel() != g.n_edges or not p_out.is_contiguous() or not a_out.is_contiguous()):
        raise ValueError("p_out / a_out must be contiguous uint8 / float32 arrays of n_edges entries")
    p = _empty(g.n_edges, torch.uint8, Z.device) if p_out is None else p_out
    a = _empty(g.n_edges, torch.float32, Z.device) if a_out is None else a_out
    s = _empty((N, K), torch.float32, Z.device) if s_out is None else s_out
    ws = _workspace(g.c_plan(), Z.device, K, d)
    _lib.check(lib.dl_route_fwd(g.c_struct(), Z.data_ptr(), K, d, dt, float(t), p.data_ptr(), a.data_ptr(),
                                s.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "dl_route_fwd")
    return p, a, s


def aggregate_fwd(g: Graph, Z: torch.Tensor, beta: float, p, a, s, H_out: torch.Tensor | None = None):
    """-> H f32[N,K,d] (only the graph's rows are written).  model.py:73-75."""
    lib = _lib.load()
    Z, dt = _tab(Z)
    _need_cuda(Z, g.rowptr, p, a, s)
    N, K, d = _check_rows(g, Z)
    H = _empty_like(Z) if H_out is None else H_out
    if H.dtype != Z.dtype:
        raise TypeError("H must have the storage type of Z")
    ws = _workspace(g.c_plan(), Z.device, K, d)
    _lib.check(lib.dl_aggregate_fwd(g.c_struct(), Z.data_ptr(), K, d, dt, float(beta), p.data_ptr(), a.data_ptr(),
                                    s.data_ptr(), H.data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
               "dl_aggregate_fwd")
    return H


def score_terms_available(K: int, d: int, dt: int) -> bool:
    """Whether dl_score_pairs_fwd hands out the per-factor logit terms the backward can reuse (the tuned scorer only)."""
    lib = _lib.load()
    return bool(lib.dl_has_fast_path_dtype(K, d, dt)) and not lib.dl_set_force_generic(-1)


def score_pairs_fwd(Z, H, pu, pv, t: float, pairs: PairList | None = None, want_coef: bool = False):
    """-> prob f32[P].  model.py:109-113 at the listed pairs.  ``pairs`` (the PairList the index arrays
    belong to) enables the LDS-staged, XCD-sliced kernel; without it every pair is scored on its own."""
    lib = _lib.load()
    (Z, dt), (H, dth) = _tab(Z), _tab(H)
    _need_cuda(Z, H, pu, pv)
    N, K, d = _nkd(Z)
    if H.shape != Z.shape or dt != dth:
        raise ValueError("Z and H differ in shape or storage type")
    if pu.dtype != torch.int32 or pv.dtype != torch.int32:
        raise TypeError("pair indices must be int32")
    P = int(pu.numel())
    prob = _empty(P, torch.float32, Z.device)
    by_u = pairs.c_struct_by_u() if pairs is not None else None
    # per-factor logit terms for the backward: only the tuned scorer produces them
    coef = None
    if want_coef and pairs is not None and score_terms_available(K, d, dt):
        coef = _empty((2, P, K), torch.float32, Z.device)
    _lib.check(lib.dl_score_pairs_fwd(Z.data_ptr(), H.data_ptr(), N, K, d, dt, float(t), pu.data_ptr(), pv.data_ptr(),
                                      P, by_u, prob.data_ptr(), coef.data_ptr() if coef is not None else None,
                                      _stream()), "dl_score_pairs_fwd")
    return (prob, coef) if want_coef else prob


def score_allpairs_fwd(Z, H, t: float) -> torch.Tensor:
    """-> prob f32[N,N] for all ordered pairs: model.py:109-113 as written, without a pair list."""
    lib = _lib.load()
    (Z, dt), (H, dth) = _tab(Z), _tab(H)
    _need_cuda(Z, H)
    N, K, d = _nkd(Z)
    if H.shape != Z.shape or dt != dth:
        raise ValueError("Z and H differ in shape or storage type")
    prob = _empty((N, N), torch.float32, Z.device)
    ws = _ws.get(int(lib.dl_score_allpairs_workspace_bytes(N, K, d, dt)), Z.device)
    _lib.check(lib.dl_score_allpairs_fwd(Z.data_ptr(), H.data_ptr(), N, K, d, dt, float(t), prob.data_ptr(),
                                         ws.data_ptr(), ws.numel(), _stream()), "dl_score_allpairs_fwd")
    return prob


def score_pairs_bwd(Z, H, pairs: PairList, t: float, prob, g_prob, dZ_out=None, dH_out=None, coef=None):
    """-> dZ, dH f32[N,K,d] (rows of the incidence plan are written)."""
    lib = _lib.load()
    (Z, dt), (H, _dth), prob, g_prob = _tab(Z), _tab(H), _f32c(prob), _f32c(g_prob)
    _need_cuda(Z, H, prob, g_prob, pairs.inc.rowptr)
    N, K, d = _nkd(Z)
    if prob.numel() != g_prob.numel():
        raise ValueError("prob / g_prob lengths differ")
    dZ = _empty(Z.shape, torch.float32, Z.device) if dZ_out is None else dZ_out
    dH = _empty(Z.shape, torch.float32, Z.device) if dH_out is None else dH_out
    inc = pairs.c_struct(int(prob.numel()))
    ws = _workspace(pairs.c_plan(), Z.device, K, d)
    if coef is not None and tuple(coef.shape) != (2, prob.numel(), K):
        raise ValueError("coef must be the [2, P, K] array of score_pairs_fwd for the same pair list")
    _lib.check(lib.dl_score_pairs_bwd(Z.data_ptr(), H.data_ptr(), K, d, dt, float(t), inc, prob.data_ptr(),
                                      g_prob.data_ptr(), coef.data_ptr() if coef is not None else None,
                                      dZ.data_ptr(), dH.data_ptr(), ws.data_ptr(), ws.numel(),
                                      _stream()), "dl_score_pairs_bwd")
    return dZ, dH


def one_pass_scorer_wanted(table_dtype, n_nodes: int, K: int, d: int) -> bool:
    """The training step's choice between the one-pass scorer (dl_score_pairs_train: 8 KB of partner rows per pair at K = 8,
    d = 64) and forward-with-stored-terms + two coefficient gathers (12 KB per pair) — ONE rule for model.forward_pairs_loss,
    dist.sharded_forward_loss and bench.py (DL_ONE_PASS_SCORER=0/1 forces either).  Measured (tools/score_train_time.py):
    one pass wins or ties wherever a wave-per-entry kernel exists (K in {4, 8} at d = 64; K = 16, d = 128 in both table types
    since round 5: Penn94-shaped bf16 15.9 vs 22.3 ms, fp32 29.9 vs 45.1) and for every fp32 shape (squirrel 387 vs 553 us);
    what remains for the separate kernels is a WIDE row (K d >= 2048) of bf16 tables that sit in the caches and has only
    the group-per-entry kernel (one wave per SIMD there)."""
    mode = os.environ.get("DL_ONE_PASS_SCORER", "auto")
    if mode in ("0", "1"):
        return mode == "1"
    if table_dtype == torch.float32 or K * d < 2048 or (K, d) == (16, 128):
        return True
    return 2 * n_nodes * K * d * 2 > (512 << 20)


def score_pairs_train_supported(pairs: PairList, K: int, d: int, dt: int) -> bool:
    return bool(_lib.load().dl_score_pairs_train_supported(pairs.c_struct(pairs.n_pairs), K, d, dt))


def score_pairs_train(Z, H, pairs: PairList, t: float, label, weight):
    """-> prob f32[P], dZ, dH f32[N,K,d]: scorer forward, the weighted-BCE gradient of (label, weight) and the scorer
    backward in ONE pass over the incidence plan (dl_score_pairs_train; tuned (K, d) only).  dZ / dH are the
    gradients of sum_q weight BCE(prob, label)."""
    lib = _lib.load()
    (Z, dt), (H, _dth), label, weight = _tab(Z), _tab(H), _f32c(label), _f32c(weight)
    _need_cuda(Z, H, label, weight, pairs.inc.rowptr)
    N, K, d = _nkd(Z)
    P = int(label.numel())
    if weight.numel() != P or P != pairs.n_pairs:
        raise ValueError("label / weight must cover the pair list")
    prob = _empty(P, torch.float32, Z.device)
    dZ = _empty(Z.shape, torch.float32, Z.device)
    dH = _empty(Z.shape, torch.float32, Z.device)
    pairs.bind_labels(label, weight, P)                             # per-entry labels once the same tensors come back (graph.py)
    inc = pairs.c_struct(P)
    ws = _workspace(pairs.c_plan(), Z.device, K, d)
    _lib.check(lib.dl_score_pairs_train(Z.data_ptr(), H.data_ptr(), K, d, dt, float(t), inc, label.data_ptr(),
                                        weight.data_ptr(), prob.data_ptr(), dZ.data_ptr(), dH.data_ptr(), ws.data_ptr(),
                                        ws.numel(), _stream()), "dl_score_pairs_train")
    return prob, dZ, dH


def route_aggregate_bwd(g: Graph, Z, beta: float, t: float, p, a, s, dH, dZ_accum=None) -> torch.Tensor:
    """-> dZ f32[N,K,d] (added onto ``dZ_accum`` in place when given).  Unsharded graphs only."""
    lib = _lib.load()
    (Z, dt), dH = _tab(Z), _f32c(dH)
    _need_cuda(Z, dH, g.rowptr, p, a, s)
    N, K, d = _check_rows(g, Z)
    if dZ_accum is None:
        dZ, acc = _empty(Z.shape, torch.float32, Z.device), 0
    else:
        if not dZ_accum.is_contiguous() or dZ_accum.shape != Z.shape or dZ_accum.dtype != torch.float32:
            raise ValueError("dZ_accum must be a contiguous fp32 [N,K,d] tensor")
        dZ, acc = dZ_accum, 1
    ws = _workspace(g.c_plan(), Z.device, K, d)
    _lib.check(lib.dl_route_aggregate_bwd(g.c_struct(), Z.data_ptr(), K, d, dt, float(beta), float(t), p.data_ptr(),
                                          a.data_ptr(), s.data_ptr(), dH.data_ptr(), dZ.data_ptr(), acc,
                                          ws.data_ptr(), ws.numel(), _stream()), "dl_route_aggregate_bwd")
    return dZ


def route_aggregate_bwd_phase1(g: Graph, Z, beta: float, p, a, s, dH, ds_out: torch.Tensor):
    """-> dw, dwr f32[E]; writes ds[N,K] rows of the graph (all-gather ds before phase 2 when sharded)."""
    lib = _lib.load()
    (Z, dt), dH = _tab(Z), _f32c(dH)
    _need_cuda(Z, dH, g.rowptr, p, a, s, ds_out)
    N, K, d = _check_rows(g, Z)
    dw = _empty(g.n_edges, torch.float32, Z.device)
    dwr = _empty(g.n_edges, torch.float32, Z.device)
    ws = _workspace(g.c_plan(), Z.device, K, d)
    _lib.check(lib.dl_route_aggregate_bwd_phase1(g.c_struct(), Z.data_ptr(), K, d, dt, float(beta), p.data_ptr(),
                                                 a.data_ptr(), s.data_ptr(), dH.data_ptr(), dw.data_ptr(),
                                                 dwr.data_ptr(), ds_out.data_ptr(), ws.data_ptr(), ws.numel(),
                                                 _stream()), "dl_route_aggregate_bwd_phase1")
    return dw, dwr


def route_aggregate_bwd_phase2(g: Graph, Z, beta: float, t: float, p, a, s, dH, dw, dwr, ds, dZ_out: torch.Tensor,
                               accumulate: bool):
    lib = _lib.load()
    (Z, dt), dH = _tab(Z), _f32c(dH)
    _need_cuda(Z, dH, g.rowptr, p, a, s, ds, dZ_out)
    N, K, d = _check_rows(g, Z)
    ws = _workspace(g.c_plan(), Z.device, K, d)
    _lib.check(lib.dl_route_aggregate_bwd_phase2(g.c_struct(), Z.data_ptr(), K, d, dt, float(beta), float(t),
                                                 p.data_ptr(), a.data_ptr(), s.data_ptr(), dH.data_ptr(),
                                                 dw.data_ptr(), dwr.data_ptr(), ds.data_ptr(), dZ_out.data_ptr(),
                                                 1 if accumulate else 0, ws.data_ptr(), ws.numel(), _stream()),
               "dl_route_aggregate_bwd_phase2")
    return dZ_out


def route_aggregate_bwd_scaled(g: Graph, Z, beta: float, t: float, p, a, s, dH, dZ_in, scale) -> torch.Tensor:
    """-> dZ = scale * (dZ_in + backward(dH)), a NEW tensor: dH and dZ_in are only read (they may be an autograd node's
    saved tensors), scale is a 0-dim / 1-element fp32 tensor on the device (an upstream d/dloss).  dl_route_aggregate_bwd_scaled:
    the backward is linear in (dH, dZ_in), so a caller holding UNSCALED gradients needs no scaling passes."""
    lib = _lib.load()
    (Z, dt), dH, dZ_in = _tab(Z), _f32c(dH), _f32c(dZ_in)
    scale = scale.to(device=Z.device, dtype=torch.float32).reshape(1)
    _need_cuda(Z, dH, dZ_in, g.rowptr, p, a, s)
    N, K, d = _check_rows(g, Z)
    if dZ_in.shape != Z.shape or dH.shape != Z.shape:
        raise ValueError("dH and dZ_in must be [N,K,d] like Z")
    dZ = _empty(Z.shape, torch.float32, Z.device)
    ws = _workspace(g.c_plan(), Z.device, K, d)
    _lib.check(lib.dl_route_aggregate_bwd_scaled(g.c_struct(), Z.data_ptr(), K, d, dt, float(beta), float(t), p.data_ptr(),
                                                 a.data_ptr(), s.data_ptr(), dH.data_ptr(), dZ_in.data_ptr(),
                                                 scale.data_ptr(), dZ.data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
               "dl_route_aggregate_bwd_scaled")
    return dZ


class _PaddedFeatures:
    """Zero-padded copies of feature matrices whose row length is not a multiple of 4, one per source TENSOR and version
    counter.  Entries are keyed by the tensor OBJECT (its id, checked against a weak reference: an address — or an id —
    can be reused by another tensor once this one is gone, a live object cannot) and are dropped when their tensor dies.
    The padded copy is what the kernels — and any captured HIP graph — read, so it must live exactly as long as the
    caller's x does: with one entry per tensor, a second model, an evaluation on other features or a second run in the
    same process cannot evict the copy a live graph still replays (train._graphed_epoch also holds it)."""

    def __init__(self):
        self._by_id: dict = {}

    def get(self, x: torch.Tensor, Fp: int) -> torch.Tensor:
        key = id(x)
        hit = self._by_id.get(key)
        if hit is None or hit[0]() is not x or hit[1] != x._version or hit[2].shape[1] != Fp:
            ref = weakref.ref(x, lambda _r, k=key, d=self._by_id: d.pop(k, None) if d.get(k, (None,))[0] is _r else None)
            hit = (ref, x._version, torch.nn.functional.pad(x, (0, Fp - x.shape[1])))
            self._by_id[key] = hit
        return hit[2]


_xpad = _PaddedFeatures()


def padded_features(x: torch.Tensor) -> torch.Tensor:
    """The tensor project_fwd / project_bwd actually read for x (x itself when its rows are 16-byte aligned)."""
    F = x.shape[1]
    Fp = (F + 3) // 4 * 4
    return x if Fp == F else _xpad.get(x, Fp)


class _XPlanes:
    """Persistent bf16 planes of a feature matrix and of its transpose (dl_project_xplanes_build), one entry per feature
    TENSOR (object identity + version counter, like _PaddedFeatures): model.py:106 evaluates the MLPs on the same x every
    epoch, and the projection kernels otherwise split x (forward) and x^T (backward) into their three bf16 planes on every
    call.  An entry is made the SECOND time a tensor is seen (a one-off call splits inside its own workspace, as before),
    only for graphs the kernels process as one node block, and dies with its tensor.
    CONTRACT: the planes follow the tensor's identity and VERSION COUNTER — a write that bypasses the counter (``x.data``,
    a raw-pointer kernel) leaves stale planes in use; write through torch (any in-place op bumps the version) or set
    DL_X_PLANES=0.  Up to MAX_BYTES (1 GiB, about 3x the size of x) of device memory per cached tensor."""
    MAX_ROWS = 2 << 17              # fwd_block_rows: larger graphs are processed in node blocks that re-split themselves
    MAX_BYTES = 1 << 30

    def __init__(self):
        self._by_id: dict = {}

    def get(self, x: torch.Tensor, force: bool = False):
        if os.environ.get("DL_X_PLANES", "1") == "0" or x.shape[0] > self.MAX_ROWS or not x.is_cuda:
            return None
        # no version counter to watch (inference-mode tensors raise on ._version), or a view made for this call (a row chunk of
        # the sharded path: a new tensor object every epoch, it would never be seen twice): split per call, as before the cache
        if x.is_inference() or x._base is not None:
            return None
        key = id(x)
        hit = self._by_id.get(key)
        if hit is not None and (hit[0]() is not x or hit[1] != x._version):
            hit = None
        if hit is None:
            ref = weakref.ref(x, lambda _r, k=key, d=self._by_id: d.pop(k, None) if d.get(k, (None,))[0] is _r else None)
            hit = (ref, x._version, None)
            self._by_id[key] = hit
            if not force:
                return None                                         # first sight: remember the tensor, build next time
        if hit[2] is None:
            lib = _lib.load()
            N, F = x.shape
            nbytes = int(lib.dl_project_xplanes_bytes(N, F))
            if nbytes == 0 or nbytes > self.MAX_BYTES:
                return None
            buf = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            _lib.check(lib.dl_project_xplanes_build(x.data_ptr(), N, F, buf.data_ptr(), nbytes, _stream()), "dl_project_xplanes_build")
            hit = (hit[0], hit[1], buf)
            self._by_id[key] = hit
        return hit[2]


_xplanes = _XPlanes()


def xplanes_for(x: torch.Tensor, force: bool = False):
    """The persistent planes of the tensor the projection kernels read for x (its zero-padded copy when F % 4 != 0), or
    None (first sight of the tensor, a blocked graph, DL_X_PLANES=0).  force=True builds them now (before a graph capture)."""
    return _xplanes.get(padded_features(_f32c(x)), force)


def _pad_features(x, W1):
    """Rows of F floats with F % 4 != 0 have no common 16-byte alignment, which would force the kernels onto
    scalar loads.  Zero-pad the feature axis of x (once per tensor: the features are constant data) and of W1
    (per call: one small copy) to a multiple of 4; zero columns add nothing to any product."""
    F = x.shape[1]
    Fp = (F + 3) // 4 * 4
    if Fp == F:
        return x, W1
    return padded_features(x), torch.nn.functional.pad(W1, (0, Fp - F))


def keep_hidden(N: int, F: int, K: int, nhid: int) -> bool:
    """Should the forward keep the hidden layer for the backward?  Recomputing it costs 2*F FLOP per hidden unit
    against 8 bytes of traffic: measured faster at every width tried (F = 128: fwd+bwd 0.35 -> 0.30 ms, F = 2089:
    2.9 -> 2.0 ms, tools/project_keep_times.py; snap-patents-sized, 45 GiB of hidden layer: epoch 445 -> 393 ms) — while
    the [K,nhid,N] buffer stays within a quarter of the device's memory, at most 64 GiB (of 288 GB on an MI355X); past
    that the recompute is what keeps large graphs in memory."""
    mode = os.environ.get("DL_KEEP_HIDDEN", "auto")
    if mode in ("0", "1"):
        return mode == "1"
    return N * K * nhid * 4 <= _keep_hidden_limit()


_KEEP_LIMIT = None


def _keep_hidden_limit() -> int:
    global _KEEP_LIMIT
    if _KEEP_LIMIT is None:
        total = torch.cuda.get_device_properties(torch.cuda.current_device()).total_memory if torch.cuda.is_available() else 0
        _KEEP_LIMIT = min(64 << 30, total // 4)
    return _KEEP_LIMIT


_TILE_WIDTHS = (32, 64, 128)          # factor widths d the matrix-core projection kernels are instantiated for


def project_tile_width(d: int) -> int | None:
    """The kernels' own widths serve themselves; any other d <= 128 runs at the next width with zero-padded output
    weights (zero rows of the LAST layer: they add nothing to any product and cost (dp - d) / dp of the smaller GEMM);
    None beyond 128 (the library GEMMs then)."""
    for w in _TILE_WIDTHS:
        if d <= w:
            return w
    return None


def _pad_out_rows(W, b, dp):
    """Zero-pad dim 1 of W [K,d,*] and b [K,d] to dp."""
    d = W.shape[1]
    return torch.nn.functional.pad(W, (0, 0, 0, dp - d)), torch.nn.functional.pad(b, (0, dp - d))


def project_fwd(x, W1, b1, W2=None, b2=None, pad: bool = True, keep_hid: bool = False):
    """Z [N,K,d] = K MLPs of x on the matrix cores.  Two-layer: W1 [K,nhid,F], b1 [K,nhid], W2 [K,d,nhid],
    b2 [K,d]; single layer: W1 [K,d,F], b1 [K,d], W2 = b2 = None.  model.py:13-15 / 24-27 / 106.
    keep_hid=True (two-layer): returns (Z, hid) with hid the kept hidden layer for project_bwd.
    Any d <= 128: widths other than 32 / 64 / 128 run at the next of those (project_tile_width)."""
    lib = _lib.load()
    x, W1, b1 = _f32c(x), _f32c(W1), _f32c(b1)
    _need_cuda(x, W1, b1)
    if W1.shape[2] != x.shape[1]:
        raise ValueError("W1 does not match the feature count of x")
    d_true = W1.shape[1] if W2 is None else W2.shape[1]
    dp = project_tile_width(d_true)
    if dp is None:
        raise _lib.DisenlinkHipError(f"projection kernels serve d <= {_TILE_WIDTHS[-1]}, got {d_true}")
    if dp != d_true:                                           # run at the tile width, hand back the first d columns
        if W2 is None:
            W1p, b1p = _pad_out_rows(W1, b1, dp)
            return project_fwd(x, W1p, b1p, None, None, pad=pad)[:, :, :d_true].contiguous()
        W2p, b2p = _pad_out_rows(_f32c(W2), _f32c(b2), dp)
        out = project_fwd(x, W1, b1, W2p, b2p, pad=pad, keep_hid=keep_hid)
        return (out[0][:, :, :d_true].contiguous(), out[1]) if keep_hid else out[:, :, :d_true].contiguous()
    if pad:
        x, W1 = _pad_features(x, W1)
    N, F = x.shape
    K = W1.shape[0]
    if W2 is None:
        d, nhid, w2p, b2p = W1.shape[1], 1, None, None
    else:
        W2, b2 = _f32c(W2), _f32c(b2)
        d, nhid, w2p, b2p = W2.shape[1], W1.shape[1], W2.data_ptr(), b2.data_ptr()
        if W2.shape != (K, d, nhid) or b2.shape != (K, d) or b1.shape != (K, nhid):
            raise ValueError("inconsistent projection weight shapes")
    if W1.shape[2] != F:
        raise ValueError("W1 does not match the feature count of x")
    Z = _empty((N, K, d), torch.float32, x.device)
    hid = None
    if keep_hid and W2 is not None:
        hid = _empty(int(lib.dl_project_hidden_floats(N, K, nhid)), torch.float32, x.device)
    ws = _ws.get(int(lib.dl_project_fwd_workspace_bytes(N, F, K, nhid, d, int(W2 is not None))), x.device)
    xp = _xplanes.get(x) if W2 is not None else None           # x's planes, split once for the run (second call on)
    _lib.check(lib.dl_project_fwd_xp(x.data_ptr(), N, F, K, nhid, d, W1.data_ptr(), b1.data_ptr(), w2p, b2p,
                                     Z.data_ptr(), hid.data_ptr() if hid is not None else None, ws.data_ptr(), ws.numel(),
                                     xp.data_ptr() if xp is not None else None, _stream()), "dl_project_fwd")
    return (Z, hid) if keep_hid else Z


def project_bwd(x, W1, b1, W2, dZ, pad: bool = True, hid=None, one_allocation: bool = True):
    """Weight / bias gradients of the projection from dZ [N,K,d]: (dW1, db1, dW2, db2), shaped like the weights
    (dW2 = db2 = None for the single layer).  autograd of model.py:13-15 / 24-27.  hid: the hidden layer kept
    by project_fwd(keep_hid=True), else it is recomputed.  one_allocation: the gradients are views of ONE flat buffer (the
    sharded step all-reduces it as it is); False gives four separate tensors (registered operators may not return
    outputs that share storage)."""
    lib = _lib.load()
    x, W1, b1, dZ = _f32c(x), _f32c(W1), _f32c(b1), _f32c(dZ)
    _need_cuda(x, W1, b1, dZ)
    if W1.shape[2] != x.shape[1]:
        raise ValueError("inconsistent projection shapes")
    d_true = W1.shape[1] if W2 is None else W2.shape[1]
    dp = project_tile_width(d_true)
    if dp is None:
        raise _lib.DisenlinkHipError(f"projection kernels serve d <= {_TILE_WIDTHS[-1]}, got {d_true}")
    if dp != d_true:                                           # zero-padded output columns carry zero gradient in
        if dZ.shape[2] != d_true:
            raise ValueError("inconsistent projection shapes")
        dZp = torch.nn.functional.pad(dZ, (0, dp - d_true))
        if W2 is None:
            W1p, b1p = _pad_out_rows(W1, b1, dp)
            dW1, db1, _n1, _n2 = project_bwd(x, W1p, b1p, None, dZp, pad=pad, one_allocation=one_allocation)
            return dW1[:, :d_true].contiguous(), db1[:, :d_true].contiguous(), None, None
        W2p = torch.nn.functional.pad(_f32c(W2), (0, 0, 0, dp - d_true))
        dW1, db1, dW2, db2 = project_bwd(x, W1, b1, W2p, dZp, pad=pad, hid=hid, one_allocation=one_allocation)
        return dW1, db1, dW2[:, :d_true].contiguous(), db2[:, :d_true].contiguous()
    F_true = x.shape[1]
    if pad:
        x, W1 = _pad_features(x, W1)
    N, F = x.shape
    K = W1.shape[0]
    two = W2 is not None
    if two:
        W2 = _f32c(W2)
        d, nhid = W2.shape[1], W1.shape[1]
    else:
        d, nhid = W1.shape[1], 1
    if dZ.shape != (N, K, d) or W1.shape[2] != F:
        raise ValueError("inconsistent projection shapes")
    # the (up to) four gradients are carved out of ONE allocation, back to back: the sharded training step all-reduces
    # them as one flat tensor without packing (dist.allreduce_gradients), every piece 16-byte aligned
    # (with a zero-padded feature axis the kernels write dW1 at the padded width into a buffer of its own and the columns
    # that exist are copied into the first piece — the one copy the trim costs anyway — so the four gradients stay
    # adjacent at EVERY feature width: snap-patents' F = 269 gave four collectives in round 4)
    trimmed = F != F_true
    shapes = [tuple(W1.shape[:2]) + (F_true,), tuple(b1.shape)] + ([tuple(W2.shape), (K, d)] if two else [])
    sizes = [int(torch.Size(sh).numel()) for sh in shapes]
    if one_allocation and all(n % 4 == 0 for n in sizes):
        flat = _empty((sum(sizes),), torch.float32, x.device)
        parts, off = [], 0
        for sh, n in zip(shapes, sizes):
            parts.append(flat[off:off + n].view(sh))
            off += n
    else:
        parts = [_empty(sh, torch.float32, x.device) for sh in shapes]
    dW1_out = parts[0]
    dW1 = _empty(tuple(W1.shape), torch.float32, x.device) if trimmed else dW1_out
    db1 = parts[1]
    dW2, db2 = (parts[2], parts[3]) if two else (None, None)
    ws = _ws.get(int(lib.dl_project_bwd_workspace_bytes(N, F, K, nhid, d, int(two))), x.device)
    xp = _xplanes.get(x) if two else None
    _lib.check(lib.dl_project_bwd_xp(x.data_ptr(), N, F, K, nhid, d, W1.data_ptr(), b1.data_ptr(),
                                     W2.data_ptr() if two else None, dZ.data_ptr(),
                                     hid.data_ptr() if (two and hid is not None) else None, dW1.data_ptr(), db1.data_ptr(),
                                     dW2.data_ptr() if two else None, db2.data_ptr() if two else None,
                                     ws.data_ptr(), ws.numel(), xp.data_ptr() if xp is not None else None, _stream()),
               "dl_project_bwd")
    if trimmed:
        dW1_out.copy_(dW1[..., :F_true])
    return dW1_out, db1, dW2, db2


def project_supported(d: int) -> bool:
    """Do the matrix-core projection kernels serve factor width d?  (Natively 32 / 64 / 128; every other d <= 128 at the
    next of those widths through zero-padded output weights, project_tile_width.)"""
    return project_tile_width(int(d)) is not None


# ---------------------------------------------------------------------- autograd
class Project(torch.autograd.Function):
    """Z = MLP_k(x) for all k.  Forward: the fused MFMA kernel (hidden activations never leave the
    register file).  Backward: dl_project_bwd (hidden layer recomputed on the matrix cores); x is data and
    gets no gradient."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2):
        ctx.save_for_backward(x, W1, b1, W2 if W2 is not None else x.new_empty(0))
        ctx.two_layer = W2 is not None
        keep = ctx.two_layer and any(ctx.needs_input_grad) and keep_hidden(x.shape[0], x.shape[1], W1.shape[0], W1.shape[1])
        if keep:
            Z, ctx.hid = project_fwd(x, W1, b1, W2, b2, keep_hid=True)
            return Z
        ctx.hid = None
        return project_fwd(x, W1, b1, W2, b2)

    @staticmethod
    def backward(ctx, dZ):
        x, W1, b1, W2 = ctx.saved_tensors
        dW1, db1, dW2, db2 = _project_grads(x, W1, b1, W2 if ctx.two_layer else None, dZ.contiguous(), ctx.hid)
        ctx.hid = None
        return None, dW1, db1, dW2, db2


class RouteAggregate(torch.autograd.Function):
    """Z [N,K,d] -> H [N,K,d]: Disentangle_layer.forward (model.py:55-77) on the CSR of adj."""

    @staticmethod
    def forward(ctx, Z, graph: Graph, beta: float, t: float):
        Z = _f32c(Z)
        p, a, s = route_fwd(graph, Z, t)
        H = aggregate_fwd(graph, Z, beta, p, a, s)
        ctx.graph, ctx.beta, ctx.t = graph, beta, t
        ctx.save_for_backward(Z, a, s)
        ctx.p = p
        return H

    @staticmethod
    def backward(ctx, dH):
        Z, a, s = ctx.saved_tensors
        dZ = route_aggregate_bwd(ctx.graph, Z, ctx.beta, ctx.t, ctx.p, a, s, dH.contiguous())
        return dZ, None, None, None


def _project_grads(x, W1, b1, W2, dZ, hid=None):
    """Backward of the projection on stacked weights -> (dW1, db1, dW2, db2): the MFMA kernels of
    dl_project_bwd (env DL_PROJECT_BWD=library selects the library-GEMM form kept for timing comparisons)."""
    if os.environ.get("DL_PROJECT_BWD", "native") != "library":
        return project_bwd(x, W1, b1, W2, dZ, hid=hid)
    if W2 is None:                                              # Z[n,k,:] = W1[k] x[n] + b1[k]
        return torch.einsum("nkd,nf->kdf", dZ, x), dZ.sum(dim=0), None, None
    pre = torch.einsum("nf,khf->nkh", x, W1) + b1               # [N,K,nhid]
    hid = torch.relu(pre)
    dW2 = torch.einsum("nkd,nkh->kdh", dZ, hid)
    dhid = torch.einsum("nkd,kdh->nkh", dZ, W2) * (pre > 0)
    return torch.einsum("nkh,nf->khf", dhid, x), dhid.sum(dim=0), dW2, dZ.sum(dim=0)


class ProjectStacked(torch.autograd.Function):
    """Projection over the module's shared [K, ...] parameter buffers (see Disentangle._restack): the K
    per-factor Parameters are the autograd inputs, the kernel reads their common storage — no stacking copy —
    and each parameter's gradient is a view of the stacked gradient."""

    @staticmethod
    def forward(ctx, x, bufs, K, *params):
        W1, b1, W2, b2 = bufs
        ctx.bufs, ctx.K = bufs, K
        ctx.save_for_backward(x)
        ctx.hid = None
        if W2 is not None and any(ctx.needs_input_grad) and keep_hidden(x.shape[0], x.shape[1], K, W1.shape[1]):
            Z, ctx.hid = project_fwd(x, W1, b1, W2, b2, keep_hid=True)
            return Z
        return project_fwd(x, W1, b1, W2, b2)

    @staticmethod
    def backward(ctx, dZ):
        (x,) = ctx.saved_tensors
        W1, b1, W2, b2 = ctx.bufs
        dW1, db1, dW2, db2 = _project_grads(x, W1, b1, W2, dZ.contiguous(), ctx.hid)
        ctx.hid = None
        K = ctx.K
        grads = list(dW1.unbind(0)) + list(db1.unbind(0))
        if W2 is not None:
            grads += list(dW2.unbind(0)) + list(db2.unbind(0))
        assert len(grads) == (2 if W2 is None else 4) * K
        return (None, None, None, *grads)


class PairBCE(torch.autograd.Function):
    """loss = sum_q w[q] BCE(prob[q], label[q]) with torch's clamps, loss and d loss / d prob from ONE kernel
    (main_disentangled.py:195 on pair lists; see metrics.pair_bce_weights)."""

    @staticmethod
    def forward(ctx, prob, label, weight):
        lib = _lib.load()
        prob, label, weight = _f32c(prob), _f32c(label), _f32c(weight)
        _need_cuda(prob, label, weight)
        if not (prob.numel() == label.numel() == weight.numel()):
            raise ValueError("prob, label and weight differ in length")
        loss = _empty(1, torch.float32, prob.device)
        g = _empty_like(prob)
        ws = _ws.get(8192, prob.device)
        _lib.check(lib.dl_pair_bce(prob.data_ptr(), label.data_ptr(), weight.data_ptr(), prob.numel(), loss.data_ptr(),
                                   g.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "dl_pair_bce")
        ctx.save_for_backward(g)
        return loss[0]

    @staticmethod
    def backward(ctx, g_loss):
        (g,) = ctx.saved_tensors
        return g * g_loss, None, None


class HotPathPairs(torch.autograd.Function):
    """Z [N,K,d] fp32 -> (emb [N,K,d] fp32, prob [P]): route + aggregate + pair scorer as ONE autograd node.
    ``table_dtype`` (torch.float32 / torch.bfloat16) is the storage type of the gathered Z and H tables;
    arithmetic and every gradient stay fp32 (the cast of Z happens inside, so dZ comes back in fp32)."""

    @staticmethod
    def forward(ctx, Z, graph: Graph, pairs: PairList, beta: float, t: float, table_dtype):
        ctx.set_materialize_grads(False)        # an unused output (emb in the training loss) arrives as None, not zeros
        Zt = _f32c(Z) if table_dtype == torch.float32 else _f32c(Z).to(table_dtype)
        p, a, s = route_fwd(graph, Zt, t)
        H = aggregate_fwd(graph, Zt, beta, p, a, s)
        if ctx.needs_input_grad[0]:
            prob, coef = score_pairs_fwd(Zt, H, pairs.pu, pairs.pv, t, pairs, want_coef=True)
        else:
            prob, coef = score_pairs_fwd(Zt, H, pairs.pu, pairs.pv, t, pairs), None
        ctx.graph, ctx.pairs, ctx.beta, ctx.t, ctx.p, ctx.coef = graph, pairs, beta, t, p, coef
        ctx.save_for_backward(Zt, H, a, s, prob)
        return H.float(), prob

    @staticmethod
    def backward(ctx, g_emb, g_prob):
        Zt, H, a, s, prob = ctx.saved_tensors
        g_prob = torch.zeros_like(prob) if g_prob is None else g_prob.contiguous()
        dZ, dH = score_pairs_bwd(Zt, H, ctx.pairs, ctx.t, prob, g_prob, coef=ctx.coef)
        if g_emb is not None:
            dH += g_emb
        route_aggregate_bwd(ctx.graph, Zt, ctx.beta, ctx.t, ctx.p, a, s, dH, dZ_accum=dZ)
        return dZ, None, None, None, None, None


class HotPathPairsLoss(torch.autograd.Function):
    """Z [N,K,d] fp32 -> (emb, prob [P], loss): route + aggregate + pair scorer + weighted BCE of (label, weight) as ONE
    autograd node whose scorer part runs forward AND backward in a single pass (score_pairs_train): the gradients of
    the loss w.r.t. the scorer's inputs are ready when the forward returns, and the backward only scales them by the
    incoming d/dloss and continues into aggregation and routing.  A gradient arriving on ``prob`` (another loss on the
    same scores) is added through the ordinary scorer backward."""

    @staticmethod
    def forward(ctx, Z, graph: Graph, pairs: PairList, beta: float, t: float, table_dtype, label, weight):
        ctx.set_materialize_grads(False)
        lib = _lib.load()
        Zt = _f32c(Z) if table_dtype == torch.float32 else _f32c(Z).to(table_dtype)
        p, a, s = route_fwd(graph, Zt, t)
        H = aggregate_fwd(graph, Zt, beta, p, a, s)
        prob, dZs, dHs = score_pairs_train(Zt, H, pairs, t, label, weight)
        loss = _empty(1, torch.float32, prob.device)
        g = _empty_like(prob)                                   # dl_pair_bce's gradient output: the loss VALUE is what is used
        ws = _ws.get(8192, prob.device)
        _lib.check(lib.dl_pair_bce(prob.data_ptr(), _f32c(label).data_ptr(), _f32c(weight).data_ptr(), prob.numel(),
                                   loss.data_ptr(), g.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "dl_pair_bce")
        ctx.graph, ctx.pairs, ctx.beta, ctx.t, ctx.p = graph, pairs, beta, t, p
        ctx.save_for_backward(Zt, H, a, s, prob, dZs, dHs)
        return H.float(), prob, loss[0]

    @staticmethod
    def backward(ctx, g_emb, g_prob, g_loss):
        Zt, H, a, s, prob, dZs, dHs = ctx.saved_tensors
        if g_loss is not None and g_prob is None and g_emb is None and g_loss.numel() == 1:
            # the usual case (loss.backward()): everything downstream is linear in the scorer's gradients, so d/dloss
            # scales the RESULT inside the last kernel — no scaling passes over the two [N,K,d] arrays, which stay unwritten
            return (route_aggregate_bwd_scaled(ctx.graph, Zt, ctx.beta, ctx.t, ctx.p, a, s, dHs, dZs, g_loss),
                    None, None, None, None, None, None, None)
        if g_loss is not None:
            dZ, dH = dZs * g_loss, dHs * g_loss
        else:
            dZ, dH = torch.zeros_like(dZs), torch.zeros_like(dHs)
        if g_prob is not None:                                   # some other function of the scores
            dZ2, dH2 = score_pairs_bwd(Zt, H, ctx.pairs, ctx.t, prob, g_prob.contiguous())
            dZ += dZ2
            dH += dH2
        if g_emb is not None:
            dH += g_emb
        route_aggregate_bwd(ctx.graph, Zt, ctx.beta, ctx.t, ctx.p, a, s, dH, dZ_accum=dZ)
        return dZ, None, None, None, None, None, None, None


class ScorePairs(torch.autograd.Function):
    """(Z, H) -> prob[P] at the listed pairs: model.py:109-113 + sigmoid."""

    @staticmethod
    def forward(ctx, Z, H, pairs: PairList, t: float):
        Z, H = _f32c(Z), _f32c(H)
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        prob, coef = score_pairs_fwd(Z, H, pairs.pu, pairs.pv, t, pairs, want_coef=True) if need else \
            (score_pairs_fwd(Z, H, pairs.pu, pairs.pv, t, pairs), None)
        ctx.pairs, ctx.t, ctx.coef = pairs, t, coef
        ctx.save_for_backward(Z, H, prob)
        return prob

    @staticmethod
    def backward(ctx, g_prob):
        Z, H, prob = ctx.saved_tensors
        dZ, dH = score_pairs_bwd(Z, H, ctx.pairs, ctx.t, prob, g_prob.contiguous(), coef=ctx.coef)
        return dZ, dH, None, None


class DensePairPlanCache:
    """Pair plan of the dense [N,N] score gradient's support — owned by ONE module (no process-wide state).

    The reference's caller takes its loss on ``link_pred[mask == 1]`` with masks that are fixed for a run
    (main_disentangled.py:167-190,195), so d loss / d link_pred can be non-zero only on the masks' support.  The plan
    must be keyed on THAT support, not on the entries of the gradient that happen to be non-zero: a saturated positive
    (p == 1.0 in fp32, y == 1) has exactly zero BCE gradient in one epoch and a non-zero one in the next (SURVEY.md §0
    finding 4), so the gradient's non-zero set moves under fixed masks.

      * ``set_pairs(masks ...)`` (Disentangle.set_loss_pairs / assume_static_loss_masks(mask, ...)): the plan IS the
        support of the caller's masks; entries whose gradient is zero cost a little arithmetic and add exactly zero.
      * no masks given: the plan is learnt — first from the caller's own INDEXING of link_pred (round 6: forward returns
        a LinkPred, a Tensor subclass whose only difference is that ``link_pred[mask]`` / ``link_pred[rows, cols]`` tells
        this cache the support of the index before handing over to torch: main_disentangled.py:195, 202 index it with
        the fixed masks every epoch, so after the first epoch the plan is their union and never changes again), and, as
        the safety net behind that, from the gradients: a backward whose non-zero entries are not all inside the cached
        set extends it by the new ones (union) and rebuilds; it never shrinks to the current non-zero set.  (Learning
        from the gradients ALONE rebuilt the plan in every one of the first 40 epochs on squirrel and chameleon — each
        epoch desaturates a few more pairs — 4 ms per epoch: profiles/r7d_dropin_*.)
      * validation: non-zeros(g) must be a SUBSET of the plan.  Default: two device counters are read back per backward
        (one 16-byte host read; the reference's own loop reads back the validation scores and the loss every epoch,
        main_disentangled.py:204,214).  ``static=True`` (needs masks): no host read — the same two counters stay on
        the device and become a validity factor, 1 if the subset relation holds, NaN if not, multiplied into the
        gradients: a caller that promised fixed masks and takes a loss elsewhere gets NaN gradients at once, never
        silently wrong ones."""

    def __init__(self):
        self.pairs = None        # PairList of the cached index set
        self.flat = None         # int64 [n] row-major positions, ascending
        self.key = None          # (N, device)
        self.static = False
        self.from_masks = False  # the set is the caller's declared support (never extended behind their back)
        self.rebuilds = 0        # how often the plan was (re)built: a learnt plan is rebuilt whenever new entries carry gradient
        self._seen_index = set() # fingerprints of the index masks link_pred was indexed with (note_index)
        self._seen_ij = []       # (rows, cols) index tensors seen: weak references + version counters
        self._pending = []       # flat positions learnt from indexing, not yet in the plan
        self._hash_vec = None

    # ---- the caller's declared support
    def set_pairs(self, N: int, device, *supports):
        """``supports``: dense [N,N] masks (entries != 0 belong to the support: the caller's summed train masks hold
        2, 3, ... where index pairs repeat, main_disentangled.py:176-179 — those entries are outside the loss and get
        zero gradient, which is harmless here) and / or (rows, cols) index tuples.  The plan is the union."""
        flats = []
        for sup in supports:
            if isinstance(sup, (tuple, list)):
                r, c = (torch.as_tensor(v, device=device).reshape(-1).long() for v in sup)
                if r.numel() != c.numel():
                    raise ValueError("(rows, cols) differ in length")
                if r.numel() and (int(r.min()) < 0 or int(c.min()) < 0 or int(r.max()) >= N or int(c.max()) >= N):
                    raise ValueError("pair index outside [0, N)")
                flats.append(r * N + c)
            else:
                m = torch.as_tensor(sup, device=device)
                if m.dim() != 2 or m.shape[0] != N or m.shape[1] != N:
                    raise ValueError(f"a loss mask must be [N, N] = [{N}, {N}], got {tuple(m.shape)}")
                flats.append(torch.nonzero(m.reshape(-1)).reshape(-1))
        if not flats:
            raise ValueError("no loss masks / pairs given")
        self._install(torch.unique(torch.cat(flats)), N, device)
        self.from_masks = True

    def clear(self):
        self.pairs = self.flat = self.key = None
        self.from_masks = False
        self._seen_index, self._seen_ij, self._pending = set(), [], []

    # ---- learnt from the caller's indexing of link_pred (LinkPred.__torch_function__)
    def note_index(self, N: int, index) -> None:
        """``link_pred[index]`` is about to be evaluated: remember the support of a boolean [N,N] mask or of a
        (rows, cols) pair of index tensors.  A mask is recognised by a fingerprint (count, hashed row sums, hashed
        column sums: two reductions over the mask and one 24-byte read — the indexing that follows synchronises anyway),
        so the masks the loop passes every epoch cost a nonzero() once."""
        if self.from_masks or self.static:
            return                                                  # the caller declared the support: nothing to learn
        try:
            if torch.is_tensor(index) and index.dtype == torch.bool and tuple(index.shape) == (N, N) and index.is_cuda:
                # the mask's bytes as 8-byte words against a fixed random int64 vector (wrapping multiply-add): one fused pass over
                # N*N bytes (the first form — row and column sums of the bool matrix — cost 1.2 ms per epoch on squirrel)
                flat = index.reshape(-1)
                n8 = flat.numel() // 8
                if self._hash_vec is None or self._hash_vec.numel() != n8 or self._hash_vec.device != index.device:
                    g = torch.Generator().manual_seed(0x5eed)
                    self._hash_vec = torch.randint(-(1 << 62), 1 << 62, (n8,), generator=g, dtype=torch.int64).to(index.device)
                words = flat[:n8 * 8].view(torch.int64)
                fp = tuple(torch.stack([(words * self._hash_vec).sum(),
                                        words.sum(), flat[n8 * 8:].sum()]).tolist())
                if fp in self._seen_index:
                    return
                self._seen_index.add(fp)
                self._pending.append(torch.nonzero(index.reshape(-1)).reshape(-1))
            elif isinstance(index, tuple) and len(index) == 2 and all(torch.is_tensor(v) and v.dtype == torch.int64 and v.dim() == 1
                                                                        for v in index) and index[0].is_cuda:
                # recognised by the tensor OBJECTS and their version counters (no device read: this form of indexing does not
                # synchronise) — not by addresses: index tensors made afresh every epoch come back at the address of the ones
                # just freed, with other contents
                rows, cols = index
                for r0, r1, v0, v1 in self._seen_ij:
                    if r0() is rows and r1() is cols and (v0, v1) == (rows._version, cols._version):
                        return
                self._seen_ij.append((weakref.ref(rows), weakref.ref(cols), rows._version, cols._version))
                del self._seen_ij[:-16]
                self._pending.append((rows % N) * N + (cols % N))
            if len(self._pending) > 32:                             # many gathers between two backwards: keep one merged set
                self._pending = [torch.unique(torch.cat(self._pending))]
        except RuntimeError:                                        # an index torch itself will reject: let torch say so
            return

    def _install(self, flat: torch.Tensor, N: int, device):
        self.flat = flat
        # (only the incidence plan is walked by dl_score_allpairs_bwd: the forward plan of the list is not built)
        self.pairs = PairList.build(torch.div(flat, N, rounding_mode="floor"), flat % N, N, build_by_u=False)
        self.key = (N, device)
        self.rebuilds += 1

    # ---- per backward
    def lookup(self, g_prob: torch.Tensor):
        """-> (pairs, validity factor or None)"""
        N = g_prob.shape[0]
        g_flat = g_prob.reshape(-1)
        key = (N, g_prob.device)
        if self.pairs is not None and self.key != key:
            if self.from_masks:
                raise ValueError(f"the loss pairs were declared for {self.key}, the gradient is for {key}")
            self.clear()
        if self._pending and not self.from_masks:                   # supports learnt from the caller's indexing since the last backward
            known = [self.flat] if self.flat is not None and self.key == key else []
            self._install(torch.unique(torch.cat(known + self._pending)), N, g_prob.device)
            self._pending = []
        if self.pairs is None:
            if self.static:
                raise RuntimeError("assume_static_loss_masks() needs the masks (or index pairs) the loss is taken on: "
                                   "the support of a loss cannot be read off one gradient")
            self._install(torch.nonzero(g_flat).reshape(-1), N, g_prob.device)
            return self.pairs, None
        # non-zeros(g) inside the plan == non-zeros(g) anywhere  <=>  subset
        probe = torch.stack([torch.count_nonzero(g_flat), torch.count_nonzero(g_flat[self.flat])])
        if self.static:
            return self.pairs, torch.where(probe[0] == probe[1], 1.0, float("nan")).to(torch.float32)
        cnt, kept = probe.tolist()
        if cnt != kept:
            if self.from_masks:
                raise RuntimeError(f"{cnt - kept} entries of d loss / d link_pred lie outside the declared loss pairs "
                                   "(Disentangle.set_loss_pairs): declare every mask the loss indexes link_pred with")
            self._install(torch.unique(torch.cat([self.flat, torch.nonzero(g_flat).reshape(-1)])), N, g_prob.device)
        return self.pairs, None


def score_allpairs_bwd(Z, H, pairs: PairList, t: float, prob, g_prob):
    """-> dZ, dH f32[N,K,d]: backward of score_allpairs_fwd on the fixed pair plan ``pairs`` (the support of the
    caller's loss masks); prob and g_prob are the dense [N,N] arrays."""
    lib = _lib.load()
    (Z, dt), (H, _dth), prob, g_prob = _tab(Z), _tab(H), _f32c(prob), _f32c(g_prob)
    _need_cuda(Z, H, prob, g_prob, pairs.inc.rowptr)
    N, K, d = _nkd(Z)
    if tuple(prob.shape) != (N, N) or tuple(g_prob.shape) != (N, N):
        raise ValueError("prob / g_prob must be the dense [N, N] arrays")
    dZ = _empty(Z.shape, torch.float32, Z.device)
    dH = _empty(Z.shape, torch.float32, Z.device)
    P = pairs.n_pairs
    ws = _workspace(pairs.c_plan(), Z.device, K, d)
    _lib.check(lib.dl_score_allpairs_bwd(Z.data_ptr(), H.data_ptr(), N, K, d, dt, float(t), pairs.c_struct(P),
                                         pairs.pu.data_ptr(), pairs.pv.data_ptr(), P, prob.data_ptr(), g_prob.data_ptr(),
                                         dZ.data_ptr(), dH.data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
               "dl_score_allpairs_bwd")
    return dZ, dH


class LinkPred(torch.Tensor):
    """The dense link_pred of Disentangle.forward(x, adj): a torch.Tensor in every respect, except that indexing it —
    ``a_pred[pos_train_adj == 1]`` (main_disentangled.py:195, 202, 217) — first tells the owning module's DensePairPlanCache
    which entries are being taken (note_index), so that the dense backward knows the support of the caller's loss without
    a declaration and without learning it from gradients whose zero pattern moves.  Every operation, indexing included, is
    then torch's own, and every result is a plain torch.Tensor."""
    _dl_cache = None

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        if func is torch.Tensor.__getitem__ and len(args) == 2 and isinstance(args[0], LinkPred):
            cache = args[0]._dl_cache
            # (only where a backward can follow: an evaluation loop under no_grad indexes without end and never consumes
            # what it reported)
            if cache is not None and args[0].dim() == 2 and args[0].requires_grad and torch.is_grad_enabled():
                cache.note_index(int(args[0].shape[0]), args[1])
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **(kwargs or {}))

    # Copies and serialised forms are DATA: plain tensors, without the owning module's cache (torch's default __deepcopy__
    # refuses a subclass without new_empty(); pickling the subclass would drag the cache along and break weights_only loads).
    def __deepcopy__(self, memo):
        return self.as_subclass(torch.Tensor).__deepcopy__(memo)

    def __reduce_ex__(self, proto):
        return self.as_subclass(torch.Tensor).__reduce_ex__(proto)


def as_link_pred(prob: torch.Tensor, cache: "DensePairPlanCache") -> torch.Tensor:
    if os.environ.get("DL_LINK_PRED_SUBCLASS", "1") == "0":        # plain tensor: the plan is learnt from the gradients alone
        return prob
    out = prob.as_subclass(LinkPred)
    out._dl_cache = cache
    return out


class ScoreAllPairs(torch.autograd.Function):
    """(Z, H) -> prob [N,N], the dense output the reference's caller indexes with masks
    (main_disentangled.py:195).  Backward: dl_score_allpairs_bwd on the pair plan of ``cache`` (the calling module's
    DensePairPlanCache: the support of the caller's loss masks)."""

    @staticmethod
    def forward(ctx, Z, H, t: float, cache: "DensePairPlanCache | None" = None):
        Z, H = _f32c(Z), _f32c(H)
        prob = score_allpairs_fwd(Z, H, t)
        ctx.t = t
        ctx.cache = cache if cache is not None else DensePairPlanCache()
        ctx.save_for_backward(Z, H, prob)
        return prob

    @staticmethod
    def backward(ctx, g_prob):
        Z, H, prob = ctx.saved_tensors
        pairs, valid = ctx.cache.lookup(g_prob)
        dZ, dH = score_allpairs_bwd(Z, H, pairs, ctx.t, prob, g_prob)
        if valid is not None:
            dZ, dH = dZ * valid, dH * valid
        return dZ, dH, None, None
