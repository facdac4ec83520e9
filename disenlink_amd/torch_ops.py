"""The C ABI of libdisenlink_hip.so as REGISTERED torch operators (``torch.ops.disenlink.*``): schemas, fake
(shape-only) implementations and autograd formulas through ``torch.library``, on top of the same raw wrappers of
``ops.py`` — so the dispatcher sees the hot path, ``torch.compile`` traces through it (no graph breaks at ctypes calls)
and profilers name the operators.

    import disenlink_amd.torch_ops                      # registers the operators
    g = torch_ops.register_graph(graph)                 # Graph / PairList -> integer handles (operators take tensors,
    q = torch_ops.register_pairs(pairs)                 #                     numbers and handles, not Python objects)
    H, p, a, s = torch.ops.disenlink.route_aggregate(Z, g, beta, t)
    prob = torch.ops.disenlink.score_pairs(Z, H, q, t)

``Disentangle(..., use_torch_ops=True)`` routes ``forward_pairs`` through these operators (same kernels, same bits).
The plans stay Python objects held in a registry: a handle is a plain int, constant under ``torch.compile`` (a new
graph / pair list is a new handle and retraces).  There is no CPU implementation: the operators run on the GPU only.

Reference lines each operator stands for: include/disenlink_hip.h (route: model.py:56-72, aggregate: model.py:73-75,
scorer: model.py:109-113, backward: autograd under main_disentangled.py:198).
"""
from __future__ import annotations

import torch
from torch import Tensor

from . import _lib, ops
from .graph import Graph, PairList, by_handle

# Handles live in disenlink_amd/graph.py: assigned when a Graph / PairList is BUILT and held weakly there — an operator
# call holds its objects through the caller's own references (the module's forward_pairs has them as arguments), so a
# caller that builds a new Graph or PairList per epoch (resampled negatives, sweeps) accumulates nothing here.
_pinned: dict[int, object] = {}            # register_*(…, pin=True): kept alive until release(handle)


def register_graph(g: Graph, pin: bool = False) -> int:
    """-> handle of `g` for the operators below (a plain attribute read: torch.compile traces it).  Nothing keeps the
    graph alive unless ``pin=True`` (then until ``release(handle)``): hold your own reference while operators use it."""
    if pin:
        _pinned[g._dl_handle] = g
    return g._dl_handle


def register_pairs(p: PairList, pin: bool = False) -> int:
    if pin:
        _pinned[p._dl_handle] = p
    return p._dl_handle


def release(handle: int) -> None:
    """Drop the pin of a handle, if any.  Handles vanish by themselves with their object."""
    _pinned.pop(handle, None)


def _g(h: int) -> Graph:
    g = by_handle(h, Graph)
    if g is None:
        raise ValueError(f"no graph registered under handle {h} (torch_ops.register_graph; the object must still be alive)")
    return g


def _p(h: int) -> PairList:
    p = by_handle(h, PairList)
    if p is None:
        raise ValueError(f"no pair list registered under handle {h} (torch_ops.register_pairs; the object must still be alive)")
    return p


# ---------------------------------------------------------------------------------------------- routing + aggregation
@torch.library.custom_op("disenlink::route_fwd", mutates_args=(), device_types="cuda")
def route_fwd(Z: Tensor, graph: int, t: float) -> tuple[Tensor, Tensor, Tensor]:
    """(p uint8[E], a f32[E], s f32[N,K]) — dl_route_fwd"""
    return ops.route_fwd(_g(graph), Z, t)


@route_fwd.register_fake
def _(Z, graph, t):
    g = _g(graph)
    return (Z.new_empty(g.n_edges, dtype=torch.uint8), Z.new_empty(g.n_edges, dtype=torch.float32),
            Z.new_empty((Z.shape[0], Z.shape[1]), dtype=torch.float32))


@torch.library.custom_op("disenlink::aggregate_fwd", mutates_args=(), device_types="cuda")
def aggregate_fwd(Z: Tensor, p: Tensor, a: Tensor, s: Tensor, graph: int, beta: float) -> Tensor:
    """H [N,K,d] — dl_aggregate_fwd"""
    return ops.aggregate_fwd(_g(graph), Z, beta, p, a, s)


@aggregate_fwd.register_fake
def _(Z, p, a, s, graph, beta):
    return torch.empty_like(Z)


@torch.library.custom_op("disenlink::route_aggregate_bwd", mutates_args=(), device_types="cuda")
def route_aggregate_bwd(Z: Tensor, p: Tensor, a: Tensor, s: Tensor, dH: Tensor, graph: int, beta: float,
                        t: float) -> Tensor:
    """dZ [N,K,d] — dl_route_aggregate_bwd"""
    return ops.route_aggregate_bwd(_g(graph), Z, beta, t, p, a, s, dH.contiguous())


@route_aggregate_bwd.register_fake
def _(Z, p, a, s, dH, graph, beta, t):
    return Z.new_empty(Z.shape, dtype=torch.float32)


@torch.library.custom_op("disenlink::route_aggregate", mutates_args=(), device_types="cuda")
def route_aggregate(Z: Tensor, graph: int, beta: float, t: float) -> tuple[Tensor, Tensor, Tensor, Tensor]:
    """(H, p, a, s): Disentangle_layer.forward (model.py:55-77) on the CSR of adj; differentiable in Z through H."""
    g = _g(graph)
    Zc = Z.contiguous()
    p, a, s = ops.route_fwd(g, Zc, t)
    return ops.aggregate_fwd(g, Zc, beta, p, a, s), p, a, s


@route_aggregate.register_fake
def _(Z, graph, beta, t):
    g = _g(graph)
    return (torch.empty_like(Z), Z.new_empty(g.n_edges, dtype=torch.uint8), Z.new_empty(g.n_edges, dtype=torch.float32),
            Z.new_empty((Z.shape[0], Z.shape[1]), dtype=torch.float32))


def _ra_setup(ctx, inputs, output):
    Z, graph, beta, t = inputs
    _H, p, a, s = output
    ctx.save_for_backward(Z, p, a, s)
    ctx.graph, ctx.beta, ctx.t = graph, beta, t


def _ra_backward(ctx, dH, _dp, _da, _ds):
    Z, p, a, s = ctx.saved_tensors
    return torch.ops.disenlink.route_aggregate_bwd(Z, p, a, s, dH, ctx.graph, ctx.beta, ctx.t), None, None, None


route_aggregate.register_autograd(_ra_backward, setup_context=_ra_setup)


# ---------------------------------------------------------------------------------------------- pair scorer
@torch.library.custom_op("disenlink::score_pairs_terms", mutates_args=(), device_types="cuda")
def score_pairs_terms(Z: Tensor, H: Tensor, pairs: int, t: float) -> tuple[Tensor, Tensor]:
    """(prob [P], coef [2,P,K]) — dl_score_pairs_fwd with the per-factor terms the backward reuses (tuned shapes; an
    empty coef for the generic kernels)."""
    pl = _p(pairs)
    prob, coef = ops.score_pairs_fwd(Z.contiguous(), H.contiguous(), pl.pu, pl.pv, t, pl, want_coef=True)
    return prob, (coef if coef is not None else Z.new_empty((0,), dtype=torch.float32))


@score_pairs_terms.register_fake
def _(Z, H, pairs, t):
    P = _p(pairs).n_pairs
    dt = _lib.DL_BF16 if Z.dtype == torch.bfloat16 else _lib.DL_F32   # the same rule as the real operator (ops.score_pairs_fwd)
    terms = ops.score_terms_available(Z.shape[1], Z.shape[2], dt)
    return Z.new_empty(P, dtype=torch.float32), Z.new_empty((2, P, Z.shape[1]) if terms else (0,), dtype=torch.float32)


@torch.library.custom_op("disenlink::score_pairs_bwd", mutates_args=(), device_types="cuda")
def score_pairs_bwd(Z: Tensor, H: Tensor, prob: Tensor, g_prob: Tensor, coef: Tensor, pairs: int,
                    t: float) -> tuple[Tensor, Tensor]:
    """(dZ, dH) — dl_score_pairs_bwd"""
    return ops.score_pairs_bwd(Z, H, _p(pairs), t, prob, g_prob.contiguous(), coef=coef if coef.numel() else None)


@score_pairs_bwd.register_fake
def _(Z, H, prob, g_prob, coef, pairs, t):
    return Z.new_empty(Z.shape, dtype=torch.float32), Z.new_empty(Z.shape, dtype=torch.float32)


def _sp_setup(ctx, inputs, output):
    Z, H, pairs, t = inputs
    prob, coef = output
    ctx.save_for_backward(Z, H, prob, coef)
    ctx.pairs, ctx.t = pairs, t


def _sp_backward(ctx, g_prob, _g_coef):
    Z, H, prob, coef = ctx.saved_tensors
    dZ, dH = torch.ops.disenlink.score_pairs_bwd(Z, H, prob, g_prob, coef, ctx.pairs, ctx.t)
    return dZ, dH, None, None


score_pairs_terms.register_autograd(_sp_backward, setup_context=_sp_setup)


def score_pairs(Z: Tensor, H: Tensor, pairs: int, t: float) -> Tensor:
    """prob [P] = sigmoid(sum_k (h_k[u].h_k[v]) exp(z_k[u].z_k[v] / t)) at the registered pairs; differentiable."""
    return torch.ops.disenlink.score_pairs_terms(Z, H, pairs, t)[0]


# ---------------------------------------------------------------------------------------------- projection
@torch.library.custom_op("disenlink::project_fwd", mutates_args=(), device_types="cuda")
def project_fwd(x: Tensor, W1: Tensor, b1: Tensor, W2: Tensor, b2: Tensor) -> Tensor:
    """Z [N,K,d] of the two-layer factor MLPs on the matrix cores — dl_project_fwd (W1 [K,nhid,F], W2 [K,d,nhid])"""
    return ops.project_fwd(x, W1, b1, W2, b2)


@project_fwd.register_fake
def _(x, W1, b1, W2, b2):
    return x.new_empty((x.shape[0], W1.shape[0], W2.shape[1]), dtype=torch.float32)


@torch.library.custom_op("disenlink::project_bwd", mutates_args=(), device_types="cuda")
def project_bwd(x: Tensor, W1: Tensor, b1: Tensor, W2: Tensor, dZ: Tensor) -> tuple[Tensor, Tensor, Tensor, Tensor]:
    """(dW1, db1, dW2, db2) — dl_project_bwd (the hidden layer is recomputed)"""
    return ops.project_bwd(x, W1, b1, W2, dZ.contiguous(), one_allocation=False)


@project_bwd.register_fake
def _(x, W1, b1, W2, dZ):
    return (torch.empty_like(W1), torch.empty_like(b1), torch.empty_like(W2),
            x.new_empty((W2.shape[0], W2.shape[1]), dtype=torch.float32))


def _pj_setup(ctx, inputs, output):
    x, W1, b1, W2, _b2 = inputs
    ctx.save_for_backward(x, W1, b1, W2)


def _pj_backward(ctx, dZ):
    x, W1, b1, W2 = ctx.saved_tensors
    dW1, db1, dW2, db2 = torch.ops.disenlink.project_bwd(x, W1, b1, W2, dZ)
    return None, dW1, db1, dW2, db2


project_fwd.register_autograd(_pj_backward, setup_context=_pj_setup)


def forward_pairs(model, x: Tensor, graph: int, pairs: int):
    """(emb [N,K*d], prob [P]) of a two-layer ``Disentangle`` through the registered operators only — the function
    ``torch.compile`` traces end to end (``Disentangle(use_torch_ops=True).forward_pairs`` calls it)."""
    fs = model.factors                                         # the K per-factor Parameters are the autograd leaves
    Z = torch.ops.disenlink.project_fwd(x, torch.stack([f.mlp1.weight for f in fs]), torch.stack([f.mlp1.bias for f in fs]),
                                        torch.stack([f.mlp2.weight for f in fs]), torch.stack([f.mlp2.bias for f in fs]))
    H = torch.ops.disenlink.route_aggregate(Z, graph, float(model.beta), float(model.temperature))[0]
    prob = score_pairs(Z, H, pairs, float(model.temperature))
    return H.reshape(H.shape[0], -1), prob
