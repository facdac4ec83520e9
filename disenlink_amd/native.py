"""The COMPILED torch binding (libdisenlink_torch.so, csrc/torch/dl_torch.cpp): the training step's hot path as one C++
autograd node registered with TORCH_LIBRARY over the same C ABI the ctypes binding uses — ``torch.ops.disenlink_native.*``.

    from disenlink_amd import native
    if native.available():
        H, prob, loss = native.hot_path_pairs_loss(Z, graph, pairs, beta, t, label, weight)

Same kernels and bits as ``ops.HotPathPairsLoss`` (route + aggregate + one-pass scorer + loss value; backward: routing /
aggregation with d/dloss applied inside the last kernel); what it removes is the Python between the launches (ctypes
marshalling, autograd.Function frames): host time of the eager loop on small graphs.  Round 5: the projection over the
module's shared buffers (``project_stacked``: forward + the four stacked gradients), the Adam step (``adam_step``) and
the AUC counts (``auc_pair_counts``) are compiled operators too, and the hot path takes bf16 tables — an eager epoch of
the pair-list training loop then has no ctypes call and no Python ``autograd.Function`` left.  ``Disentangle.forward_pairs_loss`` uses it when it is built and the shape has a
tuned kernel (``DL_NATIVE_OPS=0`` keeps the Python operators).  There is no fallback inside: a missing library means
``available()`` is False and the ctypes path runs — which itself has no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib, ops
from .graph import Graph, PairList

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdisenlink_torch.so")
_state = {"loaded": None}
BINDING_ABI = 6                  # csrc/torch/dl_torch.cpp: DL_TORCH_BINDING_ABI
_OPS = ("hot_path_pairs_loss", "project_stacked", "adam_step", "auc_pair_counts", "epoch_finish", "binding_abi")


def load() -> bool:
    """Load libdisenlink_torch.so once (after libdisenlink_hip.so, whose symbols it binds) -> whether it is there."""
    if _state["loaded"] is None:
        ok = False
        if os.path.exists(LIB_PATH) and os.environ.get("DL_NATIVE_OPS", "1") != "0":
            _lib.load()
            try:
                torch.ops.load_library(LIB_PATH)
                ns = getattr(torch.ops, "disenlink_native", None)
                missing = [o for o in _OPS if ns is None or not hasattr(ns, o)]
                abi = None if missing else int(ns.binding_abi())
                ok = not missing and abi == BINDING_ABI
                if not ok:                         # a library built before the operator set / a schema changed
                    import warnings
                    warnings.warn(f"libdisenlink_torch.so is stale (missing operators {missing}, binding ABI {abi}, expected "
                                  f"{BINDING_ABI}); rebuild with python -m disenlink_amd.build — using the ctypes binding of the same C ABI")
            except OSError as e:                   # built against another torch: the ctypes binding (the same HIP kernels) carries on
                import warnings
                warnings.warn(f"libdisenlink_torch.so does not load ({e}); rebuild with python -m disenlink_amd.build — "
                              "using the ctypes binding of the same C ABI")
        _state["loaded"] = ok
    return _state["loaded"]


def available() -> bool:
    return load()


def hot_path_pairs_loss(Z: torch.Tensor, graph: Graph, pairs: PairList, beta: float, t: float, label, weight,
                        table_dtype=torch.float32):
    """(H [N,K,d] fp32, prob [P], loss) — see ops.HotPathPairsLoss; Z fp32 on the GPU, graph unsharded; table_dtype =
    torch.bfloat16 stores the gathered Z / H tables as bf16 (arithmetic and gradients stay fp32)."""
    if not load():
        raise _lib.DisenlinkHipError("libdisenlink_torch.so is not built (python -m disenlink_amd.build)")
    N, K, d = Z.shape
    P = int(label.numel())
    if P != pairs.n_pairs:
        raise ValueError("label / weight must cover the pair list")
    graph.c_struct()
    pairs.bind_labels(label, weight, P)                             # per-entry labels once the same tensors come back (graph.py)
    pairs.c_struct(P)
    ws_g = ops._workspace(graph.c_plan(), Z.device, K, d)
    ws_p = ops._workspace(pairs.c_plan(), Z.device, K, d)
    ws_b = ops._ws_bce(Z.device)
    H, prob, loss = torch.ops.disenlink_native.hot_path_pairs_loss(Z, C.addressof(graph._struct), C.addressof(pairs._struct),
                                                                   graph.n_edges, float(beta), float(t), label, weight, ws_g, ws_p, ws_b,
                                                                   1 if table_dtype == torch.bfloat16 else 0)
    # the node's backward dereferences the two structs (and the device arrays they point to): they live exactly as long as
    # the node does, whatever the caller does with its Graph / PairList in between (a per-epoch resampled pair list)
    if loss.grad_fn is not None:
        loss.grad_fn.metadata["disenlink_keepalive"] = (graph, pairs)
    return H, prob, loss


def project_ok(x: torch.Tensor, d: int, single_layer: bool) -> bool:
    """Does the compiled projection node serve this call?  (two-layer form, fp32 rows of 16-byte-aligned length, a factor
    width the kernels are instantiated for; the Python operator pads / re-stacks everything else)"""
    return load() and not single_layer and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[1] % 4 == 0 \
        and d in (32, 64, 128) and os.environ.get("DL_PROJECT_BWD", "native") != "library"


def project_stacked(x: torch.Tensor, bufs, params) -> torch.Tensor:
    """Z [N,K,d] = the K factor MLPs of x over the module's shared buffers (ops.ProjectStacked as a C++ autograd node):
    bufs = (W1 [K,nhid,F], b1, W2 [K,d,nhid], b2), params = the 4 K per-factor Parameters that view them."""
    W1, b1, W2, b2 = bufs
    need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
    keep = need_grad and ops.keep_hidden(x.shape[0], x.shape[1], W1.shape[0], W1.shape[1])
    return torch.ops.disenlink_native.project_stacked(x, W1, b1, W2, b2, params, keep, ops.xplanes_for(x))


def adam_step(bufs, params, exp_avg, exp_avg_sq, state, lr, beta1, beta2, eps, weight_decay, host_step: int = 0) -> None:
    """optim.StackedAdam's update (dl_adam_step; dl_adam_step_at when the caller counts the steps: host_step >= 1) with the
    gradient bookkeeping in C++; params = the parameters of every buffer, buffer by buffer (K each)."""
    torch.ops.disenlink_native.adam_step(bufs, params, exp_avg, exp_avg_sq, state, float(lr), float(beta1), float(beta2),
                                         float(eps), float(weight_decay), int(host_step))


def auc_pair_counts(score: torch.Tensor, pos_idx: torch.Tensor, neg_idx: torch.Tensor) -> torch.Tensor:
    """int64[1]: sum over positives of (2 #{negatives below} + #{negatives equal}) — dl_auc_pair_counts."""
    return torch.ops.disenlink_native.auc_pair_counts(score, pos_idx, neg_idx)


def epoch_finish(score_val, pos_idx, neg_idx, u2, loss, params, best, state, hist, ring_ptr: int, ring: int, denom2: float,
                 max_epochs: int, patience: int) -> None:
    """early_stop.DeviceEarlyStop.finish in one C++ call: dl_auc_pair_counts_add + dl_epoch_finish."""
    torch.ops.disenlink_native.epoch_finish(score_val, pos_idx, neg_idx, u2, loss, params, best, state, hist, int(ring_ptr),
                                            int(ring), float(denom2), int(max_epochs), int(patience))
