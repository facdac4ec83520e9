"""The reference's train / early-stop / eval loop (main_disentangled.py:191-224) on pair lists.

Same schedule as the reference: Adam(lr, weight_decay=5e-4) (:150, the flag is ignored there too),
one full-batch forward per epoch, validation AUC computed from THAT forward's probabilities (i.e.
the weights before the step, :202-204), best weights snapshotted after the step (:209), patience
200 (:212), test AUC with the best weights (:215-219).

`model` is any module with ``forward_pairs(x, graph, pairs) -> (emb, prob)``: the drop-in
``disenlink_amd.model.Disentangle`` on the GPU, or (in tests only) an oracle-backed module on CPU.
"""
from __future__ import annotations

import os
from copy import deepcopy
from dataclasses import dataclass, field

import numpy as np
import torch
from torch.optim import Adam

from .graph import Graph, PairList
from .metrics import AucPlan, auc_tie_avg, pair_bce_loss, pair_bce_loss_fused, pair_bce_weights
from .splits import LinkSplit


@dataclass
class PreparedRun:
    graph: Graph
    train_val_pairs: PairList       # [pos_train | neg_train | val], scored by one forward per epoch
    n_pos: int
    n_neg: int
    label_pos: torch.Tensor
    label_neg: torch.Tensor
    label_val: torch.Tensor
    test_pairs: PairList
    label_test: torch.Tensor
    m: int


def prepare_run(split: LinkSplit, device, seg_len: int = 32, row_bytes: int = 2048) -> PreparedRun:
    """``row_bytes`` = K * d * (4, or 2 with bf16 tables) of the model (sizes the XCD slicing of the pair plans)."""
    t = lambda a, dt=None: torch.as_tensor(a, device=device) if dt is None else torch.as_tensor(a, dtype=dt, device=device)
    graph = Graph.from_edge_rows(t(split.train_src), t(split.train_dst), split.n_nodes, seg_len=seg_len,
                                 row_bytes=row_bytes)
    pu = np.concatenate([split.pos_train.u, split.neg_train.u, split.val.u])
    pv = np.concatenate([split.pos_train.v, split.neg_train.v, split.val.v])
    tv = PairList.build(t(pu), t(pv), split.n_nodes, row_bytes=row_bytes)
    te = PairList.build(t(split.test.u), t(split.test.v), split.n_nodes, row_bytes=row_bytes)
    return PreparedRun(graph, tv, split.pos_train.u.size, split.neg_train.u.size,
                       t(split.pos_train.label, torch.float32), t(split.neg_train.label, torch.float32),
                       t(split.val.label, torch.float32), te, t(split.test.label, torch.float32), split.m)


@dataclass
class RunResult:
    test_auc: float
    best_val_auc: float
    epochs_run: int
    losses: list = field(default_factory=list)
    val_aucs: list = field(default_factory=list)


_FUSED_ADAM = os.environ.get("DL_FUSED_ADAM", "1") != "0"
_STACKED_ADAM = os.environ.get("DL_STACKED_ADAM", "1") != "0"
_ADAM_KERNEL = os.environ.get("DL_ADAM_KERNEL", "dl")        # "torch": torch._fused_adam_ over the stacked buffers


def _make_adam(model, on_gpu: bool, lr: float, weight_decay: float, capturable: bool = False):
    """The reference's Adam (main_disentangled.py:150).  On the GPU, for the drop-in module: the same fused update over
    the module's 4 shared parameter buffers instead of its 4K views (optim.StackedAdam), by default through the
    library's one-launch ``dl_adam_step``: torch.optim.Adam's ARITHMETIC, equal to torch's fused kernel up to rounding
    (fmaf weight decay, a different FMA contraction of the bias corrections: relative 1e-6 per step, tested) — NOT the
    same bits.  ``DL_ADAM_KERNEL=torch`` keeps ``torch._fused_adam_`` over the 4 buffers, which IS bit-identical to
    ``torch.optim.Adam(fused=True)`` over the views (parity runs against torch trajectories); ``DL_STACKED_ADAM=0``
    returns to torch's own optimiser."""
    if on_gpu and _FUSED_ADAM and _STACKED_ADAM and getattr(model, "_stacked_params", None) is not None \
            and model._stacked_params() is not None:
        from .optim import StackedAdam
        return StackedAdam(model, lr=lr, weight_decay=weight_decay, capturable=capturable,
                           use_torch_kernel=_ADAM_KERNEL == "torch")
    if capturable:
        return Adam(model.parameters(), lr=lr, weight_decay=weight_decay, capturable=True, fused=_FUSED_ADAM)
    return Adam(model.parameters(), lr=lr, weight_decay=weight_decay, fused=on_gpu and _FUSED_ADAM)


def _scores_and_loss(model, x, run, label_all, weight_all):
    """(prob over [pos | neg | validation], loss) of one training forward on the GPU path; the model decides whether
    the scorer runs forward + loss gradient + backward in one pass (model.forward_pairs_loss, DL_ONE_PASS_SCORER)."""
    if hasattr(model, "forward_pairs_loss"):
        _emb, prob, loss = model.forward_pairs_loss(x, run.graph, run.train_val_pairs, label_all, weight_all)
        return prob, loss
    _emb, prob = model.forward_pairs(x, run.graph, run.train_val_pairs)
    return prob, pair_bce_loss_fused(prob, label_all, weight_all)


def _loss_vectors(run, device):
    """Labels and weights over the WHOLE scored list [pos | neg | validation]: the validation pairs carry weight
    zero, so the fused loss runs on the scorer's output as it is — no slice, hence no zero-filled gradient
    buffer and copy in the backward."""
    n_val = run.label_val.numel()
    label = torch.cat([run.label_pos, run.label_neg, run.label_val.to(run.label_pos.dtype)])
    weight = torch.cat([pair_bce_weights(run.n_pos, run.n_neg, run.m, device),
                        torch.zeros(n_val, dtype=torch.float32, device=device)])
    return label, weight


def _graphed_epoch(model, x, run, lr, weight_decay, b, label_all, weight_all, es_factory=None):
    """Capture one full epoch (forward, fused loss, backward, Adam step, validation AUC) into a HIP graph:
    every launch of the epoch — ours and torch's — is replayed with one host call, which removes the
    launch-bound host time of small graphs.  Returns (replay, out) with out = [loss, auc] on the device."""
    opt = _make_adam(model, True, lr, weight_decay, capturable=True)
    out = torch.zeros(2, dtype=torch.float64, device=x.device)
    val_plan = AucPlan(run.label_val)
    seed = torch.ones((), dtype=torch.float32, device=x.device)
    es = es_factory(val_plan) if es_factory is not None else None   # early_stop.DeviceEarlyStop or None

    def epoch():
        prob, loss = _scores_and_loss(model, x, run, label_all, weight_all)
        # gradients are (re)created by the backward inside the capture — they come from the graph's own memory pool, so
        # replays find them at the same addresses (torch's whole-network capture recipe).  Keeping pre-allocated
        # gradients instead (zero_grad(set_to_none=False)) cost 4K fills and 4K accumulating adds per epoch.
        opt.zero_grad(set_to_none=True)
        loss.backward(seed)
        opt.step()
        if es is not None:
            es.finish(loss, prob[b:])                               # device-side bookkeeping (early_stop.py): no `out`
        else:
            out[0] = loss.detach().double()
            out[1] = val_plan.auc(prob[b:])

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                                   # warm-up on a side stream, as capture requires
        snap = model.snapshot_state() if hasattr(model, "snapshot_state") else deepcopy(model.state_dict())
        for _ in range(2):
            epoch()
        model.load_state_dict(snap)                                 # warm-up must not train
        # Adam's moments and step counters were created by the warm-up steps and must stay the SAME tensors for
        # the capture (state created inside the capture would be re-initialised by every replay): reset in place
        for st in opt.state.values():
            for v in st.values():
                if torch.is_tensor(v):
                    v.zero_()
    torch.cuda.current_stream().wait_stream(side)
    # (several epochs per graph, or two graph instances replayed in turn, do not make the replayed loop faster:
    # profiles/r5z_graph_epochs_execs_sweep.txt — its 25-40 us per epoch over the eager loop are per-node dispatch cost)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        epoch()
    model.load_state_dict(snap)                                     # capture ran the epoch once: undo it
    for st in opt.state.values():
        for v in st.values():
            if torch.is_tensor(v):
                v.zero_()
    if es is not None:
        es.reset()                                                  # ... and the warm-up / capture epochs' bookkeeping
    # A captured graph replays raw addresses: every tensor its kernels touch that was NOT allocated during the capture
    # must outlive the graph.  The optimiser (moments, step counters) and the AUC index sets are created here, so they
    # are handed back for the caller to hold for as long as it replays (dropping them frees memory the graph still
    # reads and writes: the next allocation of that size would be corrupted, or an index set would turn to garbage).
    # ... and so is the zero-padded copy of x the projection kernels read when F % 4 != 0 (ops.padded_features): it is
    # cached per source tensor, and held here as well so that nothing but the end of this run can free it
    # ... and the persistent bf16 planes of x and x^T the projection reads (ops._XPlanes: built by the second warm-up epoch)
    from .ops import padded_features, xplanes_for
    return graph.replay, (es if es is not None else out), (graph, opt, val_plan, epoch, label_all, weight_all, x,
                                                           padded_features(x), xplanes_for(x), seed, es)


def run_link_prediction(model, x: torch.Tensor, run: PreparedRun, epochs: int = 2000, lr: float = 1e-4,
                        patience: int = 200, weight_decay: float = 5e-4, log=None, use_graph: bool = False) -> RunResult:
    """``use_graph=True`` (GPU only): replay each epoch from a captured HIP graph (same arithmetic, same
    schedule; only the host-side launch cost disappears)."""
    if use_graph and x.is_cuda:
        return _run_graphed(model, x, run, epochs, lr, patience, weight_decay, log)
    # same update rule as the reference's Adam (main_disentangled.py:150); on the GPU torch's single-kernel
    # ("fused") implementation of it instead of one multi-tensor launch per elementwise step
    opt = _make_adam(model, bool(x.is_cuda), lr, weight_decay)
    snapshot = getattr(model, "snapshot_state", None) or (lambda: deepcopy(model.state_dict()))
    best_auc, stale, weights = 0.0, 0, snapshot()
    res = RunResult(float("nan"), 0.0, 0)
    a, b = run.n_pos, run.n_pos + run.n_neg
    for lab in (run.label_val, run.label_test):                     # validated once: the per-epoch AUC never syncs
        if not 0 < float(lab.sum()) < lab.numel():
            raise ValueError("AUC undefined with one class")
    val_plan = AucPlan(run.label_val)
    fused = x.is_cuda                                               # fused loss+gradient kernel on the GPU path
    if fused:
        label_all, weight_all = _loss_vectors(run, x.device)
    from .early_stop import DeviceEarlyStop, drive
    if fused and DeviceEarlyStop.usable(model, x, val_plan):
        # early stopping and the best-weights snapshot on the device, the history read one epoch behind the launches:
        # the GPU never waits for the host between epochs (early_stop.py)
        es = DeviceEarlyStop(model, val_plan, epochs, patience)
        seed = torch.ones((), dtype=torch.float32, device=x.device)  # d loss / d loss, made once instead of a fill per epoch

        def launch_epoch():
            model.train()
            prob, loss = _scores_and_loss(model, x, run, label_all, weight_all)
            opt.zero_grad()
            loss.backward(seed)
            opt.step()
            model.eval()
            es.finish(loss, prob[b:])                               # AUC from the pre-step forward (:202-204), weights after the step (:209)

        res.best_val_auc = drive(es, epochs, patience, launch_epoch, res, log)
        es.restore()
        with torch.no_grad():
            _emb, prob = model.forward_pairs(x, run.graph, run.test_pairs)
        res.test_auc = float(auc_tie_avg(run.label_test, prob, check=False))
        return res
    for epoch in range(epochs):
        model.train()
        if fused:
            prob, loss = _scores_and_loss(model, x, run, label_all, weight_all)
        else:
            _emb, prob = model.forward_pairs(x, run.graph, run.train_val_pairs)
            loss = pair_bce_loss(prob[:a], run.label_pos, prob[a:b], run.label_neg, run.m)
        opt.zero_grad()
        loss.backward()
        opt.step()
        model.eval()
        auc_t = val_plan.auc(prob[b:])                               # from the pre-step forward, like :202-204
        loss_v, auc = torch.stack([loss.detach().double(), auc_t]).tolist()     # ONE device->host sync per epoch
        res.losses.append(loss_v)
        res.val_aucs.append(auc)
        res.epochs_run = epoch + 1
        if auc > best_auc:
            stale, best_auc = 0, auc
            weights = snapshot()                                    # state AFTER the step, like :209
        else:
            stale += 1
        if stale > patience:
            break
        if log is not None:
            log(f"epoch: {epoch} loss: {res.losses[-1]} val_auc: {best_auc}")
    model.load_state_dict(weights)
    with torch.no_grad():
        _emb, prob = model.forward_pairs(x, run.graph, run.test_pairs)
    res.test_auc = float(auc_tie_avg(run.label_test, prob, check=False))
    res.best_val_auc = best_auc
    return res


def _run_graphed(model, x, run, epochs, lr, patience, weight_decay, log) -> RunResult:
    for lab in (run.label_val, run.label_test):
        if not 0 < float(lab.sum()) < lab.numel():
            raise ValueError("AUC undefined with one class")
    b = run.n_pos + run.n_neg
    label_all, weight_all = _loss_vectors(run, x.device)
    from .early_stop import DeviceEarlyStop, drive
    factory = lambda plan: DeviceEarlyStop(model, plan, epochs, patience) if DeviceEarlyStop.usable(model, x, plan) else None
    replay, out, keep_alive = _graphed_epoch(model, x, run, lr, weight_decay, b, label_all, weight_all, factory)
    res = RunResult(float("nan"), 0.0, 0)
    if isinstance(out, DeviceEarlyStop):
        es = out
        res.best_val_auc = drive(es, epochs, patience, replay, res, log)
        torch.cuda.synchronize()
        del replay, keep_alive                                      # only now may the graph's external tensors go
        es.restore()
        with torch.no_grad():
            _emb, prob = model.forward_pairs(x, run.graph, run.test_pairs)
        res.test_auc = float(auc_tie_avg(run.label_test, prob, check=False))
        return res
    snapshot = getattr(model, "snapshot_state", None) or (lambda: deepcopy(model.state_dict()))
    best_auc, stale, weights = 0.0, 0, snapshot()
    for epoch in range(epochs):
        replay()
        loss_v, auc = out.tolist()                                  # the one sync of the epoch
        res.losses.append(loss_v)
        res.val_aucs.append(auc)
        res.epochs_run = epoch + 1
        if auc > best_auc:
            stale, best_auc = 0, auc
            weights = snapshot()
        else:
            stale += 1
        if stale > patience:
            break
        if log is not None:
            log(f"epoch: {epoch} loss: {loss_v} val_auc: {best_auc}")
    torch.cuda.synchronize()
    del replay, keep_alive                                          # only now may the graph's external tensors go
    model.load_state_dict(weights)
    with torch.no_grad():
        _emb, prob = model.forward_pairs(x, run.graph, run.test_pairs)
    res.test_auc = float(auc_tie_avg(run.label_test, prob, check=False))
    res.best_val_auc = best_auc
    return res


# --------------------------------------------------------------------------- the same loop, row-sharded
@dataclass
class ShardedRun:
    shard: object                   # dist.Shard over [pos_train | neg_train | val] sorted by (u, v)
    test_shard: object              # the same partition and local graph with the test pairs
    label_all: torch.Tensor         # [P_total] labels of the sorted list (replicated)
    weight_all: torch.Tensor        # [P_total] loss weights: 1/n_pos, 1/(m n_neg), 0 on the validation pairs
    val_local: torch.Tensor         # positions (in the LOCAL slice) of this rank's validation pairs
    val_plan: object                # metrics.ShardedAucPlan over them
    test_plan: object


def prepare_run_sharded(split: LinkSplit, rank: int, world: int, device, row_bytes: int = 2048, group=None,
                        n_chunks: int = 1) -> ShardedRun:
    """Every rank holds the (host) split; the scored list [pos_train | neg_train | val] is sorted by (u, v) — pairs are
    scored by the owner of u — and its labels / loss weights are permuted along (validation pairs: weight 0, so one
    fused loss runs over the whole list, as in the single-GPU loop)."""
    from . import dist as dd
    from .metrics import ShardedAucPlan
    pu = np.concatenate([split.pos_train.u, split.neg_train.u, split.val.u])
    pv = np.concatenate([split.pos_train.v, split.neg_train.v, split.val.v])
    n_pos, n_neg, n_val = split.pos_train.u.size, split.neg_train.u.size, split.val.u.size
    label = np.concatenate([split.pos_train.label, split.neg_train.label, split.val.label]).astype(np.float32)
    weight = np.concatenate([np.full(n_pos, 1.0 / max(n_pos, 1), np.float32),
                             np.full(n_neg, 1.0 / (split.m * max(n_neg, 1)), np.float32), np.zeros(n_val, np.float32)])
    is_val = np.concatenate([np.zeros(n_pos + n_neg, bool), np.ones(n_val, bool)])
    order = np.lexsort((pv, pu))
    pu, pv, label, weight, is_val = pu[order], pv[order], label[order], weight[order], is_val[order]
    shard = dd.Shard.build(rank, world, split.n_nodes, split.train_src, split.train_dst, pu, pv, device,
                           row_bytes=row_bytes, n_chunks=n_chunks)
    q0, q1 = shard.pair_lo, shard.pair_hi
    t = lambda a_: torch.as_tensor(a_, device=device)
    val_local = t(np.flatnonzero(is_val[q0:q1]))
    val_plan = ShardedAucPlan(t(label[q0:q1][is_val[q0:q1]]), group)
    to = np.lexsort((split.test.v, split.test.u))
    test_shard = shard.with_pairs(split.test.u[to], split.test.v[to])
    test_plan = ShardedAucPlan(t(split.test.label[to][test_shard.pair_lo:test_shard.pair_hi].astype(np.float32)), group)
    for plan, what in ((val_plan, "validation"), (test_plan, "test")):
        if plan.n_pos == 0 or plan.n_neg == 0:
            raise ValueError(f"AUC undefined with one class ({what} pairs)")
    return ShardedRun(shard, test_shard, t(label), t(weight), val_local, val_plan, test_plan)


def run_link_prediction_sharded(model, x_local: torch.Tensor, run: ShardedRun, epochs: int = 2000, lr: float = 1e-4,
                                patience: int = 200, weight_decay: float = 5e-4, log=None, backend=None,
                                group=None) -> RunResult:
    """main_disentangled.py:191-224 with the rows of the graph, of the features and of the pair list sharded over the
    ranks of `group` (one process per GPU): `x_local` = the feature rows of ``run.shard.local_real_rows()``; `model` = a
    replica with IDENTICAL initial weights on every rank.  Per epoch: dist.sharded_forward_loss (all-gathers of Z, s, H;
    the one-pass scorer over the rank's incidence rows where the backend has it), backward, ONE all-reduce of the weight
    gradients (dist.allreduce_gradients), the same Adam step on every replica; the loss value and the tie-aware
    validation AUC (metrics.ShardedAucPlan: from that forward's probabilities, i.e. the weights before the step) cross
    the ranks in one small all-reduce — every rank sees the same numbers, so early stopping and the best-weights
    snapshot stay in lock-step without a broadcast.  Test AUC with the best weights, as :215-219."""
    import torch.distributed as dist
    from . import dist as dd
    on_gpu = bool(x_local.is_cuda)
    opt = _make_adam(model, on_gpu, lr, weight_decay)
    snapshot = getattr(model, "snapshot_state", None) or (lambda: deepcopy(model.state_dict()))
    best_auc, stale, weights = 0.0, 0, snapshot()
    res = RunResult(float("nan"), 0.0, 0)
    cpu_wire = on_gpu and dist.get_backend(group) == "gloo"
    for epoch in range(epochs):
        model.train()
        _emb, prob, loss = dd.sharded_forward_loss(model, x_local, run.shard, run.label_all, run.weight_all,
                                                   backend=backend, group=group)
        opt.zero_grad()
        loss.backward()
        dd.allreduce_gradients(model, group)
        opt.step()
        model.eval()
        both = torch.stack([loss.detach().double(), run.val_plan.partial(prob.detach().index_select(0, run.val_local))])
        if cpu_wire:
            both = both.cpu()
        dist.all_reduce(both, group=group)                          # global loss, global 2U: one message
        loss_v, u2 = both.tolist()                                  # the one device->host sync of the epoch
        auc = run.val_plan.auc_from_sum(u2)
        res.losses.append(loss_v)
        res.val_aucs.append(auc)
        res.epochs_run = epoch + 1
        if auc > best_auc:
            stale, best_auc = 0, auc
            weights = snapshot()                                    # state AFTER the step, like :209
        else:
            stale += 1
        if stale > patience:
            break
        if log is not None:
            log(f"epoch: {epoch} loss: {res.losses[-1]} val_auc: {best_auc}")
    model.load_state_dict(weights)
    with torch.no_grad():
        _emb, prob = dd.sharded_forward(model, x_local, run.test_shard, backend=backend, group=group)
    res.test_auc = run.test_plan.auc(prob)
    res.best_val_auc = best_auc
    return res
