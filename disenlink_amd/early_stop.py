"""Early stopping and the best-weights snapshot of the training loop, kept ON THE DEVICE.

The reference ends every epoch on the host (main_disentangled.py:199-214): it reads the loss and the validation AUC back,
compares the AUC with the best so far, deep-copies ``model.state_dict()`` when it improved and counts the epochs since
(patience 200).  Read back every epoch, the GPU idles from the copy until the host has launched the next epoch's first
kernel — 74 us of a 940 us epoch on the squirrel-shaped graph (``profiles/r5z_epoch_sequence.txt``), a quarter of a
chameleon epoch — and torch's scalar glue (cast, divide, stack, copy) is five more launches.

``DeviceEarlyStop`` keeps that bookkeeping in device memory (``dl_epoch_finish``, one launch per epoch: the AUC from the
integer counts, the comparison, the conditional copy of the module's 4 shared parameter buffers, the patience counter, a
``stopped`` flag) and the host reads the (loss, AUC) history ONE EPOCH BEHIND, after it has queued the next epoch: the
values, the decisions and the kept weights are the ones the reference's loop produces — the host mirrors the same
comparisons on the same doubles to know when to stop, and the device ignores whatever was queued past that point.
"""
from __future__ import annotations

import ctypes as C

import torch


class DeviceEarlyStop:
    LAG = 1                                                         # launches the host's read-back trails
    EVENTS = 4

    def __init__(self, model, val_plan, max_epochs: int, patience: int, epochs_per_launch: int = 1):
        self.per = max(1, int(epochs_per_launch))                   # a replayed graph may hold several epochs
        self.RING = 4 * self.per
        from . import _lib
        self.lib = _lib.load()
        self.check = _lib.check
        self.model, self.plan = model, val_plan
        self.bufs = list(model._stacked.values())                   # the 4 shared [K, ...] parameter buffers
        dev = self.bufs[0].device
        if not all(b.is_cuda and b.dtype == torch.float32 and b.is_contiguous() for b in self.bufs):
            raise RuntimeError("DeviceEarlyStop: contiguous fp32 CUDA parameter buffers expected")
        self.best = [b.detach().clone() for b in self.bufs]         # `weights = deepcopy(state_dict)` before the loop
        n = len(self.bufs)
        self._p = (C.c_void_p * n)(*[b.data_ptr() for b in self.bufs])
        self._b = (C.c_void_p * n)(*[b.data_ptr() for b in self.best])
        self._numel = (C.c_size_t * n)(*[b.numel() for b in self.bufs])
        self.max_epochs, self.patience = int(max_epochs), int(patience)
        nbytes = int(self.lib.dl_epoch_state_bytes())
        self.state = torch.zeros((nbytes + 7) // 8, dtype=torch.int64, device=dev)
        self.hist = torch.zeros(max(self.max_epochs, 1), 2, dtype=torch.float64, device=dev)
        self.u2 = torch.zeros(1, dtype=torch.int64, device=dev)     # kept at zero between evaluations by dl_epoch_finish
        self.denom2 = 2.0 * float(val_plan.n_pos) * float(val_plan.n_neg)
        # pinned host memory the kernel writes (loss, auc, epoch + 1) into directly: no copy launch per epoch
        self.ring = torch.zeros(self.RING, 4, dtype=torch.float64).pin_memory()
        self.events = [torch.cuda.Event() for _ in range(self.EVENTS)]
        from . import native
        self._native = native.available()

    @staticmethod
    def usable(model, x, val_plan) -> bool:
        import os
        if os.environ.get("DL_DEVICE_EARLY_STOP", "1") == "0" or not x.is_cuda:
            return False
        if getattr(model, "_stacked_params", None) is None or model._stacked_params() is None:
            return False
        denom = float(val_plan.n_pos) * float(val_plan.n_neg)
        return 0 < denom <= val_plan.PAIR_LIMIT and all(b.dtype == torch.float32 for b in model._stacked.values())

    def reset(self):
        """Forget everything (after graph warm-up / capture epochs, which run the launch like any other epoch)."""
        torch.cuda.synchronize()
        self.state.zero_()
        self.hist.zero_()
        self.u2.zero_()
        self.ring.zero_()
        for b, src in zip(self.best, self.bufs):
            b.copy_(src)

    def finish(self, loss: torch.Tensor, score_val: torch.Tensor):
        """The epoch's last two launches: validation counts of `score_val` (float32, the plan's label order) into u2, then
        the bookkeeping.  `loss`: the 0-dim float32 loss of the epoch's forward."""
        score_val = score_val.detach().reshape(-1)
        if score_val.dtype != torch.float32 or not score_val.is_contiguous():
            raise RuntimeError("DeviceEarlyStop.finish: contiguous float32 scores expected")
        loss = loss.detach()
        if loss.dtype != torch.float32 or loss.numel() != 1:
            raise RuntimeError("DeviceEarlyStop.finish: a float32 scalar loss expected")
        self._alive = (loss, score_val)                             # the launches are asynchronous
        p = self.plan
        if self._native:                                            # the same two launches from the compiled binding
            from . import native
            native.epoch_finish(score_val, p.pos_idx, p.neg_idx, self.u2, loss.reshape(1), self.bufs, self.best, self.state,
                                self.hist, self.ring.data_ptr(), self.RING, self.denom2, self.max_epochs, self.patience)
            return
        st = torch.cuda.current_stream().cuda_stream
        self.check(self.lib.dl_auc_pair_counts_add(score_val.data_ptr(), p.pos_idx.data_ptr(), p.n_pos, p.neg_idx.data_ptr(),
                                                   p.n_neg, self.u2.data_ptr(), st), "dl_auc_pair_counts_add")
        self.check(self.lib.dl_epoch_finish(len(self.bufs), self._p, self._b, self._numel, loss.data_ptr(), self.u2.data_ptr(),
                                            self.denom2, self.state.data_ptr(), self.hist.data_ptr(), self.max_epochs,
                                            self.patience, self.ring.data_ptr(), self.RING, st), "dl_epoch_finish")

    def post(self, launch: int):
        """Mark the end of launch number `launch` (an event; not part of a captured graph)."""
        self.events[launch % self.EVENTS].record()

    def read(self, epoch: int, launch: int | None = None):
        """(loss, auc) of `epoch`, once the launch that holds it is done: from the pinned slot the kernel wrote."""
        slot = epoch % self.RING
        self.events[(epoch // self.per if launch is None else launch) % self.EVENTS].synchronize()
        loss_v, auc, tag, _ = self.ring[slot].tolist()
        if int(tag) != epoch + 1:                                   # cannot happen while LAG < RING and the run has not stopped
            raise RuntimeError(f"DeviceEarlyStop: slot {slot} holds epoch {int(tag) - 1}, expected {epoch}")
        return loss_v, auc

    def restore(self):
        """model.load_state_dict(weights) of main_disentangled.py:215: the best weights back into the shared buffers."""
        with torch.no_grad():
            for b, src in zip(self.bufs, self.best):
                b.copy_(src)


def drive(es: DeviceEarlyStop, epochs: int, patience: int, launch, res, log=None):
    """The epoch loop over `launch()` (which queues es.per epochs, each INCLUDING es.finish): launches run LAG ahead of the
    host's reading of the history; `res` (train.RunResult) receives losses / val_aucs / epochs_run exactly as the
    reference's loop would fill them.  Returns best_auc."""
    best_auc, stale, stopped = 0.0, 0, False
    per = es.per

    def take(e):
        nonlocal best_auc, stale, stopped
        loss_v, auc = es.read(e)
        res.losses.append(loss_v)
        res.val_aucs.append(auc)
        res.epochs_run = e + 1
        if auc > best_auc:
            stale, best_auc = 0, auc
        else:
            stale += 1
        if stale > patience:
            stopped = True
        elif log is not None:
            log(f"epoch: {e} loss: {loss_v} val_auc: {best_auc}")

    done = launched = 0
    while launched < epochs and not stopped:
        launch()
        es.post(launched // per)
        launched += per                                             # (the device ignores epochs past `epochs`)
        while launched - done > es.LAG * per and not stopped:
            take(done)
            done += 1
    while not stopped and done < min(launched, epochs):
        take(done)
        done += 1
    return best_auc
