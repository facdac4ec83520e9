"""Adam over the module's SHARED parameter buffers.

The reference trains with ``torch.optim.Adam(model.parameters(), lr, weight_decay=5e-4)`` (main_disentangled.py:150).
``Disentangle`` keeps its 4K per-factor Parameters as views of 4 contiguous ``[K, ...]`` buffers (model._restack), and the
projection's backward produces their gradients as 4 stacked tensors too — so the update runs on 4 tensors instead of 4K.
On the GPU it is ``dl_adam_step`` of libdisenlink_hip.so: torch.optim.Adam's arithmetic (weight decay added to the
gradient, bias-corrected moments, step counter on the device: no sync, graph-capturable), one float4 per thread in one
launch — torch's fused Adam walks 65,536-element chunks with one block each, 13 blocks and 44 us of latency for this
model's 0.8M parameters; this takes a few microseconds.  ``use_torch_kernel=True`` keeps ``torch._fused_adam_`` over the
4 buffers (bit-identical to ``Adam(fused=True)`` over the views: the test of the stacking itself).  Falls back to a
stacking copy of the gradients when they are not views of one buffer.
"""
from __future__ import annotations

import torch


import os as _os
_HOST_STEP = _os.environ.get("DL_ADAM_HOST_STEP", "1") != "0"      # 0: always the device-side step counter (A/B runs)


def stacked_grad(ps, buf) -> torch.Tensor:
    """The K gradients of one parameter group as ONE [K, ...] tensor shaped like the group's shared buffer.  The
    projection's backward hands out the K slices of a stacked gradient (ops.ProjectStacked): when the K .grad tensors
    lie back to back in one storage — checked by address, since views made in a backward (grad mode off) carry no
    ``_base`` — the stacked tensor is a view over them (no copy); otherwise they are stacked (one copy)."""
    g0 = ps[0].grad
    if g0 is None:
        raise RuntimeError("a parameter has no gradient")
    if g0.is_contiguous() and g0.dtype == buf.dtype and g0.shape == buf.shape[1:]:
        # (autograd hands every parameter a gradient of its own shape and dtype, and the K parameters of a group are
        # alike: what is left to check per gradient is its address and that it is dense; this runs every step)
        p0, step, gl = g0.data_ptr(), g0.numel() * g0.element_size(), ps[-1].grad
        st0 = g0.untyped_storage()
        if gl is not None and gl.untyped_storage().data_ptr() == st0.data_ptr() and \
                st0.nbytes() >= (p0 - st0.data_ptr()) + len(ps) * step:
            i = 0
            for p in ps:
                g = p.grad
                if g is None or g.data_ptr() != p0 + i * step or not g.is_contiguous():
                    break
                i += 1
            else:
                return g0.as_strided(tuple(buf.shape), tuple(buf.stride()))     # the K gradients ARE one stacked tensor
    return torch.stack([p.grad for p in ps])


def flat_view(tensors):
    """One 1-D view over `tensors` when they are dense, of one dtype and lie back to back in ONE storage (the projection
    backward carves its four stacked gradients out of one allocation), else None."""
    t0 = tensors[0]
    st0 = t0.untyped_storage()
    off = t0.data_ptr()
    total = 0
    for t in tensors:
        if not t.is_contiguous() or t.dtype != t0.dtype or t.untyped_storage().data_ptr() != st0.data_ptr() \
                or t.data_ptr() != off + total * t0.element_size():
            return None
        total += t.numel()
    if st0.nbytes() < (off - st0.data_ptr()) + total * t0.element_size():
        return None
    return t0.as_strided((total,), (1,))


class StackedAdam:
    """Drop-in for the training loop's use of ``torch.optim.Adam`` (zero_grad / step / state for graph capture);
    ``capturable=True`` keeps the step counters on the device and does nothing that a HIP-graph capture forbids."""

    def __init__(self, model, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 capturable: bool = False, use_torch_kernel: bool = False):
        if model._stacked_params() is None:
            raise ValueError("StackedAdam needs a module whose parameters alias its stacked buffers")
        self.model = model
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), betas, float(eps), float(weight_decay)
        self.keys = list(model._stacked.keys())
        self.bufs = [model._stacked[k] for k in self.keys]
        self.groups = dict(model._param_groups())                  # key -> the K Parameters viewing bufs[key]
        dev = self.bufs[0].device
        self.exp_avg = [torch.zeros_like(b) for b in self.bufs]
        self.exp_avg_sq = [torch.zeros_like(b) for b in self.bufs]
        # one step counter per buffer, as torch's fused Adam wants them (on the device: no sync, capturable)
        self.steps = [torch.zeros((), dtype=torch.float32, device=dev) for _ in self.bufs]
        self.capturable = capturable
        self.use_torch_kernel = bool(use_torch_kernel) or not self.bufs[0].is_cuda or \
            any(b.dtype != torch.float32 for b in self.bufs)
        self.dl_state = torch.zeros(3, dtype=torch.float32, device=dev)      # dl_adam_step: step, lr/(1-b1^t), sqrt(1-b2^t)
        # what a captured graph replays must stay alive and in place: exposed like torch's optimizer.state
        self.state = {i: {"step": self.steps[i], "exp_avg": self.exp_avg[i], "exp_avg_sq": self.exp_avg_sq[i]}
                      for i in range(len(self.bufs))}
        # the eager loop's step count lives on the host — as a tensor inside `state`, so that whoever resets the optimiser the
        # way torch's are reset (zeroing every tensor of optimizer.state: train._graphed_epoch after its warm-up) resets it too
        self._host_count = torch.zeros((), dtype=torch.int64)
        self._host_stale = False                                    # the device counter ran ahead (captured / replayed steps)
        self.state["dl"] = {"state": self.dl_state, "host_step": self._host_count}
        if not self.use_torch_kernel:
            import ctypes as C
            n = len(self.bufs)
            self._C = C
            self._ptrs = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
            self._p, self._m, self._v = self._ptrs(self.bufs), self._ptrs(self.exp_avg), self._ptrs(self.exp_avg_sq)
            self._numel = (C.c_size_t * n)(*[b.numel() for b in self.bufs])

    def zero_grad(self, set_to_none: bool = True):
        for ps in self.groups.values():
            for p in ps:
                if set_to_none:
                    p.grad = None
                elif p.grad is not None:
                    p.grad.zero_()

    def _native(self) -> bool:
        ok = self.__dict__.get("_native_ok")
        if ok is None:
            from . import native
            ok = native.available()
            self._native_ok = ok
            self._flat_params = [p for k in self.keys for p in self.groups[k]]
        return ok

    def _stacked_grad(self, key) -> torch.Tensor:
        return stacked_grad(self.groups[key], self.model._stacked[key])

    @torch.no_grad()
    def step(self):
        if [self.model._stacked[k].data_ptr() for k in self.keys] != [b.data_ptr() for b in self.bufs]:
            raise RuntimeError("the module's parameter buffers were rebuilt (.to() / load on another device): "
                               "create the optimiser afterwards")
        # an eager loop counts its steps on the host (dl_adam_step_at: no counter launch in front of the update); a loop that
        # is captured and replayed cannot — its counter lives in dl_state (dl_adam_step)
        host_step = 0
        if not self.use_torch_kernel and self.bufs[0].is_cuda and torch.cuda.is_current_stream_capturing():
            # a step number passed as a kernel argument would be baked into the graph: every replay would repeat it.  Under
            # capture the counter is the device's (dl_state[0], which dl_adam_step_at keeps current too), whatever `capturable` says
            self._host_stale = True
        elif not self.use_torch_kernel and not self.capturable and _HOST_STEP:
            if self._host_stale:                                    # replays advanced the device counter: read it back once
                self._host_count.fill_(int(self.dl_state[0].item()))
                self._host_stale = False
            self._host_count += 1
            host_step = int(self._host_count)
        if not self.use_torch_kernel and self._native():
            from . import native                                   # gradient bookkeeping + launch in C++ (no ctypes, no per-parameter Python)
            native.adam_step(self.bufs, self._flat_params, self.exp_avg, self.exp_avg_sq, self.dl_state, self.lr, self.betas[0],
                             self.betas[1], self.eps, self.weight_decay, host_step)
            return
        grads = [self._stacked_grad(k) for k in self.keys]
        if not self.use_torch_kernel:
            from . import _lib
            self._grads_alive = grads                              # the launch is asynchronous
            gp = self._ptrs([g.contiguous() for g in grads])
            st = torch.cuda.current_stream().cuda_stream
            if host_step:
                _lib.check(_lib.load().dl_adam_step_at(len(self.bufs), self._p, gp, self._m, self._v, self._numel,
                                                       self.dl_state.data_ptr(), host_step, self.lr, self.betas[0], self.betas[1],
                                                       self.eps, self.weight_decay, st), "dl_adam_step_at")
            else:
                _lib.check(_lib.load().dl_adam_step(len(self.bufs), self._p, gp, self._m, self._v, self._numel,
                                                    self.dl_state.data_ptr(), self.lr, self.betas[0], self.betas[1], self.eps,
                                                    self.weight_decay, st), "dl_adam_step")
            return
        torch._foreach_add_(self.steps, 1)
        torch._fused_adam_(self.bufs, grads, self.exp_avg, self.exp_avg_sq, [], self.steps, lr=self.lr,
                           beta1=self.betas[0], beta2=self.betas[1], weight_decay=self.weight_decay, eps=self.eps,
                           amsgrad=False, maximize=False)
