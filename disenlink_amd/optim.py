"""Adam over the module's SHARED parameter buffers.

The reference trains with ``torch.optim.Adam(model.parameters(), lr, weight_decay=5e-4)`` (main_disentangled.py:150).
``Disentangle`` keeps its 4K per-factor Parameters as views of 4 contiguous ``[K, ...]`` buffers (model._restack), and the
projection's backward produces their gradients as 4 stacked tensors too — so the update can run on 4 tensors instead of
4K: the same fused elementwise kernel (``torch._fused_adam_``: identical arithmetic per element, hence the same bits as
``Adam(fused=True)`` over the views), one launch over 4 chunks lists instead of 32, and an eighth of the optimiser's
per-step Python work (squirrel: 45 -> ~10 us of kernel time per epoch; the eager chameleon epoch is host-bound and gains
more).  Falls back to a stacking copy of the gradients when they are not views of one buffer.
"""
from __future__ import annotations

import torch


class StackedAdam:
    """Drop-in for the training loop's use of ``torch.optim.Adam`` (zero_grad / step / state for graph capture);
    ``capturable=True`` keeps the step counters on the device and does nothing that a HIP-graph capture forbids."""

    def __init__(self, model, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 capturable: bool = False):
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
        # what a captured graph replays must stay alive and in place: exposed like torch's optimizer.state
        self.state = {i: {"step": self.steps[i], "exp_avg": self.exp_avg[i], "exp_avg_sq": self.exp_avg_sq[i]}
                      for i in range(len(self.bufs))}

    def zero_grad(self, set_to_none: bool = True):
        for ps in self.groups.values():
            for p in ps:
                if set_to_none:
                    p.grad = None
                elif p.grad is not None:
                    p.grad.zero_()

    def _stacked_grad(self, key) -> torch.Tensor:
        ps = self.groups[key]
        g0 = ps[0].grad
        if g0 is None:
            raise RuntimeError("StackedAdam.step(): a parameter has no gradient")
        base = g0._base
        buf = self.model._stacked[key]
        if base is not None and base.shape == buf.shape and base.is_contiguous() and base.dtype == buf.dtype:
            p0, step = base.data_ptr(), base.stride(0) * base.element_size()
            if all(p.grad is not None and p.grad._base is base and p.grad.data_ptr() == p0 + i * step
                   for i, p in enumerate(ps)):
                return base                                        # the K gradients ARE one stacked tensor
        return torch.stack([p.grad for p in ps])

    @torch.no_grad()
    def step(self):
        if [self.model._stacked[k].data_ptr() for k in self.keys] != [b.data_ptr() for b in self.bufs]:
            raise RuntimeError("the module's parameter buffers were rebuilt (.to() / load on another device): "
                               "create the optimiser afterwards")
        grads = [self._stacked_grad(k) for k in self.keys]
        torch._foreach_add_(self.steps, 1)
        torch._fused_adam_(self.bufs, grads, self.exp_avg, self.exp_avg_sq, [], self.steps, lr=self.lr,
                           beta1=self.betas[0], beta2=self.betas[1], weight_decay=self.weight_decay, eps=self.eps,
                           amsgrad=False, maximize=False)
