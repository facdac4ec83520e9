"""Caller-side loss and AUC on pair lists, on the device.

  pair_bce_loss   main_disentangled.py:195 — BCE(mean, log clamped at -100 by torch) on the positive
                  pairs + BCE on the negative pairs / m, in PROBABILITY space (finding 4 of SURVEY.md §0:
                  saturated fp32 sigmoids must give exactly zero gradient, so no with-logits fusion).
  auc_tie_avg     main_disentangled.py:202-204, :217-219 — sklearn.roc_auc_score on fp32 probabilities,
                  i.e. Mann-Whitney U with tie-averaged ranks; computed with torch ops on the device so
                  the per-epoch device->host copy of all validation scores disappears.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def pair_bce_loss(prob_pos, label_pos, prob_neg, label_neg, m: int) -> torch.Tensor:
    return F.binary_cross_entropy(prob_pos, label_pos) + F.binary_cross_entropy(prob_neg, label_neg) / m


def pair_bce_weights(n_pos: int, n_neg: int, m: int, device) -> torch.Tensor:
    """Per-pair weights that turn sum_q w BCE into BCE_mean(pos) + BCE_mean(neg) / m for a list [pos | neg]."""
    w = torch.empty(n_pos + n_neg, dtype=torch.float32, device=device)
    w[:n_pos] = 1.0 / max(n_pos, 1)
    w[n_pos:] = 1.0 / (m * max(n_neg, 1))
    return w


def pair_bce_loss_fused(prob_train, label_train, weight_train) -> torch.Tensor:
    """The same loss from the fused HIP kernel (loss and its gradient in one pass); GPU tensors only."""
    from .ops import PairBCE
    return PairBCE.apply(prob_train, label_train, weight_train)


def auc_tie_avg(label: torch.Tensor, score: torch.Tensor, check: bool = True) -> torch.Tensor:
    """0-dim float64 tensor on score.device.  ``check=True`` raises if only one class is present (one
    device->host sync); with ``check=False`` nothing here synchronises and a one-class input gives nan."""
    label = label.reshape(-1)
    score = score.reshape(-1).detach()
    n = score.numel()
    pos = label > 0.5
    n_pos = pos.sum().to(torch.float64)
    n_neg = n - n_pos
    if check and (float(n_pos) == 0 or float(n_neg) == 0):
        raise ValueError("AUC undefined with one class")
    order = torch.argsort(score, stable=True)
    ss = score[order]
    # a run of equal scores occupies sorted positions [first, last]: average 1-based rank (first+last)/2 + 1.
    # Two binary searches instead of unique / nonzero: no data-dependent shape, hence no host sync.
    first = torch.searchsorted(ss, ss, right=False)
    last = torch.searchsorted(ss, ss, right=True) - 1
    rank_sorted = (first + last).to(torch.float64) / 2.0 + 1.0
    r_pos = (rank_sorted * pos[order].to(torch.float64)).sum()
    return (r_pos - n_pos * (n_pos + 1.0) / 2.0) / (n_pos * n_neg)


class AucPlan:
    """The two index sets of a FIXED label vector (validation / test labels do not change during a run), found once.
    ``auc(score)`` then needs no sort of the whole score vector: only the negatives are sorted, every positive is
    located among them by two binary searches, and

        AUC = ( sum_p #{n: s_n < s_p} + 1/2 #{n: s_n == s_p} ) / (n_pos * n_neg)

    — the Mann-Whitney statistic with tie-averaged ranks, i.e. sklearn.roc_auc_score (main_disentangled.py:202-204,
    :217-219), from integer counts.  No data-dependent shapes: nothing synchronises, and it can be graph-captured."""

    PAIR_LIMIT = 4.0e11              # dl_auc_pair_counts_supported: n_pos * n_neg up to which the kernel is used

    def __init__(self, label: torch.Tensor):
        label = label.reshape(-1)
        pos = label > 0.5
        self.pos_idx = torch.nonzero(pos).reshape(-1)              # the one sync, at construction
        self.neg_idx = torch.nonzero(~pos).reshape(-1)
        self.n_pos, self.n_neg = int(self.pos_idx.numel()), int(self.neg_idx.numel())
        # 2 * n_pos * n_neg as a device tensor: a tensor / tensor division is correctly rounded, a division by a Python
        # scalar is turned into a multiplication by its reciprocal on the GPU (1 ulp off: 0.49999999999999994 for 1/2)
        self._denom2 = torch.full((), 2.0 * float(self.n_pos) * float(self.n_neg), dtype=torch.float64, device=label.device)

    def auc(self, score: torch.Tensor) -> torch.Tensor:
        """0-dim float64 tensor on score.device (nan when one class is absent)."""
        score = score.reshape(-1).detach()
        denom = float(self.n_pos) * float(self.n_neg)
        if score.is_cuda and score.dtype == torch.float32 and 0 < denom <= self.PAIR_LIMIT:
            # dl_auc_pair_counts: one launch — slices of the smaller class sorted in LDS, the other class located in
            # them by binary searches; exact integer counts (~10 us at 10^4 x 5*10^4 against ~130 us for the sort
            # path below, which only enormous validation sets take)
            from . import _lib, native
            score = score.contiguous()
            if native.available():                                 # the same launch from the compiled binding
                return native.auc_pair_counts(score, self.pos_idx, self.neg_idx)[0].to(torch.float64) / self._denom2
            lib = _lib.load()
            u2 = torch.empty(1, dtype=torch.int64, device=score.device)
            _lib.check(lib.dl_auc_pair_counts(score.data_ptr(), self.pos_idx.data_ptr(), self.n_pos, self.neg_idx.data_ptr(),
                                              self.n_neg, u2.data_ptr(), torch.cuda.current_stream().cuda_stream),
                       "dl_auc_pair_counts")
            return u2[0].to(torch.float64) / self._denom2
        sp = score.index_select(0, self.pos_idx)
        # stable=True: the merge-sort path, the one the rank-based auc_tie_avg has always used (also under HIP-graph
        # capture); which order equal negatives end up in does not matter here
        sn = torch.sort(score.index_select(0, self.neg_idx), stable=True).values
        below = torch.searchsorted(sn, sp, right=False)
        upto = torch.searchsorted(sn, sp, right=True)
        u2 = (below + upto).sum()                                   # 2 * (below + (upto - below) / 2), in int64: exact
        return u2.to(torch.float64) / self._denom2.to(score.device) if denom > 0 else torch.full((), float("nan"), dtype=torch.float64,
                                                                                device=score.device)


class ShardedAucPlan:
    """Tie-aware AUC of an evaluation set whose scores are spread over the ranks of a process group (row-sharded runs:
    every rank holds the scores of the pairs whose first endpoint it owns).  The Mann-Whitney counts are ADDITIVE over
    slices of either class:

        2 U = sum over ranks r of  sum_{p in pos(r)} ( 2 #{n: s_n < s_p} + #{n: s_n == s_p} ),   n over ALL negatives

    so per evaluation: one all-gather of the negatives' scores (padded to the largest rank's count; the sizes are fixed
    for a run), each rank counts ITS positives against all negatives (the same kernel as AucPlan — dl_auc_pair_counts — on
    the GPU, a sort + two binary searches elsewhere), and the integer counts are summed by the caller's all-reduce
    (`partial(score)` returns this rank's 2U as an exact float64; `auc(score)` does the all-reduce itself).  Equal to
    sklearn.roc_auc_score on the concatenated vectors (main_disentangled.py:202-204, :217-219), whatever the sharding."""

    def __init__(self, label_local: torch.Tensor, group=None):
        import torch.distributed as dist
        self.group, self.dist = group, dist
        label_local = label_local.reshape(-1)
        dev = label_local.device
        pos = label_local > 0.5
        self.pos_idx = torch.nonzero(pos).reshape(-1)
        self.neg_idx = torch.nonzero(~pos).reshape(-1)
        world = dist.get_world_size(group)
        self._cpu_wire = label_local.is_cuda and dist.get_backend(group) == "gloo"
        red = "cpu" if self._cpu_wire else dev
        mine = torch.tensor([self.pos_idx.numel(), self.neg_idx.numel()], dtype=torch.int64, device=red)
        allc = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allc, mine, group=group)
        counts = torch.stack(allc).cpu()
        self.n_pos, self.n_neg = int(counts[:, 0].sum()), int(counts[:, 1].sum())
        self.max_neg = max(1, int(counts[:, 1].max()))
        self.world = world
        # positions of the VALID entries of the gathered [world, max_neg] buffer; this rank's scores follow it
        valid = torch.cat([r * self.max_neg + torch.arange(int(counts[r, 1])) for r in range(world)]) if self.n_neg else \
            torch.zeros(0, dtype=torch.int64)
        self.all_neg_idx = valid.to(dev)
        self.local_pos_idx = (self.pos_idx + world * self.max_neg).contiguous()
        self.n_local_pos = int(self.pos_idx.numel())
        self._denom2 = 2.0 * float(self.n_pos) * float(self.n_neg)

    def partial(self, score_local: torch.Tensor) -> torch.Tensor:
        """This rank's share of 2U (0-dim float64 on score_local.device; exact: an integer below 2^53)."""
        dist = self.dist
        score_local = score_local.reshape(-1).detach().float()
        dev = score_local.device
        send = torch.zeros(self.max_neg, dtype=torch.float32, device=dev)
        send[: self.neg_idx.numel()] = score_local.index_select(0, self.neg_idx)
        buf = torch.empty(self.world * self.max_neg + score_local.numel(), dtype=torch.float32, device=dev)
        gathered = buf[: self.world * self.max_neg]
        if self._cpu_wire:
            host = torch.empty(gathered.shape, dtype=torch.float32)
            dist.all_gather_into_tensor(host, send.cpu(), group=self.group)
            gathered.copy_(host)
        else:
            dist.all_gather_into_tensor(gathered, send, group=self.group)
        buf[self.world * self.max_neg:] = score_local
        if self.n_local_pos == 0 or self.n_neg == 0:
            return torch.zeros((), dtype=torch.float64, device=dev)
        if buf.is_cuda and float(self.n_local_pos) * float(self.n_neg) <= AucPlan.PAIR_LIMIT:
            from . import _lib
            u2 = torch.empty(1, dtype=torch.int64, device=dev)
            _lib.check(_lib.load().dl_auc_pair_counts(buf.data_ptr(), self.local_pos_idx.data_ptr(), self.n_local_pos,
                                                      self.all_neg_idx.data_ptr(), self.n_neg, u2.data_ptr(),
                                                      torch.cuda.current_stream().cuda_stream), "dl_auc_pair_counts")
            return u2[0].to(torch.float64)
        sp = buf.index_select(0, self.local_pos_idx)
        sn = torch.sort(buf.index_select(0, self.all_neg_idx), stable=True).values
        return (torch.searchsorted(sn, sp, right=False) + torch.searchsorted(sn, sp, right=True)).sum().to(torch.float64)

    def auc_from_sum(self, u2_sum) -> float:
        return float(u2_sum) / self._denom2 if self._denom2 > 0 else float("nan")

    def auc(self, score_local: torch.Tensor) -> float:
        u2 = self.partial(score_local)
        red = u2.cpu() if self._cpu_wire else u2
        self.dist.all_reduce(red, group=self.group)
        return self.auc_from_sum(red)
