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
            from . import _lib
            lib = _lib.load()
            score = score.contiguous()
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
