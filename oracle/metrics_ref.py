"""CPU restatement of the caller-side loss and AUC.  TEST INFRASTRUCTURE.

  pair_bce      main_disentangled.py:195  F.binary_cross_entropy on masked entries:
                mean over the UNIQUE masked pairs, log clamped at -100, negative term / m,
                labels read from ori_adj (a sampled "negative" can carry label 1).
  auc_tie_avg   main_disentangled.py:202-204, :217-219  sklearn.roc_auc_score on fp32
                probabilities == Mann-Whitney U with tie-averaged ranks.
"""
from __future__ import annotations

import numpy as np


def auc_tie_avg(y: np.ndarray, score: np.ndarray) -> float:
    y = np.asarray(y).astype(bool)
    score = np.asarray(score)
    n_pos = int(y.sum())
    n_neg = int(y.size - n_pos)
    if n_pos == 0 or n_neg == 0:
        raise ValueError("AUC undefined with one class")
    order = np.argsort(score, kind="mergesort")
    ss = score[order]
    # average 1-based rank per run of equal scores
    boundary = np.flatnonzero(np.r_[True, ss[1:] != ss[:-1], True])
    lo, hi = boundary[:-1], boundary[1:]
    avg = (lo + 1 + hi) / 2.0
    ranks = np.empty(score.size, dtype=np.float64)
    ranks[order] = np.repeat(avg, hi - lo)
    return float((ranks[y].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * float(n_neg)))


def bce_mean(prob: np.ndarray, label: np.ndarray) -> float:
    """torch BCE: -(y*max(log p,-100) + (1-y)*max(log(1-p),-100)), mean; computed in float32."""
    prob = prob.astype(np.float32)
    with np.errstate(divide="ignore"):
        lp = np.maximum(np.log(prob), np.float32(-100))
        l1p = np.maximum(np.log(np.float32(1) - prob), np.float32(-100))
    per = -(label * lp + (np.float32(1) - label) * l1p)
    return float(per.astype(np.float32).mean(dtype=np.float32))


def pair_bce(prob_pos, label_pos, prob_neg, label_neg, m: int) -> float:
    return bce_mean(prob_pos, label_pos) + bce_mean(prob_neg, label_neg) / m


def bce_grad(prob, label, scale):
    """d/dprob of scale*mean(BCE): torch clamps the log, not the gradient: (p-y)/max(p(1-p),1e-12)/n."""
    prob = prob.astype(np.float32)
    den = np.maximum(prob * (np.float32(1) - prob), np.float32(1e-12))
    return (np.float32(scale) * (prob - label) / den / np.float32(prob.size)).astype(np.float32)
