"""Per-run split / adjacency / pair-set builder, emitting edge rows and pair lists instead of the
reference's twelve dense ``[N,N]`` masks (main_disentangled.py:134-190).

Semantics kept from the reference:
  * 85 / 5 / 10 split over edge ROWS (``train_test_split(range(E_rows), train_size=0.85)`` then
    2/3 of the rest = test) — :134-136.  The reference does not seed it (SURVEY.md §0 finding 5);
    here the permutation comes from ``numpy.random.default_rng(seed)`` so runs can be repeated.
  * the training adjacency is the binarised symmetrisation of the train rows — :139-142.
  * negatives: m independent draws; for every edge row (i, j) a node k with (i, k) not among the
    edge rows (PyG ``structured_negative_sampling``, call site :160; the PyG version is unpinned in
    the reference, so only this contract is reproduced, not its random stream).
  * labels are read from ``ori_adj`` (membership in the directed edge rows) — :195, :203, :218.
  * duplicates.  The TRAIN masks ``pos_train_adj`` / ``neg_train_adj`` are ``sparse_coo(...).to_dense()`` and are
    never binarised (:176-179): duplicate index pairs SUM, and the loss indexes ``a_pred[mask == 1]`` (:195), so a
    pair that occurs more than once — a repeated edge row, or the same (i, k) drawn by two of the m negative draws —
    is NOT part of the loss.  The train pair lists here therefore keep exactly the pairs that occur once.  The
    validation / test masks ``all_val_adj`` / ``all_test_adj`` ARE binarised (:187-190): those lists are unique'd.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class PairSet:
    u: np.ndarray          # int64 [P]
    v: np.ndarray
    label: np.ndarray      # float32 [P], ori_adj[u, v]


@dataclass
class LinkSplit:
    n_nodes: int
    train_src: np.ndarray  # train edge rows (directed, duplicates kept) -> adj via Graph.from_edge_rows
    train_dst: np.ndarray
    pos_train: PairSet     # loss term 1   (a_pred[pos_train_adj == 1])
    neg_train: PairSet     # loss term 2   (a_pred[neg_train_adj == 1]) / m
    val: PairSet           # all_val_adj == 1
    test: PairSet          # all_test_adj == 1
    m: int
    raw: dict | None = None  # make_link_split(keep_raw=True): the index lists before de-duplication (fixture generators)


def _pair_set(u, v, n, edge_keys_sorted, only_single: bool) -> PairSet:
    """Distinct pairs in row-major order (the order mask indexing visits them).  only_single: keep the pairs that occur
    exactly once (``mask == 1`` on a summed, un-binarised mask); else every distinct pair (a binarised mask)."""
    key, cnt = np.unique(u.astype(np.int64) * n + v.astype(np.int64), return_counts=True)
    if only_single:
        key = key[cnt == 1]
    uu, vv = key // n, key % n
    pos = np.searchsorted(edge_keys_sorted, key)
    pos = np.minimum(pos, max(edge_keys_sorted.size - 1, 0))
    label = (edge_keys_sorted[pos] == key).astype(np.float32) if edge_keys_sorted.size else np.zeros(key.size, np.float32)
    return PairSet(uu, vv, label)


def structured_negatives(src, n, edge_keys_sorted, rng) -> np.ndarray:
    """For every row i = src[r] one node k, uniform over [0, n), with (i, k) not an edge row."""
    k = rng.integers(0, n, size=src.size)
    bad = np.ones(src.size, dtype=bool)
    for _ in range(1000):
        key = src[bad] * n + k[bad]
        pos = np.minimum(np.searchsorted(edge_keys_sorted, key), edge_keys_sorted.size - 1)
        hit = edge_keys_sorted[pos] == key
        idx = np.flatnonzero(bad)
        bad[idx[~hit]] = False
        if not bad.any():
            return k
        k[bad] = rng.integers(0, n, size=int(bad.sum()))
    raise RuntimeError("negative sampling did not converge (a node is connected to every node)")


def make_link_split(src, dst, n_nodes: int, m: int = 5, seed: int = 0, keep_raw: bool = False) -> LinkSplit:
    src = np.asarray(src, dtype=np.int64)
    dst = np.asarray(dst, dtype=np.int64)
    E = src.size
    rng = np.random.default_rng(seed)
    perm = rng.permutation(E)
    n_train = int(0.85 * E)                                  # sklearn: floor(train_size * n)
    rest = E - n_train
    n_test = int(rest * 2 / 3)
    tr, te, va = perm[:n_train], perm[n_train:n_train + n_test], perm[n_train + n_test:]
    edge_keys = np.unique(src * n_nodes + dst)
    neg_u = {"tr": [], "va": [], "te": []}
    neg_v = {"tr": [], "va": [], "te": []}
    for _ in range(m):
        k = structured_negatives(src, n_nodes, edge_keys, rng)
        for name, idx in (("tr", tr), ("va", va), ("te", te)):
            neg_u[name].append(src[idx])
            neg_v[name].append(k[idx])
    cat = np.concatenate
    ps = lambda u, v, single: _pair_set(u, v, n_nodes, edge_keys, single)
    out = LinkSplit(
        n_nodes=n_nodes, train_src=src[tr], train_dst=dst[tr],
        pos_train=ps(src[tr], dst[tr], True),
        neg_train=ps(cat(neg_u["tr"]), cat(neg_v["tr"]), True),
        val=ps(cat([src[va]] + neg_u["va"]), cat([dst[va]] + neg_v["va"]), False),
        test=ps(cat([src[te]] + neg_u["te"]), cat([dst[te]] + neg_v["te"]), False),
        m=m)
    if keep_raw:                                              # fixture generators rebuild the reference's dense masks from these
        out.raw = dict(neg_train=(cat(neg_u["tr"]), cat(neg_v["tr"])),
                       val=(cat([src[va]] + neg_u["va"]), cat([dst[va]] + neg_v["va"])),
                       test=(cat([src[te]] + neg_u["te"]), cat([dst[te]] + neg_v["te"])))
    return out
