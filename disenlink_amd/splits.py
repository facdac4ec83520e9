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


# The sorts and binary searches of a split are integer work with one right answer: on a large graph (snap-patents: 14M
# edge rows, 70M negative draws) they take minutes in numpy on one core and seconds with torch on the GPU.  ``device``
# (optional) moves exactly those steps; every random draw stays in numpy, so the split is the same bit for bit.
def _unique(a: np.ndarray, counts: bool = False, device=None):
    if device is None:
        return np.unique(a, return_counts=counts)
    import torch
    t = torch.as_tensor(a, device=device)
    if counts:
        k, c = torch.unique(t, sorted=True, return_counts=True)
        return k.cpu().numpy(), c.cpu().numpy()
    return torch.unique(t, sorted=True).cpu().numpy()


def _member(sorted_keys: np.ndarray, key: np.ndarray, device=None) -> np.ndarray:
    """key[i] in sorted_keys, elementwise"""
    if sorted_keys.size == 0:
        return np.zeros(key.size, dtype=bool)
    if device is None:
        pos = np.minimum(np.searchsorted(sorted_keys, key), sorted_keys.size - 1)
        return sorted_keys[pos] == key
    import torch
    sk = sorted_keys if torch.is_tensor(sorted_keys) else torch.as_tensor(sorted_keys, device=device)
    k = torch.as_tensor(key, device=device)
    pos = torch.searchsorted(sk, k).clamp_(max=sk.numel() - 1)
    return (sk[pos] == k).cpu().numpy()


def _pair_set(u, v, n, edge_keys_sorted, only_single: bool, device=None) -> PairSet:
    """Distinct pairs in row-major order (the order mask indexing visits them).  only_single: keep the pairs that occur
    exactly once (``mask == 1`` on a summed, un-binarised mask); else every distinct pair (a binarised mask)."""
    key, cnt = _unique(u.astype(np.int64) * n + v.astype(np.int64), counts=True, device=device)
    if only_single:
        key = key[cnt == 1]
    uu, vv = key // n, key % n
    label = _member(edge_keys_sorted, key, device).astype(np.float32)
    return PairSet(uu, vv, label)


def structured_negatives(src, n, edge_keys_sorted, rng, device=None) -> np.ndarray:
    """For every row i = src[r] one node k, uniform over [0, n), with (i, k) not an edge row."""
    k = rng.integers(0, n, size=src.size)
    bad = np.ones(src.size, dtype=bool)
    for _ in range(1000):
        key = src[bad] * n + k[bad]
        hit = _member(edge_keys_sorted, key, device)
        idx = np.flatnonzero(bad)
        bad[idx[~hit]] = False
        if not bad.any():
            return k
        k[bad] = rng.integers(0, n, size=int(bad.sum()))
    raise RuntimeError("negative sampling did not converge (a node is connected to every node)")


def make_link_split(src, dst, n_nodes: int, m: int = 5, seed: int = 0, keep_raw: bool = False, device=None) -> LinkSplit:
    """``device`` (optional, a torch device): run the sorts / searches there (same split, bit for bit)."""
    src = np.asarray(src, dtype=np.int64)
    dst = np.asarray(dst, dtype=np.int64)
    E = src.size
    rng = np.random.default_rng(seed)
    perm = rng.permutation(E)
    n_train = int(0.85 * E)                                  # sklearn: floor(train_size * n)
    rest = E - n_train
    n_test = int(rest * 2 / 3)
    tr, te, va = perm[:n_train], perm[n_train:n_train + n_test], perm[n_train + n_test:]
    edge_keys = _unique(src * n_nodes + dst, device=device)
    ek = edge_keys
    if device is not None:                                    # searched 2 + 5 m times: keep one copy on the device
        import torch
        ek = torch.as_tensor(edge_keys, device=device)
    neg_u = {"tr": [], "va": [], "te": []}
    neg_v = {"tr": [], "va": [], "te": []}
    for _ in range(m):
        k = structured_negatives(src, n_nodes, ek, rng, device)
        for name, idx in (("tr", tr), ("va", va), ("te", te)):
            neg_u[name].append(src[idx])
            neg_v[name].append(k[idx])
    cat = np.concatenate
    ps = lambda u, v, single: _pair_set(u, v, n_nodes, ek, single, device)
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
