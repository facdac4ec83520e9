"""Edge-list (CSR + pair-list) CPU restatement, numpy.  TEST INFRASTRUCTURE.

The reference is dense ``[K,N,N]``; only entries with ``adj==1`` reach the aggregation and
only scored pairs reach the loss, so the path restates exactly on a CSR of ``adj_sym`` plus
a pair list.  This file is the executable spec of the HIP kernels:
same data layout (``Z [N,K,d]`` row-major), same per-edge arithmetic, same backward
decomposition (Appendix A.3).  Checked against the golden vectors of the reference and
against autograd of ``dense_ref`` in ``tests/test_oracle_golden.py``.

Reference lines followed:
  route        model.py:56-66   e=exp(z.z/t), alpha=e/sum, p=argmax, a=alpha_p
  normaliser   model.py:70-72   s_k[i]=sum_{j in N(i), p=k} a ; 0 -> 1
  aggregate    model.py:73-75   h_k[i]=b z_k[i]+(1-b) sum_j a_ij/s_k[j] z_k[j]   (s of the NEIGHBOUR)
  score        model.py:110-113 P=sigmoid(sum_k (h_k[u].h_k[v]) exp(z_k[u].z_k[v]/t))  (raw exp)
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


# --------------------------------------------------------------------------- graph prep
def csr_from_dense(adj: np.ndarray):
    """rowptr[N+1] int32, col[E] int32 (ascending per row), rev[E] int32 with col/row swapped."""
    n = adj.shape[0]
    r, c = np.nonzero(adj)
    return csr_from_pairs(r, c, n)


def csr_from_pairs(r, c, n: int, symmetrise: bool = False):
    """CSR of the binarised (optionally symmetrised) adjacency given directed rows (duplicates ok).

    main_disentangled.py:139-142: duplicates collapse, adj_sym = (adj + adj.T) != 0.
    """
    r = np.asarray(r, dtype=np.int64)
    c = np.asarray(c, dtype=np.int64)
    if symmetrise:
        r, c = np.concatenate([r, c]), np.concatenate([c, r])
    key = np.unique(r * n + c)
    r, c = key // n, key % n
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rowptr, r + 1, 1)
    rowptr = np.cumsum(rowptr)
    rev = np.searchsorted(key, c * n + r)
    if not (rev < key.size).all() or not (key[np.minimum(rev, key.size - 1)] == c * n + r).all():
        raise ValueError("adjacency is not symmetric: reverse edge missing")
    return rowptr.astype(np.int32), c.astype(np.int32), rev.astype(np.int32)


def edge_rows(rowptr):
    n = rowptr.size - 1
    return np.repeat(np.arange(n, dtype=np.int64), np.diff(rowptr).astype(np.int64))


# --------------------------------------------------------------------------- forward
def _edge_dots(Z, src, dst, chunk=1 << 16):
    """sigma[e,k] = z_k[src].z_k[dst] accumulated in float32, index order 0..d-1."""
    E, K = src.size, Z.shape[1]
    out = np.empty((E, K), dtype=f32)
    for lo in range(0, E, chunk):
        hi = min(E, lo + chunk)
        out[lo:hi] = np.einsum("ekd,ekd->ek", Z[src[lo:hi]], Z[dst[lo:hi]], dtype=f32)
    return out


def route(Z, rowptr, col, t):
    """Per edge: p (uint8), a (f32), alpha [E,K]; per node: s_raw [N,K] (before zero->1).  model.py:56-72."""
    N, K, _ = Z.shape
    src = edge_rows(rowptr)
    dst = col.astype(np.int64)
    sig = _edge_dots(Z, src, dst) / f32(t)
    with np.errstate(over="ignore", invalid="ignore"):
        ex = np.exp(sig, dtype=f32)
        alpha = ex / ex.sum(axis=1, dtype=f32, keepdims=True)
    p = np.argmax(alpha, axis=1)                       # first max wins (NaN rows: numpy picks the NaN)
    a = alpha[np.arange(p.size), p]
    s_raw = np.zeros((N, K), dtype=f32)
    np.add.at(s_raw, (src, p), a)
    return p.astype(np.uint8), a.astype(f32), alpha, s_raw


def aggregate(Z, rowptr, col, p, a, s_raw, beta):
    """H [N,K,d].  model.py:73-75: weight a_ij / s_k[j], s of the neighbour j, zero s -> 1."""
    N, K, d = Z.shape
    s = np.where(s_raw == 0, f32(1), s_raw)
    src = edge_rows(rowptr)
    dst = col.astype(np.int64)
    pk = p.astype(np.int64)
    w = (a / s[dst, pk]).astype(f32)
    acc = np.zeros((N, K, d), dtype=f32)
    np.add.at(acc, (src, pk), w[:, None] * Z[dst, pk])
    return (f32(beta) * Z + f32(1 - beta) * acc).astype(f32)


def score_pairs(Z, H, pu, pv, t, return_parts=False):
    """prob[P] = sigmoid(sum_k (h_k[u].h_k[v]) * exp(z_k[u].z_k[v]/t)).  model.py:110-113."""
    pu = np.asarray(pu, dtype=np.int64)
    pv = np.asarray(pv, dtype=np.int64)
    q = _edge_dots(H, pu, pv)
    with np.errstate(over="ignore", invalid="ignore"):
        ex = np.exp(_edge_dots(Z, pu, pv) / f32(t), dtype=f32)
        logit = (q * ex).sum(axis=1, dtype=f32)
        prob = (f32(1) / (f32(1) + np.exp(-logit, dtype=f32))).astype(f32)
    if return_parts:
        return prob, q, ex
    return prob


def forward(Z, rowptr, col, beta, t):
    p, a, alpha, s_raw = route(Z, rowptr, col, t)
    H = aggregate(Z, rowptr, col, p, a, s_raw, beta)
    return H, p, a, s_raw


# --------------------------------------------------------------------------- backward (Appendix A.3)
def score_pairs_bwd(Z, H, pu, pv, t, g_prob):
    """dZ, dH from d loss / d prob on the scored pairs (sigmoid backward = p(1-p))."""
    pu = np.asarray(pu, dtype=np.int64)
    pv = np.asarray(pv, dtype=np.int64)
    prob, q, ex = score_pairs(Z, H, pu, pv, t, return_parts=True)
    gl = (g_prob * prob * (f32(1) - prob)).astype(f32)
    ch = (gl[:, None] * ex).astype(f32)                    # d logit / d q_k
    cz = (gl[:, None] * q * ex / f32(t)).astype(f32)       # d logit / d (z_u.z_v)
    dH = np.zeros_like(H)
    dZ = np.zeros_like(Z)
    np.add.at(dH, pu, ch[:, :, None] * H[pv])
    np.add.at(dH, pv, ch[:, :, None] * H[pu])
    np.add.at(dZ, pu, cz[:, :, None] * Z[pv])
    np.add.at(dZ, pv, cz[:, :, None] * Z[pu])
    return dZ, dH


def route_aggregate_bwd(Z, rowptr, col, rev, p, a, s_raw, beta, t, dH):
    """dZ from dH through aggregate -> normaliser -> routing softmax (argmax carries no gradient)."""
    N, K, d = Z.shape
    src = edge_rows(rowptr)
    dst = col.astype(np.int64)
    pk = p.astype(np.int64)
    s = np.where(s_raw == 0, f32(1), s_raw)
    E = dst.size
    er = np.arange(E)
    # B1: dw_e = (1-b) dh_p[i].z_p[j]
    dw = f32(1 - beta) * np.einsum("ed,ed->e", dH[src, pk], Z[dst, pk], dtype=f32)
    # B2: ds_k[j] = -sum_{e=(i,j), p=k} dw a / s^2 (zero where the raw sum was zero), da_e = dw/s_p[j] + ds_p[i]
    ds = np.zeros((N, K), dtype=f32)
    np.add.at(ds, (dst, pk), -dw * a / (s[dst, pk] * s[dst, pk]))
    ds = np.where(s_raw == 0, f32(0), ds)
    da = dw / s[dst, pk] + ds[src, pk]
    # B3: dz[i] = b dh[i] + sum_e (1-b) a_rev/s_p[i] dh_p[j]  +  sum_e sum_k c_k z_k[j]
    dZ = (f32(beta) * dH).astype(f32)
    prev, arev = pk[rev], a[rev]
    np.add.at(dZ, (src, prev), (f32(1 - beta) * arev / s[src, prev])[:, None] * dH[dst, prev])
    _p, _a, alpha, _s = route(Z, rowptr, col, t)
    onehot = np.zeros((E, K), dtype=f32)
    onehot[er, pk] = 1
    c = ((da + da[rev]) * a)[:, None] * (onehot - alpha) / f32(t)
    np.add.at(dZ, src, c[:, :, None] * Z[dst])
    return dZ.astype(f32)
