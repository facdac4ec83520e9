"""ctypes wrapper of the C restatement (oracle/c/sparse_ref.c).  TEST INFRASTRUCTURE.

Multi-threaded (OpenMP) forward of the edge-list path; used for parity at full benchmark sizes and
as the sparse CPU baseline of bench.py on graphs the dense reference form cannot hold."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import build_c

_lib = None


def _load():
    global _lib
    if _lib is None:
        lib = C.CDLL(build_c.build())
        fp, ip, up = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
        lib.dlo_route.argtypes = [fp, C.c_int, C.c_int, C.c_int, ip, ip, C.c_float, up, fp, fp]
        lib.dlo_aggregate.argtypes = [fp, C.c_int, C.c_int, C.c_int, ip, ip, C.c_float, up, fp, fp, fp]
        lib.dlo_score_pairs.argtypes = [fp, fp, C.c_int, C.c_int, C.c_float, ip, ip, C.c_int64, fp]
        lib.dlo_route_aggregate_bwd.argtypes = [fp, C.c_int, C.c_int, C.c_int, ip, ip, C.c_float, C.c_float, up, fp, fp, fp, fp]
        lib.dlo_score_pairs_bwd.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int, C.c_float, ip, ip, ip, fp, fp, fp, fp]
        for f in (lib.dlo_route, lib.dlo_aggregate, lib.dlo_score_pairs, lib.dlo_route_aggregate_bwd, lib.dlo_score_pairs_bwd):
            f.restype = None
        _lib = lib
    return _lib


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def route(Z, rowptr, col, t):
    lib = _load()
    Z = np.ascontiguousarray(Z, dtype=np.float32)
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    col = np.ascontiguousarray(col, dtype=np.int32)
    N, K, d = Z.shape
    p = np.empty(col.size, dtype=np.uint8)
    a = np.empty(col.size, dtype=np.float32)
    s = np.empty((N, K), dtype=np.float32)
    lib.dlo_route(_f(Z), N, K, d, _i(rowptr), _i(col), float(t), p.ctypes.data_as(C.POINTER(C.c_uint8)), _f(a), _f(s))
    return p, a, s


def aggregate(Z, rowptr, col, p, a, s, beta):
    lib = _load()
    Z = np.ascontiguousarray(Z, dtype=np.float32)
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    col = np.ascontiguousarray(col, dtype=np.int32)
    N, K, d = Z.shape
    H = np.empty_like(Z)
    lib.dlo_aggregate(_f(Z), N, K, d, _i(rowptr), _i(col), float(beta),
                      np.ascontiguousarray(p, dtype=np.uint8).ctypes.data_as(C.POINTER(C.c_uint8)),
                      _f(np.ascontiguousarray(a, dtype=np.float32)), _f(np.ascontiguousarray(s, dtype=np.float32)), _f(H))
    return H


def score_pairs(Z, H, pu, pv, t):
    lib = _load()
    Z = np.ascontiguousarray(Z, dtype=np.float32)
    H = np.ascontiguousarray(H, dtype=np.float32)
    pu = np.ascontiguousarray(pu, dtype=np.int32)
    pv = np.ascontiguousarray(pv, dtype=np.int32)
    prob = np.empty(pu.size, dtype=np.float32)
    lib.dlo_score_pairs(_f(Z), _f(H), Z.shape[1], Z.shape[2], float(t), _i(pu), _i(pv), pu.size, _f(prob))
    return prob


def route_aggregate_bwd(Z, rowptr, col, p, a, s, beta, t, dH):
    """dZ [N,K,d] from dH through aggregate -> normaliser -> routing softmax (gather form; no reverse-edge map needed)."""
    lib = _load()
    Z = np.ascontiguousarray(Z, dtype=np.float32)
    dH = np.ascontiguousarray(dH, dtype=np.float32)
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    col = np.ascontiguousarray(col, dtype=np.int32)
    N, K, d = Z.shape
    dZ = np.empty_like(Z)
    lib.dlo_route_aggregate_bwd(_f(Z), N, K, d, _i(rowptr), _i(col), float(beta), float(t),
                                np.ascontiguousarray(p, dtype=np.uint8).ctypes.data_as(C.POINTER(C.c_uint8)),
                                _f(np.ascontiguousarray(a, dtype=np.float32)), _f(np.ascontiguousarray(s, dtype=np.float32)),
                                _f(dH), _f(dZ))
    return dZ


def pair_incidence(pu, pv, n_nodes):
    """Node-incidence list of a pair list: every pair once per endpoint, grouped by node."""
    pu = np.asarray(pu, dtype=np.int64)
    pv = np.asarray(pv, dtype=np.int64)
    node, other = np.concatenate([pu, pv]), np.concatenate([pv, pu])
    pair = np.concatenate([np.arange(pu.size), np.arange(pu.size)])
    order = np.argsort(node, kind="stable")                 # a fixed order per node: list order of the pairs
    incptr = np.zeros(n_nodes + 1, dtype=np.int64)
    incptr[1:] = np.cumsum(np.bincount(node, minlength=n_nodes))
    return incptr.astype(np.int32), other[order].astype(np.int32), pair[order].astype(np.int32)


def score_pairs_bwd(Z, H, pu, pv, t, prob, g_prob):
    """dZ, dH [N,K,d] from d loss / d prob on the scored pairs (sigmoid backward = p (1 - p))."""
    lib = _load()
    Z = np.ascontiguousarray(Z, dtype=np.float32)
    H = np.ascontiguousarray(H, dtype=np.float32)
    N, K, d = Z.shape
    incptr, other, pair = pair_incidence(pu, pv, N)
    dZ, dH = np.empty_like(Z), np.empty_like(Z)
    lib.dlo_score_pairs_bwd(_f(Z), _f(H), N, K, d, float(t), _i(incptr), _i(other), _i(pair),
                            _f(np.ascontiguousarray(prob, dtype=np.float32)),
                            _f(np.ascontiguousarray(g_prob, dtype=np.float32)), _f(dZ), _f(dH))
    return dZ, dH
