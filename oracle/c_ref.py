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
        for f in (lib.dlo_route, lib.dlo_aggregate, lib.dlo_score_pairs):
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
