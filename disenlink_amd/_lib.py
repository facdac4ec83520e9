"""ctypes binding of libdisenlink_hip.so (include/disenlink_hip.h).

There is no CPU or eager fallback: if the library is missing, or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "libdisenlink_hip.so")


class DlCsrPlan(C.Structure):
    _fields_ = [
        ("n_rows", C.c_int32), ("row_offset", C.c_int32), ("n_total", C.c_int32), ("n_entries", C.c_int32),
        ("rowptr", C.c_void_p), ("col", C.c_void_p),
        ("seg_len", C.c_int32), ("n_seg", C.c_int32),
        ("seg_row", C.c_void_p), ("seg_beg", C.c_void_p), ("seg_end", C.c_void_p), ("seg_slot", C.c_void_p),
        ("n_slices", C.c_int32), ("slice_max_seg", C.c_int32), ("slice_seg0", C.c_void_p),
        ("n_multi", C.c_int32), ("n_slots", C.c_int32), ("multi_row", C.c_void_p), ("multi_slot0", C.c_void_p),
        ("slot_multi", C.c_void_p), ("unit_count", C.c_void_p),
    ]


class DlGraph(C.Structure):
    _fields_ = [("csr", DlCsrPlan), ("route", DlCsrPlan), ("rev", C.c_void_p), ("route_mirror", C.c_int32)]


class DlPairIncidence(C.Structure):
    _fields_ = [("csr", DlCsrPlan), ("inc_pair", C.c_void_p), ("n_pairs", C.c_int32), ("entry_yw", C.c_void_p)]


class DlHostCsr(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("n_entries", C.c_int32), ("rowptr", C.POINTER(C.c_int32)),
                ("col", C.POINTER(C.c_int32)), ("rev", C.POINTER(C.c_int32))]


class DlHostPlan(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("seg_len", "n_seg", "n_slices", "slice_max_seg", "n_multi", "n_slots")] + \
               [(n, C.POINTER(C.c_int32)) for n in ("seg_row", "seg_beg", "seg_end", "seg_slot", "slice_seg0",
                                                    "multi_row", "multi_slot0", "slot_multi")]


_P = C.c_void_p          # device pointers travel as integers
_G, _I = C.POINTER(DlGraph), C.POINTER(DlPairIncidence)
_i, _f, _z = C.c_int, C.c_float, C.c_size_t
EXPORTS = {
    # name: (restype, argtypes) -- one entry per symbol declared in include/disenlink_hip.h
    "dl_host_csr_from_edges": (_i, [_P, _P, C.c_int64, C.c_int32, _i, C.POINTER(DlHostCsr)]),
    "dl_host_csr_free": (None, [C.POINTER(DlHostCsr)]),
    "dl_host_plan_build": (_i, [C.c_int32, C.c_int32, _P, _P, C.c_int32, C.c_int32, _P, C.c_int32, C.c_int32,
                                C.POINTER(DlHostPlan)]),
    "dl_host_plan_free": (None, [C.POINTER(DlHostPlan)]),
    "dl_version": (C.c_char_p, []),
    "dl_last_error": (C.c_char_p, []),
    "dl_config_reload": (None, []),
    "dl_has_fast_path": (_i, [_i, _i]),
    "dl_has_fast_path_dtype": (_i, [_i, _i, _i]),
    "dl_set_force_generic": (_i, [_i]),
    "dl_workspace_bytes": (_z, [C.POINTER(DlCsrPlan), _i, _i]),
    "dl_project_supported": (_i, [_i]),
    "dl_project_fwd_workspace_bytes": (_z, [_i, _i, _i, _i, _i, _i]),
    "dl_project_hidden_floats": (_z, [_i, _i, _i]),
    "dl_project_fwd": (_i, [_P, _i, _i, _i, _i, _i, _P, _P, _P, _P, _P, _P, _P, _z, _P]),
    "dl_project_xplanes_bytes": (_z, [_i, _i]),
    "dl_project_xplanes_build": (_i, [_P, _i, _i, _P, _z, _P]),
    "dl_project_fwd_xp": (_i, [_P, _i, _i, _i, _i, _i, _P, _P, _P, _P, _P, _P, _P, _z, _P, _P]),
    "dl_project_bwd_xp": (_i, [_P, _i, _i, _i, _i, _i, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _z, _P, _P]),
    "dl_project_bwd_workspace_bytes": (_z, [_i, _i, _i, _i, _i, _i]),
    "dl_project_bwd": (_i, [_P, _i, _i, _i, _i, _i, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _z, _P]),
    "dl_route_fwd": (_i, [_G, _P, _i, _i, _i, _f, _P, _P, _P, _P, _z, _P]),
    "dl_aggregate_fwd": (_i, [_G, _P, _i, _i, _i, _f, _P, _P, _P, _P, _P, _z, _P]),
    "dl_score_pairs_fwd": (_i, [_P, _P, _i, _i, _i, _i, _f, _P, _P, _i, _I, _P, _P, _P]),
    "dl_score_allpairs_workspace_bytes": (_z, [_i, _i, _i, _i]),
    "dl_score_allpairs_fwd": (_i, [_P, _P, _i, _i, _i, _i, _f, _P, _P, _z, _P]),
    "dl_score_allpairs_bwd": (_i, [_P, _P, _i, _i, _i, _i, _f, _I, _P, _P, _i, _P, _P, _P, _P, _P, _z, _P]),
    "dl_auc_pair_counts_supported": (_i, [_i, _i]),
    "dl_auc_pair_counts": (_i, [_P, _P, _i, _P, _i, _P, _P]),
    "dl_auc_pair_counts_add": (_i, [_P, _P, _i, _P, _i, _P, _P]),
    "dl_epoch_state_bytes": (C.c_size_t, []),
    "dl_epoch_finish": (_i, [_i, _P, _P, _P, _P, _P, C.c_double, _P, _P, C.c_longlong, C.c_longlong, _P, _i, _P]),
    "dl_score_pairs_train_supported": (_i, [_P, _i, _i, _i]),
    "dl_score_pairs_train": (_i, [_P, _P, _i, _i, _i, _f, _P, _P, _P, _P, _P, _P, _P, _z, _P]),
    "dl_adam_step": (_i, [_i, _P, _P, _P, _P, _P, _P, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _P]),
    "dl_adam_step_at": (_i, [_i, _P, _P, _P, _P, _P, _P, C.c_longlong, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _P]),
    "dl_pair_bce": (_i, [_P, _P, _P, _i, _P, _P, _P, _z, _P]),
    "dl_score_pairs_bwd": (_i, [_P, _P, _i, _i, _i, _f, _I, _P, _P, _P, _P, _P, _P, _z, _P]),
    "dl_route_aggregate_bwd_phase1": (_i, [_G, _P, _i, _i, _i, _f, _P, _P, _P, _P, _P, _P, _P, _P, _z, _P]),
    "dl_route_aggregate_bwd_phase2": (_i, [_G, _P, _i, _i, _i, _f, _f, _P, _P, _P, _P, _P, _P, _P, _P, _i, _P, _z, _P]),
    "dl_route_aggregate_bwd": (_i, [_G, _P, _i, _i, _i, _f, _f, _P, _P, _P, _P, _P, _i, _P, _z, _P]),
    "dl_route_aggregate_bwd_scaled": (_i, [_G, _P, _i, _i, _i, _f, _f, _P, _P, _P, _P, _P, _P, _P, _P, _z, _P]),
}

DL_F32, DL_BF16 = 0, 1

_lib = None


class DisenlinkHipError(RuntimeError):
    pass


def _hip_runtime_global() -> None:
    """libdisenlink_hip.so is linked without a HIP runtime (-no-hip-rt) and binds to the one the host
    process uses.  Under PyTorch that is torch's bundled libamdhip64.so, so that torch's streams and
    allocations are valid in our launches; promote it to the global symbol scope before loading."""
    import torch  # noqa: F401  (loads torch's HIP runtime first)
    cand = [os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"),
            "/opt/rocm/lib/libamdhip64.so"]
    for path in cand:
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)
            return
    raise DisenlinkHipError("no libamdhip64.so found (looked in torch/lib and /opt/rocm/lib)")


def load() -> C.CDLL:
    """Load the library once.  Raises if it was not built (python -m disenlink_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("DL_LIB_PATH", LIB_PATH)      # override: kernel experiments with alternative builds
    if not os.path.exists(path):
        raise DisenlinkHipError(
            f"{path} not found: the HIP library is required (there is no CPU fallback). "
            "Build it with `python -m disenlink_amd.build` or __graft_entry__.build().")
    _hip_runtime_global()
    lib = C.CDLL(path)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)      # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().dl_last_error().decode(errors="replace")
        raise DisenlinkHipError(f"{what} failed (code {rc}): {msg}")


def config_reload() -> None:
    """Have the library read its environment switches again (it reads them once, at first use: csrc/dl_config.h).
    For tests and A/B scripts that change a DL_* variable inside a running process."""
    load().dl_config_reload()
