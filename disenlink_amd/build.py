"""Build libdisenlink_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m disenlink_amd.build [--force] [--verbose]
"""
from __future__ import annotations

import concurrent.futures as cf
import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(PKG, "libdisenlink_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def _newest(paths):
    return max((os.path.getmtime(p) for p in paths), default=0.0)


def _compile(src, obj, verbose):
    cmd = [HIPCC, *FLAGS, *os.environ.get("DL_CXXFLAGS", "").split(), "-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr, file=sys.stderr)
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    if not srcs:
        raise RuntimeError(f"no HIP sources under {CSRC}")
    os.makedirs(OBJ, exist_ok=True)
    hdr_time = _newest(hdrs)
    jobs, objs = [], []
    for src in srcs:
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        stale = force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time)
        if stale:
            jobs.append((src, obj))
    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(lambda so: _compile(so[0], so[1], verbose), jobs))
    if jobs or not os.path.exists(LIB) or os.path.getmtime(LIB) < _newest(objs):
        # -no-hip-rt: leave the HIP runtime symbols undefined so the library binds to the runtime the
        # host process already uses (torch bundles its own libamdhip64; two runtimes in one process
        # would not share streams).  disenlink_amd/_lib.py puts that runtime in the global scope.
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-no-hip-rt", "-o", LIB, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
