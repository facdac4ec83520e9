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
TORCH_SRC = os.path.join(CSRC, "torch", "dl_torch.cpp")
TORCH_LIB = os.path.join(PKG, "libdisenlink_torch.so")
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


def build_torch_binding(force: bool = False, verbose: bool = False) -> str:
    """libdisenlink_torch.so: the compiled TORCH_LIBRARY binding over the C ABI (csrc/torch/dl_torch.cpp: host C++ only,
    compiled with g++ against the installed torch; links libdisenlink_hip.so next to it).  In-tree, like the kernels."""
    import torch
    from torch.utils import cpp_extension as ce
    deps = [TORCH_SRC, os.path.join(ROOT, "include", "disenlink_hip.h"), LIB]
    if not force and os.path.exists(TORCH_LIB) and os.path.getmtime(TORCH_LIB) >= _newest(deps):
        return TORCH_LIB
    try:
        inc = ce.include_paths(device_type="cuda")
    except TypeError:                                             # older signature
        inc = ce.include_paths(True)
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-shared", "-fPIC", "-std=c++17",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           *[f"-I{i}" for i in inc], "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"), TORCH_SRC, "-o", TORCH_LIB,
           f"-L{tlib}", "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_hip", f"-L{PKG}", "-l:libdisenlink_hip.so",
           f"-Wl,-rpath,{tlib}", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"building the torch binding failed:\n{r.stdout}\n{r.stderr[-4000:]}")
    return TORCH_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
    print(build_torch_binding(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
