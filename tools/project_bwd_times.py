"""Projection backward: dl_project_bwd (MFMA, recomputed hidden layer) vs the library-GEMM form, kernel time only.
usage: python tools/project_bwd_times.py [N F K nhid d]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disenlink_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
shapes = [tuple(int(v) for v in sys.argv[1:6])] if len(sys.argv) >= 6 else [
    (5201, 128, 8, 512, 64), (5201, 2089, 8, 512, 64), (41554, 128, 16, 512, 128), (2277, 2325, 5, 512, 32),
    (5201, 128, 8, 1, 64)]


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for N, F, K, nhid, d in shapes:
    two = nhid > 1
    x, dZ = torch.randn(N, F, device=dev), torch.randn(N, K, d, device=dev)
    W1 = torch.randn(K, nhid if two else d, F, device=dev) / F ** 0.5
    b1 = torch.randn(K, nhid if two else d, device=dev) * 0.1
    W2 = torch.randn(K, d, nhid, device=dev) / nhid ** 0.5 if two else None
    t_native = timeit(lambda: ops.project_bwd(x, W1, b1, W2, dZ))
    os.environ["DL_PROJECT_BWD"] = "library"
    t_lib = timeit(lambda: ops._project_grads(x, W1, b1, W2, dZ))
    os.environ["DL_PROJECT_BWD"] = "native"
    flops = 2.0 * N * K * ((2 * nhid * F + 2 * nhid * d) if two else d * F)
    print(f"N={N} F={F} K={K} nhid={nhid} d={d}: native {t_native*1e6:8.1f} us ({flops/t_native/1e12:5.1f} TFLOP/s)   "
          f"library {t_lib*1e6:8.1f} us", flush=True)
