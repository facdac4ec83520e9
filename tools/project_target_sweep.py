"""Sweep of the backward's workgroups-per-launch knob (DL_BWD_TARGET, read per call) for the kept-hidden form.
usage: python tools/project_target_sweep.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disenlink_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for N, F, K, nhid, d in [(5201, 128, 8, 512, 64), (5201, 512, 8, 512, 64), (5201, 2089, 8, 512, 64), (41554, 128, 16, 512, 128)]:
    x, dZ = torch.randn(N, F, device=dev), torch.randn(N, K, d, device=dev)
    W1 = torch.randn(K, nhid, F, device=dev) / F ** 0.5; b1 = torch.randn(K, nhid, device=dev) * 0.1
    W2 = torch.randn(K, d, nhid, device=dev) / nhid ** 0.5; b2 = torch.zeros(K, d, device=dev)
    _Z, hid = ops.project_fwd(x, W1, b1, W2, b2, keep_hid=True)
    out = []
    for tgt in (None, 128, 256, 384, 512, 768, 1024, 1536, 2048, 3072):
        if tgt is None: os.environ.pop("DL_BWD_TARGET", None)
        else: os.environ["DL_BWD_TARGET"] = str(tgt)
        out.append(f"{'auto' if tgt is None else tgt}:{timeit(lambda: ops.project_bwd(x, W1, b1, W2, dZ, hid=hid)) * 1e6:.0f}")
    os.environ.pop("DL_BWD_TARGET", None)
    print(f"N={N} F={F} K={K} nhid={nhid} d={d} bwd(kept) us  " + "  ".join(out), flush=True)
