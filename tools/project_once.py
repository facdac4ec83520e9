#!/usr/bin/env python3
"""Run the projection forward a few dozen times at one shape (for rocprofv3 --kernel-trace --stats):
    python tools/project_once.py N F K nhid d [reps]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from disenlink_amd import ops

N, F, K, nhid, d = (int(v) for v in sys.argv[1:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 30
torch.manual_seed(0)
x = torch.randn(N, F, device="cuda")
W1 = torch.randn(K, nhid, F, device="cuda") / F ** 0.5
b1 = torch.randn(K, nhid, device="cuda") * 0.1
W2 = torch.randn(K, d, nhid, device="cuda") / nhid ** 0.5
b2 = torch.randn(K, d, device="cuda") * 0.1
for _ in range(reps):
    Z = ops.project_fwd(x, W1, b1, W2, b2)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    Z = ops.project_fwd(x, W1, b1, W2, b2)
e1.record(); e1.synchronize()
flop = 2.0 * N * F * K * nhid + 2.0 * N * K * nhid * d
us = e0.elapsed_time(e1) * 1e3 / reps
print(f"N={N} F={F} K={K} nhid={nhid} d={d}: {us:.1f} us per call, {flop / us / 1e6:.1f} TF/s fp32-equivalent")
