#!/usr/bin/env python3
"""Projection forward: fused MFMA kernel vs the library two-GEMM path on pre-stacked weights
(so that only the projection itself is timed), same process, median of interleaved rounds."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from disenlink_amd import ops

for (N, F, K, nhid, d) in [(5201, 128, 8, 512, 64), (2277, 128, 8, 512, 64), (41554, 128, 16, 256, 128), (5201, 2089, 8, 512, 64), (5201, 2088, 8, 512, 64), (2277, 2325, 5, 512, 32), (41554, 4814, 16, 512, 128)]:
    torch.manual_seed(0)
    x = torch.randn(N, F, device="cuda")
    W1 = torch.randn(K, nhid, F, device="cuda") / F ** 0.5
    b1 = torch.randn(K, nhid, device="cuda") * 0.1
    W2 = torch.randn(K, d, nhid, device="cuda") / nhid ** 0.5
    b2 = torch.randn(K, d, device="cuda") * 0.1
    W1c, b1c = W1.reshape(K * nhid, F), b1.reshape(-1)

    def library():
        hid = torch.relu(torch.nn.functional.linear(x, W1c, b1c)).view(N, K, nhid)
        return torch.baddbmm(b2.unsqueeze(1), hid.transpose(0, 1), W2.transpose(1, 2)).transpose(0, 1).contiguous()

    fns = {"mfma": lambda: ops.project_fwd(x, W1, b1, W2, b2), "library": library}
    t = {k: [] for k in fns}
    for r in range(14):
        for k, fn in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); e1.synchronize()
            if r >= 2: t[k].append(e0.elapsed_time(e1) * 1e3)
    err = float((fns["mfma"]() - library()).abs().max())
    flop = 2.0 * N * F * K * nhid + 2.0 * N * K * nhid * d
    print(f"N={N} F={F} K={K} nhid={nhid} d={d}: " + "  ".join(f"{k} {np.median(v):8.1f} us ({flop / np.median(v) / 1e6:6.1f} TF/s)" for k, v in t.items()) + f"  max|diff| {err:.2e}")
