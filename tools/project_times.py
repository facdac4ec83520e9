#!/usr/bin/env python3
"""Projection: fused MFMA kernel vs the library two-GEMM path, same process (median of interleaved rounds)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from disenlink_amd import ops
from disenlink_amd.model import Disentangle

def lib_path(m, x):
    fs = m.factors; K, d = m.nfactor, m.nebed
    W1 = torch.cat([f.mlp1.weight for f in fs], 0); b1 = torch.cat([f.mlp1.bias for f in fs], 0)
    hid = torch.relu(torch.nn.functional.linear(x, W1, b1)).view(x.shape[0], K, -1)
    W2 = torch.stack([f.mlp2.weight for f in fs], 0); b2 = torch.stack([f.mlp2.bias for f in fs], 0)
    return (torch.einsum("nkh,kdh->nkd", hid, W2) + b2).contiguous()

for (N, F, K, nhid, d) in [(5201, 128, 8, 512, 64), (2277, 128, 8, 512, 64), (41554, 128, 16, 256, 128), (5201, 2089, 8, 512, 64)]:
    torch.manual_seed(0)
    m = Disentangle(F, nhid, d, nfactor=K, beta=0.5, t=1, projection="mfma").cuda()
    x = torch.randn(N, F, device="cuda")
    with torch.no_grad():
        fns = {"mfma": lambda: m.project(x), "library": lambda: lib_path(m, x)}
        t = {k: [] for k in fns}
        for r in range(12):
            for k, fn in fns.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(); e1.record(); e1.synchronize()
                if r >= 2: t[k].append(e0.elapsed_time(e1) * 1e3)
        err = float((m.project(x) - lib_path(m, x)).abs().max())
    flop = 2.0 * N * F * K * nhid + 2.0 * N * K * nhid * d
    print(f"N={N} F={F} K={K} nhid={nhid} d={d}: " + "  ".join(f"{k} {np.median(v):8.1f} us ({flop / np.median(v) / 1e6:6.1f} TF/s)" for k, v in t.items()) + f"  max|diff| {err:.2e}")
