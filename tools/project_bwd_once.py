"""A few dl_project_bwd calls at one shape, for rocprofv3 --kernel-trace --stats.
usage: python tools/project_bwd_once.py N F K nhid d"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disenlink_amd import ops  # noqa: E402

N, F, K, nhid, d = (int(v) for v in sys.argv[1:6])
dev = torch.device("cuda:0")
two = nhid > 1
x, dZ = torch.randn(N, F, device=dev), torch.randn(N, K, d, device=dev)
W1 = torch.randn(K, nhid if two else d, F, device=dev) / F ** 0.5
b1 = torch.randn(K, nhid if two else d, device=dev) * 0.1
W2 = torch.randn(K, d, nhid, device=dev) / nhid ** 0.5 if two else None
b2 = torch.zeros(K, d, device=dev) if two else None
for _ in range(10):
    ops.project_bwd(x, W1, b1, W2, dZ)                      # recompute form
    if two:
        Z, hid = ops.project_fwd(x, W1, b1, W2, b2, keep_hid=True)
        ops.project_bwd(x, W1, b1, W2, dZ, hid=hid)         # kept-hidden form
    else:
        ops.project_fwd(x, W1, b1)
torch.cuda.synchronize()
