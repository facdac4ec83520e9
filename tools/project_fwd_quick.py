"""Forward projection kernel time only (no library comparison, no accuracy check): for kernel experiments
with DL_LIB_PATH.  usage: python tools/project_fwd_quick.py [N F K nhid d]..."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disenlink_amd import ops
shapes = [(5201, 2088, 8, 512, 64), (5201, 2089, 8, 512, 64), (41554, 128, 16, 256, 128), (5201, 128, 8, 512, 64)]
if len(sys.argv) > 5:
    v = [int(a) for a in sys.argv[1:]]
    shapes = [tuple(v[i:i + 5]) for i in range(0, len(v) - 4, 5)]
for (N, F, K, nhid, d) in shapes:
    x = torch.randn(N, F, device="cuda")
    W1 = torch.randn(K, nhid, F, device="cuda") / F ** 0.5
    b1 = torch.randn(K, nhid, device="cuda") * 0.1
    W2 = torch.randn(K, d, nhid, device="cuda") / nhid ** 0.5
    b2 = torch.randn(K, d, device="cuda") * 0.1
    ts = []
    for r in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.project_fwd(x, W1, b1, W2, b2); e1.record(); e1.synchronize()
        if r >= 2: ts.append(e0.elapsed_time(e1) * 1e3)
    flop = 2.0 * N * F * K * nhid + 2.0 * N * K * nhid * d
    print(f"{os.environ.get('DL_LIB_PATH', 'default')[-24:]:>24s} N={N} F={F} K={K} nhid={nhid} d={d}: {np.median(ts):9.1f} us  {flop / np.median(ts) / 1e6:6.1f} TF/s", flush=True)
