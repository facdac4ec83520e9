"""Projection backward, both forms (recompute / from the kept hidden layer), kernel time only — for A/B runs with DL_LIB_PATH.
usage: python tools/project_bwd_quick.py [N F K nhid d]..."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disenlink_amd import ops
shapes = [(5201, 128, 8, 512, 64), (2277, 128, 8, 512, 64), (41554, 128, 16, 256, 128), (5201, 2088, 8, 512, 64)]
if len(sys.argv) > 5:
    v = [int(a) for a in sys.argv[1:]]
    shapes = [tuple(v[i:i + 5]) for i in range(0, len(v) - 4, 5)]
for (N, F, K, nhid, d) in shapes:
    x = torch.randn(N, F, device="cuda"); dZ = torch.randn(N, K, d, device="cuda")
    W1 = torch.randn(K, nhid, F, device="cuda") / F ** 0.5; b1 = torch.randn(K, nhid, device="cuda") * 0.1
    W2 = torch.randn(K, d, nhid, device="cuda") / nhid ** 0.5; b2 = torch.randn(K, d, device="cuda") * 0.1
    _Z, hid = ops.project_fwd(x, W1, b1, W2, b2, keep_hid=True)
    out = []
    for name, fn in (("recompute", lambda: ops.project_bwd(x, W1, b1, W2, dZ)), ("kept", lambda: ops.project_bwd(x, W1, b1, W2, dZ, hid=hid))):
        ts = []
        for r in range(14):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); e1.synchronize()
            if r >= 2: ts.append(e0.elapsed_time(e1) * 1e3)
        out.append(f"{name} {np.median(ts):8.1f} us")
    print(f"{os.environ.get('DL_LIB_PATH', 'default')[-26:]:>26s} N={N} F={F} K={K} nhid={nhid} d={d}: " + "   ".join(out), flush=True)
