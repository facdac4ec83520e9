"""Forward step (route, aggregate, score) with the rows of several units summed inside the launch against the separate
combine launches (DL_INKERNEL_COMBINE=0), same process and plans: same bits, time per phase.
usage: python tools/edge_scatter_ab.py [workload] [K] [d] [scale]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from disenlink_amd import _lib, ops
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "squirrel_real"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d = int(sys.argv[3]) if len(sys.argv) > 3 else 64
scale = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
sg, split, graph, pairs, model, x, Z = bench.build_workload(name, dev, K, d, 512, scale=scale)
t, beta = 1.0, 0.5
p, a, s = ops.route_fwd(graph, Z, t)

def mode(on):
    os.environ["DL_INKERNEL_COMBINE"] = "2" if on else "0"
    _lib.config_reload()

def timed(fn, n=50):
    for _ in range(10): fn()
    best = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); e1.synchronize()
        best.append(e0.elapsed_time(e1) / n * 1e3)
    return float(np.median(best))

agg = lambda: ops.aggregate_fwd(graph, Z, beta, p, a, s)
def step():
    pp, aa, ss = ops.route_fwd(graph, Z, t)
    H = ops.aggregate_fwd(graph, Z, beta, pp, aa, ss)
    return ops.score_pairs_fwd(Z, H, pairs.pu, pairs.pv, t, pairs)
res = {}
mode(False); ref = agg().clone()
for rnd in range(2):
    for on in (False, True):
        mode(on)
        same = torch.equal(agg(), ref)
        res.setdefault(on, []).append((timed(agg, 50 if sg.n_nodes < 100000 else 8), timed(step, 20 if sg.n_nodes < 100000 else 3), same))
mode(False)
print(f"{name} K={K} d={d}: rows of several units {int(graph.plan.multi_row.numel())}, partial slots {graph.plan.n_slots}")
for on in (False, True):
    print(f"  {'in-launch row sums' if on else 'separate combine   '}: aggregate " + " / ".join(f"{r[0]:.1f}" for r in res[on]) +
          " us, forward step " + " / ".join(f"{r[1]:.1f}" for r in res[on]) + f" us, same bits as the separate form: {all(r[2] for r in res[on])}")
