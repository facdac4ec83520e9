"""Forward scorer against the column slicing and the run length of the pairs-by-first-endpoint plan, interleaved in one
process.  usage: python tools/fwd_slices_sweep.py [workload] [K] [d] [slices,..] [run_len,..]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from disenlink_amd import ops
from disenlink_amd.graph import PairList
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "squirrel_real"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d = int(sys.argv[3]) if len(sys.argv) > 3 else 64
slices = [int(v) for v in (sys.argv[4] if len(sys.argv) > 4 else "4,8,16").split(",")]
runs = [int(v) for v in (sys.argv[5] if len(sys.argv) > 5 else "64").split(",")]
sg, split, graph, pairs, model, x, Z = bench.build_workload(name, dev, K, d, 512)
H = ops.aggregate_fwd(graph, Z, 0.5, *ops.route_fwd(graph, Z, 1.0))
plans = {(s, r): PairList.build(pairs.pu, pairs.pv, sg.n_nodes, run_len=r, n_slices=s, inc_slices=1) for s in slices for r in runs}
ref = ops.score_pairs_fwd(Z, H, pairs.pu, pairs.pv, 1.0, pairs)
times = {k: [] for k in plans}
for rnd in range(10):
    for k in (list(plans) if rnd % 2 == 0 else list(plans)[::-1]):
        pl = plans[k]
        fn = lambda: ops.score_pairs_fwd(Z, H, pl.pu, pl.pv, 1.0, pl)
        out = fn()
        assert torch.equal(out, ref)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); e1.synchronize()
        times[k].append(e0.elapsed_time(e1) / 20 * 1e3)
print(f"{name} K={K} d={d}: default plan slices {pairs.by_u.n_slices}; median / min us over 10 interleaved rounds (same bits checked)")
for k, v in times.items():
    print(f"  slices {k[0]:3d} run_len {k[1]:3d}: {np.median(v):8.1f} / {min(v):8.1f}   segments {plans[k].by_u.n_seg}")
