"""Scorer forward+backward on the bench workload: stored per-factor terms vs recompute in the backward."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from disenlink_amd import ops
dev = torch.device("cuda:0")
sg, split, graph, pairs, model, x, Z = bench.build_workload(sys.argv[1] if len(sys.argv) > 1 else "squirrel", dev, 8, 64, 512)
t, beta = 1.0, 0.5
p, a, s = ops.route_fwd(graph, Z, t)
H = ops.aggregate_fwd(graph, Z, beta, p, a, s)
P = pairs.n_pairs
gp = torch.full((P,), 1.0 / P, device=dev)
def stored():
    prob, coef = ops.score_pairs_fwd(Z, H, pairs.pu, pairs.pv, t, pairs, want_coef=True)
    return ops.score_pairs_bwd(Z, H, pairs, t, prob, gp, coef=coef)
def recompute():
    prob = ops.score_pairs_fwd(Z, H, pairs.pu, pairs.pv, t, pairs)
    return ops.score_pairs_bwd(Z, H, pairs, t, prob, gp)
for name, fn in (("stored terms", stored), ("recompute", recompute), ("stored terms", stored), ("recompute", recompute)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize()
    print(f"{name}: scorer fwd+bwd {(time.perf_counter() - t0) / 20 * 1e6:.1f} us", flush=True)
a1, b1 = stored(); a2, b2 = recompute()
print("max|dZ diff|", float((a1 - a2).abs().max()), "max|dH diff|", float((b1 - b2).abs().max()))
