"""Stress of the cross-XCD hand-off (publish_unit_and_sum_row: sc1 stores, agent-scope counter, sc1 loads) behind the
aggregation's and the backward phase 2's in-launch row sums: random graphs with hub rows of 2 .. ~300 units, several (K, d,
table type), each compared BITWISE with the combine launch (DL_INKERNEL_COMBINE=0), repeated launches back to back, with a
second stream keeping the memory system busy in half of the rounds.  One process, one pass; prints a summary line.
usage: python tools/inlaunch_sum_stress.py [rounds=60]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disenlink_amd import _lib, ops
from disenlink_amd.graph import Graph

DEV = "cuda"
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(2026)
shapes = [(8, 64, torch.float32), (16, 128, torch.bfloat16), (4, 32, torch.float32), (8, 64, torch.bfloat16), (16, 128, torch.float32), (5, 64, torch.float32)]


def set_mode(m):
    os.environ["DL_INKERNEL_COMBINE"] = str(m)
    _lib.config_reload()


side = torch.cuda.Stream()
noise = torch.empty(64 << 20, dtype=torch.float32, device=DEV)
launches = mismatches = 0
t0 = time.time()
for r in range(rounds):
    K, d, dtype = shapes[r % len(shapes)]
    N = int(rng.integers(300, 60000))
    E = int(rng.integers(N, 12 * N))
    n_hubs = int(rng.integers(1, 9))
    hubs = [(int(rng.integers(0, N)), int(rng.integers(130, min(80000, 40 * N)))) for _ in range(n_hubs)]
    src = np.concatenate([rng.integers(0, N, E)] + [np.full(n, h) for h, n in hubs])
    dst = np.concatenate([rng.integers(0, N, E)] + [rng.integers(0, N, n) for _h, n in hubs])
    wb = 4 if dtype == torch.float32 else 2
    G = Graph.from_edge_rows(torch.from_numpy(src), torch.from_numpy(dst), N, row_bytes=K * d * wb).to(DEV)
    Z = (torch.randn(N, K, d, generator=torch.Generator().manual_seed(r)) * 0.4).to(DEV)
    Zt = Z if dtype == torch.float32 else Z.to(dtype)
    p, a, s = ops.route_fwd(G, Zt, 1.0)
    dH = torch.randn(N, K, d, generator=torch.Generator().manual_seed(r + 1)).to(DEV)
    set_mode(0)
    ref = ops.aggregate_fwd(G, Zt, 0.55, p, a, s)
    ref_b = ops.route_aggregate_bwd(G, Zt, 0.55, 1.0, p, a, s, dH)
    for mode in (2, 1):
        set_mode(mode)
        for rep in range(6):
            if r % 2 and rep % 2:
                with torch.cuda.stream(side):                  # memory traffic beside the launch under test
                    noise.mul_(1.0001)
            got = ops.aggregate_fwd(G, Zt, 0.55, p, a, s)
            got_b = ops.route_aggregate_bwd(G, Zt, 0.55, 1.0, p, a, s, dH)
            launches += 2
            if not torch.equal(got, ref):
                mismatches += 1
                print(f"MISMATCH aggregate round {r} mode {mode} rep {rep}: N={N} K={K} d={d} {dtype} max|d|={float((got.float() - ref.float()).abs().max()):.3e}", flush=True)
            if not torch.equal(got_b, ref_b):
                mismatches += 1
                print(f"MISMATCH phase 2 round {r} mode {mode} rep {rep}: N={N} K={K} d={d} {dtype}", flush=True)
        if int(G.plan.unit_count.abs().sum()) != 0:
            mismatches += 1
            print(f"COUNTERS not back at zero: round {r} mode {mode}", flush=True)
    if r % 10 == 9:
        print(f"round {r + 1}/{rounds}: {launches} launches compared, {mismatches} mismatches, multi-unit rows of the last graph {int(G.plan.multi_row.numel())}, "
              f"slots {int(G.plan.n_slots)} ({time.time() - t0:.0f} s)", flush=True)
torch.cuda.synchronize()
os.environ.pop("DL_INKERNEL_COMBINE", None)
_lib.config_reload()
print(f"in-launch row sums: {rounds} random graphs, {launches} launches compared bitwise with the combine launch, {mismatches} mismatches")
sys.exit(1 if mismatches else 0)
