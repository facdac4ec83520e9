"""Timing bound for a SPLIT forward scorer (round 6, last experiment): prob = sigmoid(sum_k (h_u.h_v)_k exp((z_u.z_v)_k / t)) —
the exp(z.z / t) half does not depend on H, so it could run on a second stream beside [route, row sums, aggregate] and only
the h.h half would wait for H.  Before writing the two half kernels: how long do the edge scatter on one stream and the WHOLE
scorer on another take TOGETHER (the scorer reads the previous step's H — the timing is what counts) against the sequential step?
A split step would cost about  together - scorer + 2 x (half a scorer + its ramp)  ~  together + 10 us."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from disenlink_amd import ops

dev = torch.device("cuda")
K, d = 8, 64
sg, split, graph, pairs, model, x, Z = bench.build_workload("squirrel_real", dev, K, d, 512)
beta, t = 0.5, 1.0
pu, pv = pairs.pu, pairs.pv
H = ops.aggregate_fwd(graph, Z, beta, *ops.route_fwd(graph, Z, t))
sA, sB = torch.cuda.Stream(), torch.cuda.Stream(priority=0)
sHi = torch.cuda.Stream(priority=-1)


def scatter():
    return ops.aggregate_fwd(graph, Z, beta, *ops.route_fwd(graph, Z, t))


def sequential():
    Hn = scatter()
    ops.score_pairs_fwd(Z, Hn, pu, pv, t, pairs=pairs)


def together(stream_scatter):
    cur = torch.cuda.current_stream()
    e0 = torch.cuda.Event(); e0.record(cur)
    stream_scatter.wait_event(e0); sB.wait_event(e0)
    with torch.cuda.stream(stream_scatter):
        scatter()
        ea = torch.cuda.Event(); ea.record(stream_scatter)
    with torch.cuda.stream(sB):
        ops.score_pairs_fwd(Z, H, pu, pv, t, pairs=pairs)
        eb = torch.cuda.Event(); eb.record(sB)
    cur.wait_event(ea); cur.wait_event(eb)


def timed(fn, steps=200, warm=30):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) * 1e3 / steps


for rnd in range(3):
    ts = timed(sequential)
    tsc = timed(lambda: scatter())
    tsco = timed(lambda: ops.score_pairs_fwd(Z, H, pu, pv, t, pairs=pairs))
    tt = timed(lambda: together(sA))
    tth = timed(lambda: together(sHi))
    print(f"round {rnd}: sequential step {ts:6.1f} us | edge scatter alone {tsc:5.1f} | scorer alone {tsco:6.1f} | together {tt:6.1f} "
          f"(edge scatter on a high-priority stream: {tth:6.1f}) -> a split step ~ {min(tt, tth) + 10:6.1f} us", flush=True)
