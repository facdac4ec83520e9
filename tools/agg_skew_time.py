#!/usr/bin/env python3
"""Aggregation kernel time against the skew of the routing: factor 0 of Z scaled up so that a growing share of the edges is
routed to it (the class-owned accumulators walk max-class-size steps).  usage (GPU box): python tools/agg_skew_time.py"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from disenlink_amd import ops
dev = torch.device("cuda:0")
sg, split, graph, pairs, model, x, Z = bench.build_workload(sys.argv[1] if len(sys.argv) > 1 else "squirrel_real", dev, 8, 64, 512)
for scale in (1.0, 1.5, 2.0, 3.0, 6.0):
    Zs = Z.clone()
    Zs[:, 0, :] *= scale
    p, a, s = ops.route_fwd(graph, Zs, 1.0)
    share = float((p == 0).float().mean())
    for _ in range(3):
        ops.aggregate_fwd(graph, Zs, 0.5, p, a, s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.aggregate_fwd(graph, Zs, 0.5, p, a, s)
    e1.record(); e1.synchronize()
    print(f"factor-0 scale {scale}: {share * 100:5.1f} % of the edges routed to factor 0, aggregate phase {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
