#!/usr/bin/env python3
"""Per-entry-point timings on the bench workload, interleaved rounds in ONE process (median / min).

    python tools/kernel_times.py [--rounds 20] [--workload squirrel] [--seg-len 32] [--run-len 64]

Used for A/B work: build a variant (DL_CXXFLAGS="-DX=1" python -m disenlink_amd.build --force), run this,
rebuild the other variant, run again — in the same gpurun call, so both arms see the same device.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=20)
    ap.add_argument("--workload", default="squirrel")
    ap.add_argument("--K", type=int, default=8)
    ap.add_argument("--d", type=int, default=64)
    ap.add_argument("--seg-len", type=int, default=32)
    ap.add_argument("--run-len", type=int, default=64)
    ap.add_argument("--inc-seg-len", type=int, default=32)
    ap.add_argument("--slices", type=int, default=8)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--tag", default="")
    args = ap.parse_args()
    import bench
    from disenlink_amd import ops
    from disenlink_amd.graph import Graph, PairList
    dev = torch.device("cuda:0")
    sg, split, graph, pairs, model, x, Z = bench.build_workload(args.workload, dev, args.K, args.d, 512, scale=args.scale)
    graph = Graph.from_edge_rows(torch.from_numpy(split.train_src).to(dev), torch.from_numpy(split.train_dst).to(dev),
                                 sg.n_nodes, seg_len=args.seg_len)
    pairs = PairList.build(pairs.pu, pairs.pv, sg.n_nodes, seg_len=args.inc_seg_len, run_len=args.run_len,
                           n_slices=args.slices)
    beta, t = 0.5, 1.0
    P = pairs.n_pairs
    gp = torch.full((P,), 1.0 / P, device=dev)
    st = {}
    st["p"], st["a"], st["s"] = ops.route_fwd(graph, Z, t)
    st["H"] = ops.aggregate_fwd(graph, Z, beta, st["p"], st["a"], st["s"])
    st["prob"], st["coef"] = ops.score_pairs_fwd(Z, st["H"], pairs.pu, pairs.pv, t, pairs, want_coef=True)
    st["dZs"], st["dH"] = ops.score_pairs_bwd(Z, st["H"], pairs, t, st["prob"], gp, coef=st["coef"])
    st["ds"] = torch.empty_like(st["s"])
    st["dw"], st["dwr"] = ops.route_aggregate_bwd_phase1(graph, Z, beta, st["p"], st["a"], st["s"], st["dH"], st["ds"])
    dZ = torch.empty_like(Z)
    fns = {
        "route": lambda: ops.route_fwd(graph, Z, t),
        "aggregate": lambda: ops.aggregate_fwd(graph, Z, beta, st["p"], st["a"], st["s"]),
        "score_fwd": lambda: ops.score_pairs_fwd(Z, st["H"], pairs.pu, pairs.pv, t, pairs),
        "score_fwd_coef": lambda: ops.score_pairs_fwd(Z, st["H"], pairs.pu, pairs.pv, t, pairs, want_coef=True),
        "score_bwd_coef": lambda: ops.score_pairs_bwd(Z, st["H"], pairs, t, st["prob"], gp, coef=st["coef"]),
        "score_bwd_recompute": lambda: ops.score_pairs_bwd(Z, st["H"], pairs, t, st["prob"], gp),
        "bwd_phase1": lambda: ops.route_aggregate_bwd_phase1(graph, Z, beta, st["p"], st["a"], st["s"], st["dH"], st["ds"]),
        "bwd_phase2": lambda: ops.route_aggregate_bwd_phase2(graph, Z, beta, t, st["p"], st["a"], st["s"], st["dH"],
                                                             st["dw"], st["dwr"], st["ds"], dZ, False),
    }
    times = {k: [] for k in fns}
    for r in range(args.rounds + 2):
        for k, fn in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            if r >= 2:
                times[k].append(e0.elapsed_time(e1) * 1e3)
    out = {k: dict(median_us=float(np.median(v)), min_us=float(np.min(v))) for k, v in times.items()}
    fwd = out["route"]["median_us"] + out["aggregate"]["median_us"] + out["score_fwd"]["median_us"]
    print(json.dumps(dict(tag=args.tag, E=graph.n_edges, P=P, n_seg=graph.n_seg, fwd_us=fwd, kernels=out)))
    for k, v in out.items():
        print(f"  {k:22s} median {v['median_us']:9.1f} us   min {v['min_us']:9.1f} us", file=sys.stderr)


if __name__ == "__main__":
    main()
