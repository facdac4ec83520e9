#!/usr/bin/env python3
"""A/B timings of the forward entry points under environment switches, interleaved in ONE process.

    python tools/ab_times.py --workload squirrel --env DL_AGG_UNROLL=1 DL_AGG_UNROLL=4 [--seg-len 32 64] [--calls 50]

Every (variant, entry point) is timed as `calls` back-to-back launches between two HIP events (the queue stays full, so
host launch gaps do not count), repeated over interleaved rounds; the median per call is printed.
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
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--calls", type=int, default=50)
    ap.add_argument("--workload", default="squirrel")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--K", type=int, default=8)
    ap.add_argument("--d", type=int, default=64)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--env", nargs="*", default=[""], help="variants: NAME=VALUE[,NAME=VALUE...] ('' = defaults)")
    ap.add_argument("--seg-len", nargs="*", type=int, default=[32])
    ap.add_argument("--only", default="route,aggregate,score")
    args = ap.parse_args()
    import bench
    from disenlink_amd import ops
    from disenlink_amd.graph import Graph
    dev = torch.device("cuda:0")
    wb = 2 if args.dtype == "bf16" else 4
    sg, split, graph0, pairs, model, x, Z = bench.build_workload(args.workload, dev, args.K, args.d, 512, scale=args.scale,
                                                                 elem_bytes=wb)
    if args.dtype == "bf16":
        Z = Z.to(torch.bfloat16)
    beta, t = 0.5, 1.0
    graphs = {sl: Graph.from_edge_rows(torch.from_numpy(split.train_src).to(dev), torch.from_numpy(split.train_dst).to(dev),
                                       sg.n_nodes, seg_len=sl) for sl in args.seg_len}
    variants = [(sl, ev) for sl in args.seg_len for ev in args.env]
    res = {v: {k: [] for k in args.only.split(",")} for v in variants}
    ref = {}

    def setenv(ev):
        for kv in filter(None, ev.split(",")):
            k, v = kv.split("=")
            os.environ[k] = v

    def clearenv(ev):
        for kv in filter(None, ev.split(",")):
            os.environ.pop(kv.split("=")[0], None)

    for r in range(args.rounds + 1):
        for (sl, ev) in variants:
            g = graphs[sl]
            setenv(ev)
            p, a, s = ops.route_fwd(g, Z, t)
            H = ops.aggregate_fwd(g, Z, beta, p, a, s)
            fns = {"route": lambda: ops.route_fwd(g, Z, t),
                   "aggregate": lambda: ops.aggregate_fwd(g, Z, beta, p, a, s),
                   "score": lambda: ops.score_pairs_fwd(Z, H, pairs.pu, pairs.pv, t, pairs)}
            if r == 0:                                         # results must not depend on the variant
                key = sl
                if key in ref:
                    assert torch.equal(ref[key][0], H) and torch.equal(ref[key][1], s), (sl, ev)
                ref.setdefault(key, (H.clone(), s.clone()))
            for name in res[(sl, ev)]:
                fn = fns[name]
                fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.calls):
                    fn()
                e1.record()
                e1.synchronize()
                if r > 0:
                    res[(sl, ev)][name].append(e0.elapsed_time(e1) / args.calls * 1e3)
            clearenv(ev)
    for (sl, ev), d in res.items():
        line = "  ".join(f"{k} {np.median(v):8.1f} us" for k, v in d.items())
        print(f"seg_len {sl:3d} {ev or '(default)':28s} {line}   n_seg {graphs[sl].n_seg} slots {graphs[sl].plan.n_slots}")


if __name__ == "__main__":
    main()
