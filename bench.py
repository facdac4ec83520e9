#!/usr/bin/env python3
"""edges/sec (aggregate+score) at K=8, d=64 — the metric of BASELINE.json — on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload squirrel|chameleon|...]

One "step" = one forward pass of the hot path over the whole training graph with inputs
resident in HBM: route (model.py:56-72) + aggregate (model.py:73-75) over the E_sym directed
non-zeros of the training adjacency, + the pair scorer (model.py:109-113) over the P scored
training pairs.  value = (E_sym + P) * steps / time  (SURVEY.md §8d headline rate).  The
projection GEMM and the one-time CSR construction are outside the timed region, as §8d says.

N > 1 is launched by torch.distributed.run, one rank per GPU (see disenlink_amd/dist.py).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 = 256 FLOP/clk/CU x 256 CUs x 2.4 GHz
# kernel behind each timed phase (the name rocprofv3 reports) -> PMC summary of tools/pmc_traffic.py
# kernels launched by each phase of the step (names as in the rocprofv3 traces); the combine kernels of a phase are
# included at their per-launch average
PHASE_KERNEL = {"route": ("dl::fast::route_seg_kernel<8, 64, float, true>", "dl::fast::s_rowsum_thread_kernel<8>",
                          "dl::fast::vec_combine_kernel"),
                "aggregate": ("dl::fast::aggregate_seg_kernel<8, 64, float>", "dl::fast::row_combine_kernel<512, float, float>"),
                "score": ("dl::fast::score_fwd_seg_kernel<8, 64, float, false>",)}
PMC_SUMMARY = os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")


def pmc_traffic(phase):
    """HBM-side bytes per launch of the phase's kernel from the committed rocprofv3 PMC passes
    (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE; same command, same workload), or None."""
    try:
        table = json.load(open(PMC_SUMMARY))
        return float(sum(table[k]["traffic_bytes"] for k in PHASE_KERNEL[phase]))
    except (OSError, KeyError, ValueError):
        return None


def algorithmic_bytes(K, d, n_nodes, n_edges, n_pairs, w=4):
    """SURVEY.md §8(d), no cache credit: per edge route K*d*w+4+1+4, aggregate d*w+4+1+4+4;
    per node 2*K*d*w (z_i in both passes) + K*d*w (h_i) + 2*K*4 (s) + 8 (rowptr); per pair 4*K*d*w+8+4."""
    route = n_edges * (K * d * w + 9) + n_nodes * (K * d * w + K * 4 + 4)
    aggregate = n_edges * (d * w + 13) + n_nodes * (2 * K * d * w + K * 4 + 4)
    score = n_pairs * (4 * K * d * w + 12)
    return dict(route=route, aggregate=aggregate, score=score)


def build_workload(name, device, K, d, nhid, seed=0, m=5, scale=1.0, elem_bytes=4):
    from disenlink_amd.data import synthetic_graph
    from disenlink_amd.graph import Graph, PairList
    from disenlink_amd.model import Disentangle
    from disenlink_amd.splits import make_link_split
    sg = synthetic_graph(name, seed=seed, scale=scale)
    split = make_link_split(sg.src, sg.dst, sg.n_nodes, m=m, seed=seed)
    graph = Graph.from_edge_rows(torch.from_numpy(split.train_src).to(device),
                                 torch.from_numpy(split.train_dst).to(device), sg.n_nodes, row_bytes=K * d * elem_bytes)
    pu = np.concatenate([split.pos_train.u, split.neg_train.u])
    pv = np.concatenate([split.pos_train.v, split.neg_train.v])
    order = np.lexsort((pv, pu))                      # pair list laid out by u: long runs share the u rows
    pu, pv = pu[order], pv[order]
    pairs = PairList.build(torch.from_numpy(pu).to(device), torch.from_numpy(pv).to(device), sg.n_nodes,
                           row_bytes=K * d * elem_bytes)
    torch.manual_seed(seed)
    model = Disentangle(sg.n_feat, nhid, d, nfactor=K, beta=0.5, t=1).to(device)
    x = torch.from_numpy(sg.features()).to(device)
    with torch.no_grad():
        Z = model.project(x).contiguous()
    return sg, split, graph, pairs, model, x, Z


def cpu_baseline(Z_cpu, graph_cpu, n_units, beta, t, budget_s=30.0):
    """The oracle's dense restatement (same op sequence as model.py:56-76,109-113) timed on the host
    cores: one warm-up, then the median of up to 3 passes within the budget."""
    from oracle import dense_ref
    N = Z_cpu.shape[0]
    adj = torch.zeros(N, N)
    src = torch.repeat_interleave(torch.arange(N), (graph_cpu.rowptr[1:] - graph_cpu.rowptr[:-1]).long())
    adj[src, graph_cpu.col.long()] = 1
    Zk = Z_cpu.permute(1, 0, 2).contiguous()
    times = []
    t_all = time.perf_counter()
    with torch.no_grad():
        for it in range(4):
            t0 = time.perf_counter()
            H, e, _att, _p, _s = dense_ref.route_aggregate(Zk, adj, beta, t)
            P = dense_ref.score_allpairs(H, e)
            dt = time.perf_counter() - t0
            del H, e, _att, P
            if it > 0:
                times.append(dt)
            if time.perf_counter() - t_all > budget_s and times:
                break
    med = float(np.median(times))
    return dict(value=n_units / med, unit="edges/s", cores=torch.get_num_threads(), kind="port",
                sample=f"whole workload, dense [K,N,N] forward (route+aggregate+all-pairs score) of oracle/dense_ref.py, "
                       f"median of {len(times)} after 1 warm-up, {med:.2f} s per pass",
                seconds_per_pass=med)


def cpu_baseline_sparse(Z_cpu, graph_cpu, pairs_cpu, n_units, beta, t, budget_s=30.0):
    """Baseline B: the multi-threaded C restatement of the edge-list form, for graphs whose
    dense [K,N,N] form cannot exist.  Whole workload, median of up to 3 passes after one warm-up."""
    from oracle import c_ref
    Zh = Z_cpu.numpy()
    rowptr, col = graph_cpu.rowptr.numpy(), graph_cpu.col.numpy()
    pu, pv = pairs_cpu[0].numpy(), pairs_cpu[1].numpy()
    times = []
    t_all = time.perf_counter()
    for it in range(4):
        t0 = time.perf_counter()
        p, a, s = c_ref.route(Zh, rowptr, col, t)
        H = c_ref.aggregate(Zh, rowptr, col, p, a, s, beta)
        c_ref.score_pairs(Zh, H, pu, pv, t)
        dt = time.perf_counter() - t0
        if it > 0:
            times.append(dt)
        if time.perf_counter() - t_all > budget_s and times:
            break
    med = float(np.median(times))
    return dict(value=n_units / med, unit="edges/s", cores=os.cpu_count(), kind="port",
                sample=f"whole workload, edge-list forward (route+aggregate+score_pairs) of oracle/c/sparse_ref.c with "
                       f"OpenMP on all host cores (the reference's dense form cannot hold this graph), "
                       f"median of {len(times)} after 1 warm-up, {med:.2f} s per pass",
                seconds_per_pass=med)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="squirrel")
    ap.add_argument("--K", type=int, default=8)
    ap.add_argument("--d", type=int, default=64)
    ap.add_argument("--nhidden", type=int, default=512)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-generic", action="store_true")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="storage type of the gathered Z/H tables (arithmetic is fp32 either way)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # DL_REHEARSE_ON_ONE_GPU=1: every rank uses cuda:0 and the collectives go through gloo — a functional
    # rehearsal of the N>1 code on a one-GPU box (RCCL refuses two ranks on one device); never a measurement.
    rehearse = bool(os.environ.get("DL_REHEARSE_ON_ONE_GPU"))
    dev_index = 0 if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    sharded = world > 1 or bool(os.environ.get("DL_FORCE_SHARDED"))     # the env var rehearses the N>1 code on 1 GPU
    if sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from disenlink_amd import _lib, ops
    lib = _lib.load()
    if args.force_generic:
        lib.dl_set_force_generic(1)

    K, d = args.K, args.d
    beta, t = 0.5, 1.0
    if sharded:
        import torch.distributed as dist
        from disenlink_amd import dist as dl_dist
        result = dl_dist.bench_sharded(args, rank, world, device)
        if rank == 0:
            print(json.dumps(result))
        dist.destroy_process_group()
        return

    sg, split, graph, pairs, model, x, Z = build_workload(args.workload, device, K, d, args.nhidden, scale=args.scale,
                                                          elem_bytes=2 if args.dtype == "bf16" else 4)
    E, P, N = graph.n_edges, pairs.n_pairs, graph.n_nodes
    wbytes = 4
    if args.dtype == "bf16":
        Z, wbytes = Z.to(torch.bfloat16), 2

    def step():
        p, a, s = ops.route_fwd(graph, Z, t)
        H = ops.aggregate_fwd(graph, Z, beta, p, a, s)
        prob = ops.score_pairs_fwd(Z, H, pairs.pu, pairs.pv, t, pairs)
        return p, a, s, H, prob

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0

    # per-kernel durations with HIP events on the launch stream (torch's current stream), same loop
    names = ("route", "aggregate", "score")
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
    torch.cuda.synchronize()
    for i in range(args.steps):
        ev[i][0].record()
        p, a, s = ops.route_fwd(graph, Z, t)
        ev[i][1].record()
        H = ops.aggregate_fwd(graph, Z, beta, p, a, s)
        ev[i][2].record()
        prob = ops.score_pairs_fwd(Z, H, pairs.pu, pairs.pv, t, pairs)
        ev[i][3].record()
    torch.cuda.synchronize()
    ktime = {n: float(np.mean([ev[i][j].elapsed_time(ev[i][j + 1]) for i in range(args.steps)])) * 1e-3
             for j, n in enumerate(names)}
    abytes = algorithmic_bytes(K, d, N, E, P, w=wbytes)
    kernels = {n: dict(avg_us=ktime[n] * 1e6, algorithmic_bytes=abytes[n],
                       achieved_GBs=abytes[n] / ktime[n] / 1e9, frac=abytes[n] / ktime[n] / 1e9 / HBM_PEAK_GBS)
               for n in names}
    kernels["score"]["pairs_per_s"] = P / ktime["score"]                  # P / t_score
    default_case = args.workload == "squirrel" and K == 8 and d == 64 and args.dtype == "f32" and args.scale == 1.0
    for n in names:                                                       # measured fabric-side bytes beside the algorithmic ones
        kernels[n]["traffic"] = pmc_traffic(n) if default_case else None
    dom = max(names, key=lambda n: ktime[n])
    scatter_t = ktime["route"] + ktime["aggregate"]
    scatter_b = abytes["route"] + abytes["aggregate"]

    # extra: forward+backward of the same path (what one training epoch adds on top), not the headline
    gp = torch.full((P,), 1.0 / P, device=device)
    def train_step():
        p, a, s = ops.route_fwd(graph, Z, t)
        H = ops.aggregate_fwd(graph, Z, beta, p, a, s)
        prob, coef = ops.score_pairs_fwd(Z, H, pairs.pu, pairs.pv, t, pairs, want_coef=True)
        dZs, dH = ops.score_pairs_bwd(Z, H, pairs, t, prob, gp, coef=coef)
        return ops.route_aggregate_bwd(graph, Z, beta, t, p, a, s, dH, dZ_accum=dZs)
    for _ in range(3):
        train_step()
    torch.cuda.synchronize()
    nb = max(5, args.steps // 4)
    t0 = time.perf_counter()
    for _ in range(nb):
        train_step()
    torch.cuda.synchronize()
    fb_ms = (time.perf_counter() - t0) / nb * 1e3

    # extra: the scorer's training step in its two forms: forward storing terms + two backward passes,
    # and dl_score_pairs_train (forward, loss gradient and backward in one pass; the default for tables that live in HBM)
    scorer_train = None
    if args.dtype == "f32":
        from disenlink_amd.metrics import pair_bce_weights
        Hs = ops.aggregate_fwd(graph, Z, beta, *ops.route_fwd(graph, Z, t))
        yb = (torch.rand(P, device=device) < 0.2).float()
        wb = pair_bce_weights(P // 6, P - P // 6, 5, device)
        def separate():
            prob, coef = ops.score_pairs_fwd(Z, Hs, pairs.pu, pairs.pv, t, pairs, want_coef=True)
            return ops.score_pairs_bwd(Z, Hs, pairs, t, prob, gp, coef=coef)
        def one_pass():
            return ops.score_pairs_train(Z, Hs, pairs, t, yb, wb)
        times = {}
        for name, fn in (("separate_us", separate), ("one_pass_us", one_pass)):
            if name == "one_pass_us" and not ops.score_pairs_train_supported(pairs, K, d, ops._lib.DL_F32):
                continue
            for _ in range(2):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            e1.synchronize()
            times[name] = e0.elapsed_time(e1) / 5 * 1e3
        scorer_train = {**times, "pairs": P, "note": "separate = dl_score_pairs_fwd(coef) + dl_score_pairs_bwd; one pass = dl_score_pairs_train"}

    # extra: the projection (excluded from the headline) — the path's only MFMA-bound kernels.
    # fp32 in / fp32 results, priced against the fp32 matrix peak (v_mfma_f32_32x32x2_f32: 256 FLOP/clk/CU); layer 1
    # and the dW1 contraction run as six exact bf16 products per term from three bf16 planes per operand (fp32-grade
    # accuracy), so "achieved" is fp32-equivalent FLOP/s.
    proj = None
    if ops.project_supported(d) and not model.single_layer:
        st = model._stacked
        W1, b1 = st[("mlp1", "weight")], st[("mlp1", "bias")]
        W2, b2 = st[("mlp2", "weight")], st[("mlp2", "bias")]
        gZ = torch.randn(N, K, d, device=device)
        def ev_time(fn, reps=10):
            for _ in range(2):
                fn()
            e = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
            e[0].record()
            for i in range(reps):
                fn()
                e[i + 1].record()
            torch.cuda.synchronize()
            return float(np.mean([e[i].elapsed_time(e[i + 1]) for i in range(reps)])) * 1e-3
        Fx, nh = x.shape[1], W1.shape[1]
        t_f = ev_time(lambda: ops.project_fwd(x, W1, b1, W2, b2))
        t_b = ev_time(lambda: ops.project_bwd(x, W1, b1, W2, gZ))
        _Zk, hid_kept = ops.project_fwd(x, W1, b1, W2, b2, keep_hid=True)
        t_fk = ev_time(lambda: ops.project_fwd(x, W1, b1, W2, b2, keep_hid=True))
        t_bk = ev_time(lambda: ops.project_bwd(x, W1, b1, W2, gZ, hid=hid_kept))
        fl_f = 2.0 * N * K * nh * (Fx + d)
        fl_b = 2.0 * N * K * nh * (2 * Fx + 2 * d)              # recomputed layer 1, dW1, dhid, dW2
        fl_bk = 2.0 * N * K * nh * (Fx + 2 * d)                 # kept hidden layer: dW1, dhid, dW2
        proj = {"bound": "mfma", "dtype": "f32" if os.environ.get("DL_PROJECT_FP32_MFMA") else
                "f32 results; layer 1 and dW1 from three bf16 planes per operand (six exact products per term)",
                "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s (fp32-equivalent, against the fp32 matrix peak)",
                "shape": {"N": N, "F": Fx, "K": K, "nhid": nh, "d": d},
                "fwd": {"avg_us": t_f * 1e6, "achieved": fl_f / t_f / 1e12, "frac": fl_f / t_f / 1e12 / FP32_MFMA_PEAK_TFLOPS},
                "bwd": {"avg_us": t_b * 1e6, "achieved": fl_b / t_b / 1e12, "frac": fl_b / t_b / 1e12 / FP32_MFMA_PEAK_TFLOPS},
                "fwd_keeping_hidden": {"avg_us": t_fk * 1e6},
                "bwd_from_kept_hidden": {"avg_us": t_bk * 1e6, "achieved": fl_bk / t_bk / 1e12,
                                         "frac": fl_bk / t_bk / 1e12 / FP32_MFMA_PEAK_TFLOPS}}

    # extra: the dense [N,N] scorer of the drop-in forward (model.py:109-113 as written): Gram products on MFMA
    dense = None
    if N <= 12000 and args.dtype == "f32":
        Hd = ops.aggregate_fwd(graph, Z, beta, *ops.route_fwd(graph, Z, t))
        for _ in range(2):
            ops.score_allpairs_fwd(Z, Hd, t)
        evd = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        evd[0].record()
        for i in range(5):
            ops.score_allpairs_fwd(Z, Hd, t)
            evd[i + 1].record()
        torch.cuda.synchronize()
        t_d = float(np.mean([evd[i].elapsed_time(evd[i + 1]) for i in range(5)])) * 1e-3
        nt = (N + 127) // 128                                   # 128x128 tiles; only u tile <= v tile is computed
        fl_d = 4.0 * K * d * 128 * 128 * (nt * (nt + 1) // 2)   # MFMA flops executed (P is symmetric: half mirrored)
        dense = {"bound": "mfma", "dtype": "f32 results from three bf16 planes per operand (six exact products per term)"
                 if not os.environ.get("DL_DENSE_FP32_MFMA") else "f32", "peak": FP32_MFMA_PEAK_TFLOPS,
                 "unit": "TFLOP/s (fp32-equivalent, against the fp32 matrix peak)", "pairs": N * N,
                 "avg_us": t_d * 1e6, "achieved": fl_d / t_d / 1e12, "frac": fl_d / t_d / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                 "effective": 4.0 * N * N * K * d / t_d / 1e12, "pairs_per_s": N * N / t_d}

    units = E + P
    result = {
        "metric": "edges/sec (aggregate+score) at K=8 d=64",
        "value": units * args.steps / wall,
        "unit": "edges/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "real edge rows, synthetic features" if args.workload.endswith("_real") else "synthetic",
        "config": {"workload": f"{args.workload + ' (edge list of the parity fixture)' if args.workload.endswith('_real') else args.workload + '-synthetic'}(seed 0): N={N}, edge rows={sg.src.size}, 85/5/10 split, "
                               f"E_sym={E}, scored train pairs P={P} (m=5), K={K}, d={d}, beta={beta}, t={t}; "
                               "forward route+aggregate+score_pairs",
                   "K": K, "d": d, "n_nodes": N, "E_sym": E, "P": P, "fast_path": bool(lib.dl_has_fast_path(K, d))
                   and not args.force_generic},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": kernels[dom]["achieved_GBs"], "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": kernels[dom]["frac"],
                     "traffic": kernels[dom]["traffic"],
                     "traffic_source": "profiles/pmc_traffic_latest.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
                                       "separate passes of this command; bytes leaving the XCD L2s per launch)",
                     "algorithmic_bytes": kernels[dom]["algorithmic_bytes"], "avg_us": kernels[dom]["avg_us"]},
        "edge_scatter": {"kernels": "route+aggregate", "avg_us": scatter_t * 1e6, "algorithmic_bytes": scatter_b,
                         "achieved_GBs": scatter_b / scatter_t / 1e9, "frac": scatter_b / scatter_t / 1e9 / HBM_PEAK_GBS,
                         "traffic": (kernels["route"]["traffic"] + kernels["aggregate"]["traffic"])
                         if default_case and None not in (kernels["route"]["traffic"], kernels["aggregate"]["traffic"]) else None,
                         "edges_per_s": E / scatter_t},
        "kernels": kernels,
        "fwd_bwd": {"ms_per_step": fb_ms, "edges_per_s": units / (fb_ms * 1e-3)},
        "scorer_training_step": scorer_train,
        "projection": proj,
        "dense_allpairs": dense,
    }
    if not args.no_cpu_baseline:
        gcpu = graph.to("cpu")
        Zc = Z.float().cpu()
        if N <= 12000:                                          # dense [K,N,N] fits: the reference's own form
            result["cpu_baseline"] = cpu_baseline(Zc, gcpu, units, beta, t)
        else:
            result["cpu_baseline"] = cpu_baseline_sparse(Zc, gcpu, (pairs.pu.cpu(), pairs.pv.cpu()), units, beta, t)
    print(json.dumps(result))


if __name__ == "__main__":
    main()
