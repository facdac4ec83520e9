"""Row-sharded hot path: one process per GPU, RCCL collectives over xGMI (``torch.distributed``,
backend "nccl" == RCCL on ROCm).  The reference is single-process; this is the
MI355X design of SURVEY.md §8(e).

Partition: nodes are cut into ``world`` equal contiguous blocks (padded with isolated nodes so the
blocks are equal, which keeps every collective a plain all-gather).  Rank r owns the CSR rows, the
incidence rows and the feature rows of its block and a replica of the MLP weights.  Everything is
"owner computes": each rank produces only rows of its own nodes, gathering what it needs from
neighbours' rows, so there is no reduce-scatter and no float atomics anywhere.

  forward   Z_loc = MLP(x_loc)            -> all-gather Z   [N,K,d]
            route on local rows           -> all-gather s   [N,K]     (normaliser of the NEIGHBOUR, model.py:73)
            aggregate on local rows       -> all-gather H   [N,K,d]   (before scoring, BASELINE.json north_star)
            score the local slice of the pair list
  backward  all-gather (prob, g_prob)     [P]   (8 B per pair)
            scorer backward on local incidence rows -> dH_loc, dZ_loc
            all-gather dH [N,K,d]; phase 1 on local rows -> all-gather ds [N,K]; phase 2 -> dZ_loc
            MLP backward locally; all-reduce of the weight gradients

The kernels are reached through a small backend object so that the choreography can be exercised
on CPU with gloo in tests (tests/ supply an oracle-backed stand-in); the product default is the
HIP backend and there is no fallback.
"""
from __future__ import annotations

import time
from dataclasses import dataclass

import numpy as np
import torch
import torch.distributed as dist

from .graph import Graph, PairList


# --------------------------------------------------------------------------- partition
def block_size(n_nodes: int, world: int) -> int:
    return (n_nodes + world - 1) // world


def padded_nodes(n_nodes: int, world: int) -> int:
    return block_size(n_nodes, world) * world


def row_range(n_nodes: int, world: int, rank: int) -> tuple[int, int]:
    """Rows of rank `rank` in the PADDED node space [0, padded_nodes)."""
    b = block_size(n_nodes, world)
    return rank * b, (rank + 1) * b


def pair_slices(pu_sorted: np.ndarray, n_nodes: int, world: int):
    """Pairs (sorted by u) are scored by the owner of u: contiguous slices of the list."""
    b = block_size(n_nodes, world)
    cuts = np.searchsorted(pu_sorted, np.arange(world + 1) * b, side="left")
    cuts[-1] = pu_sorted.size
    return cuts


def all_gather_rows(full: torch.Tensor, lo: int, hi: int, group=None, src: torch.Tensor | None = None) -> None:
    """Every rank contributes rows [lo, hi) of `full` (equal sizes on all ranks) and receives all rows.  The
    send buffer never aliases the receive buffer: `src` if the caller still holds the local rows elsewhere,
    else a copy of full[lo:hi] (a few microseconds against a collective of 8x the bytes)."""
    local = src if src is not None else full[lo:hi].clone()
    if full.is_cuda and dist.get_backend(group) == "gloo":          # one-GPU rehearsal: gloo moves host memory
        host = torch.empty(full.shape, dtype=full.dtype)
        dist.all_gather_into_tensor(host, local.contiguous().cpu(), group=group)
        full.copy_(host)
        return
    dist.all_gather_into_tensor(full, local.contiguous(), group=group)


# --------------------------------------------------------------------------- backends
class HipBackend:
    """The product backend: libdisenlink_hip.so through disenlink_amd.ops."""

    def __init__(self):
        from . import ops
        self.ops = ops

    def route_fwd(self, g, Z, t, s_out):
        return self.ops.route_fwd(g, Z, t, s_out=s_out)[:2]

    def aggregate_fwd(self, g, Z, beta, p, a, s, H_out):
        self.ops.aggregate_fwd(g, Z, beta, p, a, s, H_out=H_out)

    def score_pairs_fwd(self, Z, H, pairs, t):
        return self.ops.score_pairs_fwd(Z, H, pairs.pu, pairs.pv, t, pairs)

    def score_pairs_bwd(self, Z, H, inc, t, prob, g_prob, dZ_out, dH_out):
        self.ops.score_pairs_bwd(Z, H, inc, t, prob, g_prob, dZ_out=dZ_out, dH_out=dH_out)

    def bwd_phase1(self, g, Z, beta, p, a, s, dH, ds_out):
        return self.ops.route_aggregate_bwd_phase1(g, Z, beta, p, a, s, dH, ds_out)

    def bwd_phase2(self, g, Z, beta, t, p, a, s, dH, dw, dwr, ds, dZ_out, accumulate):
        self.ops.route_aggregate_bwd_phase2(g, Z, beta, t, p, a, s, dH, dw, dwr, ds, dZ_out, accumulate)


def _incidence_only(pu, pv, n_nodes, lo, hi, row_bytes=2048) -> PairList:
    """PairList over the WHOLE pair list whose incidence rows are this shard's nodes; its forward
    plan is left empty (the forward scores a slice through another PairList)."""
    full = PairList.build(pu, pv, n_nodes, row_range=(lo, hi), by_u_range=(0, n_nodes), build_by_u=False,
                          row_bytes=row_bytes)
    return full


# --------------------------------------------------------------------------- shard description
@dataclass
class Shard:
    rank: int
    world: int
    n_nodes: int            # real nodes
    n_pad: int              # padded node count (extent of node-indexed arrays)
    lo: int                 # first local row (padded space)
    hi: int
    graph: Graph            # local rows of adj_sym, global columns
    pairs: PairList         # local slice of the pair list (forward)
    inc: PairList           # incidence rows of local nodes over the WHOLE pair list (backward)
    pair_lo: int            # position of the local slice in the global pair list
    pair_hi: int
    n_pairs_total: int
    pair_block: int         # padded per-rank pair count used by the (prob, g_prob) all-gather
    pair_cuts: np.ndarray

    @staticmethod
    def build(rank: int, world: int, n_nodes: int, edge_src, edge_dst, pu, pv, device,
              seg_len: int = 32, row_bytes: int = 2048) -> "Shard":
        """edge rows = TRAIN edge rows (directed, duplicates ok); pu/pv = the global pair list, sorted by pu."""
        pu = np.asarray(pu, dtype=np.int64)
        pv = np.asarray(pv, dtype=np.int64)
        if pu.size and np.any(np.diff(pu) < 0):
            raise ValueError("the pair list must be sorted by pu (pairs are scored by the owner of u)")
        n_pad = padded_nodes(n_nodes, world)
        lo, hi = row_range(n_nodes, world, rank)
        ts = torch.as_tensor(np.asarray(edge_src), device=device)
        td = torch.as_tensor(np.asarray(edge_dst), device=device)
        graph = Graph.from_edge_rows(ts, td, n_pad, symmetrise=True, seg_len=seg_len, row_range=(lo, hi),
                                     row_bytes=row_bytes)
        cuts = pair_slices(pu, n_nodes, world)
        q0, q1 = int(cuts[rank]), int(cuts[rank + 1])
        tpu, tpv = torch.as_tensor(pu, device=device), torch.as_tensor(pv, device=device)
        pairs = PairList.build(tpu[q0:q1], tpv[q0:q1], n_pad, row_range=(lo, lo), by_u_range=(lo, hi),
                               row_bytes=row_bytes)
        inc = _incidence_only(tpu, tpv, n_pad, lo, hi, row_bytes)
        block = int(np.max(np.diff(cuts))) if pu.size else 0
        return Shard(rank, world, n_nodes, n_pad, lo, hi, graph, pairs, inc, q0, q1, int(pu.size), block, cuts)

    def pad_rows(self, x_local_real: torch.Tensor) -> torch.Tensor:
        """Feature rows of this rank's block, zero rows for padding nodes."""
        rows = self.hi - self.lo
        if x_local_real.shape[0] == rows:
            return x_local_real
        out = x_local_real.new_zeros((rows,) + tuple(x_local_real.shape[1:]))
        out[:x_local_real.shape[0]] = x_local_real
        return out

    def local_real_rows(self) -> tuple[int, int]:
        return min(self.lo, self.n_nodes), min(self.hi, self.n_nodes)


# --------------------------------------------------------------------------- autograd over the shard
class ShardedHotPath(torch.autograd.Function):
    """Z_loc [rows,K,d] -> (H_loc [rows,K,d], prob_loc [local pairs]) with the collectives inside."""

    @staticmethod
    def forward(ctx, Z_loc, shard: Shard, backend, beta: float, t: float, group):
        sh = shard
        K, d = Z_loc.shape[1], Z_loc.shape[2]
        dev = Z_loc.device
        Z = torch.empty((sh.n_pad, K, d), dtype=torch.float32, device=dev)
        all_gather_rows(Z, sh.lo, sh.hi, group, src=Z_loc.detach().float())
        s = torch.empty((sh.n_pad, K), dtype=torch.float32, device=dev)
        p, a = backend.route_fwd(sh.graph, Z, t, s)
        all_gather_rows(s, sh.lo, sh.hi, group)
        H = torch.empty_like(Z)
        backend.aggregate_fwd(sh.graph, Z, beta, p, a, s, H)
        all_gather_rows(H, sh.lo, sh.hi, group)
        prob = backend.score_pairs_fwd(Z, H, sh.pairs, t)
        ctx.shard, ctx.backend, ctx.beta, ctx.t, ctx.group = sh, backend, beta, t, group
        ctx.save_for_backward(Z, H, s, a, prob)
        ctx.p = p
        return H[sh.lo:sh.hi].clone(), prob

    @staticmethod
    def backward(ctx, gH_loc, g_prob):
        sh, be, beta, t, group = ctx.shard, ctx.backend, ctx.beta, ctx.t, ctx.group
        Z, H, s, a, prob = ctx.saved_tensors
        dev = Z.device
        # (prob, g_prob) of every pair: padded equal blocks, then compacted to the global pair order
        blk = sh.pair_block
        buf = torch.zeros((sh.world, 2, blk), dtype=torch.float32, device=dev)
        n_loc = sh.pair_hi - sh.pair_lo
        buf[sh.rank, 0, :n_loc] = prob
        if g_prob is not None:
            buf[sh.rank, 1, :n_loc] = g_prob
        flat = buf.view(sh.world, -1)
        all_gather_rows(flat, sh.rank, sh.rank + 1, group)
        sizes = np.diff(sh.pair_cuts)
        prob_all = torch.cat([buf[r, 0, :int(sizes[r])] for r in range(sh.world)])
        g_all = torch.cat([buf[r, 1, :int(sizes[r])] for r in range(sh.world)])
        dZ = torch.zeros_like(Z)
        dH = torch.zeros_like(Z)
        be.score_pairs_bwd(Z, H, sh.inc, t, prob_all, g_all, dZ, dH)
        if gH_loc is not None:
            dH[sh.lo:sh.hi] += gH_loc
        all_gather_rows(dH, sh.lo, sh.hi, group)
        ds = torch.zeros_like(s)
        dw, dwr = be.bwd_phase1(sh.graph, Z, beta, ctx.p, a, s, dH, ds)
        all_gather_rows(ds, sh.lo, sh.hi, group)
        be.bwd_phase2(sh.graph, Z, beta, t, ctx.p, a, s, dH, dw, dwr, ds, dZ, True)
        return dZ[sh.lo:sh.hi].clone(), None, None, None, None, None


def sharded_forward(model, x_local: torch.Tensor, shard: Shard, backend=None, group=None):
    """(emb_local [rows,K*d], prob_local [local pairs]) of the drop-in module on this rank's shard."""
    backend = backend or HipBackend()
    Z_loc = model.project(shard.pad_rows(x_local))
    H_loc, prob = ShardedHotPath.apply(Z_loc, shard, backend, float(model.beta), float(model.temperature), group)
    return H_loc.reshape(H_loc.shape[0], -1), prob


def allreduce_gradients(model, group=None) -> None:
    """Sum the replicas' weight gradients (each rank's loss must already carry the GLOBAL normaliser)."""
    for prm in model.parameters():
        if prm.grad is None:
            prm.grad = torch.zeros_like(prm)
        dist.all_reduce(prm.grad, op=dist.ReduceOp.SUM, group=group)


# --------------------------------------------------------------------------- bench (bench.py --gpus N)
def bench_sharded(args, rank: int, world: int, device) -> dict:
    """Weak scaling: the graph grows with the GPU count (N = world x the 1-GPU node count, same degree
    law), each rank owns one block of rows.  One step = all-gather Z, route, all-gather s, aggregate,
    all-gather H, score the local pairs; value = (E_sym + P over all ranks) / max-over-ranks time."""
    from . import _lib
    from .data import synthetic_graph
    from .model import Disentangle
    from .splits import make_link_split
    lib = _lib.load()
    K, d, beta, t = args.K, args.d, 0.5, 1.0
    # DL_EMULATE_WORLD=8 on one GPU: build rank 0's shard of the 8-GPU problem and time its compute alone
    # (the collectives degenerate to no-ops) — a rehearsal of the per-rank work, not a scaling number.
    import os
    emu = int(os.environ.get("DL_EMULATE_WORLD", "0"))
    if emu > 1 and world == 1:
        return _bench_emulated(args, emu, device)
    sg = synthetic_graph(args.workload, seed=0, scale=args.scale * world)
    split = make_link_split(sg.src, sg.dst, sg.n_nodes, m=5, seed=0)
    pu = np.concatenate([split.pos_train.u, split.neg_train.u])
    pv = np.concatenate([split.pos_train.v, split.neg_train.v])
    order = np.lexsort((pv, pu))
    pu, pv = pu[order], pv[order]
    shard = Shard.build(rank, world, sg.n_nodes, split.train_src, split.train_dst, pu, pv, device, row_bytes=K * d * 4)
    torch.manual_seed(0)
    model = Disentangle(sg.n_feat, args.nhidden, d, nfactor=K, beta=beta, t=1).to(device)
    r0, r1 = shard.local_real_rows()
    x_loc = torch.from_numpy(sg.features()[r0:r1]).to(device)
    backend = HipBackend()
    with torch.no_grad():
        Z_loc = model.project(shard.pad_rows(x_loc)).contiguous()

    Z = torch.empty((shard.n_pad, K, d), dtype=torch.float32, device=device)
    s = torch.empty((shard.n_pad, K), dtype=torch.float32, device=device)
    H = torch.empty_like(Z)

    def step():
        all_gather_rows(Z, shard.lo, shard.hi, src=Z_loc)
        p, a = backend.route_fwd(shard.graph, Z, t, s)
        all_gather_rows(s, shard.lo, shard.hi)
        backend.aggregate_fwd(shard.graph, Z, beta, p, a, s, H)
        all_gather_rows(H, shard.lo, shard.hi)
        return backend.score_pairs_fwd(Z, H, shard.pairs, t)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    red_dev = "cpu" if dist.get_backend() == "gloo" else device
    wall = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=red_dev)
    dist.all_reduce(wall, op=dist.ReduceOp.MAX)
    counts = torch.tensor([shard.graph.n_edges, shard.pairs.n_pairs], dtype=torch.int64, device=red_dev)
    dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    E, P = int(counts[0]), int(counts[1])
    wall_s = float(wall[0])
    # roofline of the dominant kernel on THIS rank's shard: HIP events on the launch stream, kernels only
    # (the collectives are left out of this loop; every rank runs it so nobody waits at the teardown barrier)
    reps = max(5, min(args.steps, 50))
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(reps)]
    for i in range(reps):
        ev[i][0].record()
        p, a = backend.route_fwd(shard.graph, Z, t, s)
        ev[i][1].record()
        backend.aggregate_fwd(shard.graph, Z, beta, p, a, s, H)
        ev[i][2].record()
        backend.score_pairs_fwd(Z, H, shard.pairs, t)
        ev[i][3].record()
    torch.cuda.synchronize()
    kt = [float(np.mean([ev[i][j].elapsed_time(ev[i][j + 1]) for i in range(reps)])) * 1e-3 for j in range(3)]
    e_loc, p_loc, rows = shard.graph.n_edges, shard.pairs.n_pairs, shard.hi - shard.lo
    kb = [e_loc * (K * d * 4 + 9) + rows * (K * d * 4 + K * 4 + 4),            # SURVEY.md §8(d), as bench.algorithmic_bytes
          e_loc * (d * 4 + 13) + rows * (2 * K * d * 4 + K * 4 + 4),
          p_loc * (4 * K * d * 4 + 12)]
    j = int(np.argmax(kt))
    roofline = {"bound": "hbm", "kernel": ("route", "aggregate", "score")[j], "achieved": kb[j] / kt[j] / 1e9,
                "peak": 8000.0, "unit": "GB/s", "frac": kb[j] / kt[j] / 1e9 / 8000.0, "traffic": None,
                "algorithmic_bytes": kb[j], "avg_us": kt[j] * 1e6, "scope": "rank 0's shard, kernels only"}
    dist.barrier()
    return {
        "metric": "edges/sec (aggregate+score) at K=8 d=64",
        "value": (E + P) * args.steps / wall_s, "unit": "edges/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall_s / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "roofline": roofline,
        "config": {"workload": f"{args.workload}-synthetic x{world} (seed 0): N={sg.n_nodes}, edge rows={sg.src.size}, "
                               f"85/5/10 split, E_sym={E}, scored train pairs P={P} (m=5), K={K}, d={d}; row-sharded over "
                               f"{world} GPUs, all-gather of Z, s and H over RCCL each step; forward route+aggregate+score",
                   "K": K, "d": d, "n_nodes": sg.n_nodes, "E_sym": E, "P": P,
                   "parallelism": f"row-shard x{world}", "fast_path": bool(lib.dl_has_fast_path(K, d))},
    }


def _bench_emulated(args, emu_world: int, device) -> dict:
    from .data import synthetic_graph
    from .splits import make_link_split
    K, d, beta, t = args.K, args.d, 0.5, 1.0
    sg = synthetic_graph(args.workload, seed=0, scale=args.scale * emu_world)
    split = make_link_split(sg.src, sg.dst, sg.n_nodes, m=5, seed=0)
    pu = np.concatenate([split.pos_train.u, split.neg_train.u])
    pv = np.concatenate([split.pos_train.v, split.neg_train.v])
    order = np.lexsort((pv, pu))
    shard = Shard.build(0, emu_world, sg.n_nodes, split.train_src, split.train_dst, pu[order], pv[order], device,
                        row_bytes=K * d * 4)
    backend = HipBackend()
    Z = torch.randn((shard.n_pad, K, d), device=device) * 0.24
    s = torch.empty((shard.n_pad, K), dtype=torch.float32, device=device)
    H = torch.empty_like(Z)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    acc = np.zeros(3)
    for it in range(args.warmup + args.steps):
        ev[0].record()
        p, a = backend.route_fwd(shard.graph, Z, t, s)
        ev[1].record()
        backend.aggregate_fwd(shard.graph, Z, beta, p, a, s, H)
        ev[2].record()
        backend.score_pairs_fwd(Z, H, shard.pairs, t)
        ev[3].record()
        torch.cuda.synchronize()
        if it >= args.warmup:
            acc += [ev[i].elapsed_time(ev[i + 1]) for i in range(3)]
    acc /= args.steps
    return {"emulated_world": emu_world, "rank0_edges": shard.graph.n_edges, "rank0_pairs": shard.pairs.n_pairs,
            "n_nodes": sg.n_nodes, "table_MB": shard.n_pad * K * d * 4 / 1e6,
            "route_us": acc[0] * 1e3, "aggregate_us": acc[1] * 1e3, "score_us": acc[2] * 1e3,
            "allgather_bytes_per_rank_per_step": 2 * shard.n_pad * K * d * 4 + shard.n_pad * K * 4}
