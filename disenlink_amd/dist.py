"""Row-sharded hot path: one process per GPU, RCCL collectives over xGMI (``torch.distributed``,
backend "nccl" == RCCL on ROCm).  The reference is single-process; this is the
MI355X design of SURVEY.md §8(e).

Partition.  Nodes are cut into ``world`` CONTIGUOUS blocks balanced by work, not by node count (squirrel: median degree
17, maximum 1,904): the cut points are the quantiles of the per-node weight "edge rows touching the node + 1"
(:func:`balanced_cuts`).  Every block is padded with isolated nodes to the size B of the largest one, so that every
collective stays a plain equal-size all-gather; node ids are relabelled into that padded space (rank r owns
[r B, (r+1) B), real rows first).  Rank r owns the CSR rows, the incidence rows and the feature rows of its block and a
replica of the MLP weights.  Everything is "owner computes": each rank produces only rows of its own nodes, gathering
what it needs from neighbours' rows, so there is no reduce-scatter and no float atomics anywhere.

  forward   Z_loc = MLP(x_loc)            -> all-gather Z   [N,K,d]
            route on local rows           -> all-gather s   [N,K]     (normaliser of the NEIGHBOUR, model.py:73)
            aggregate on local rows       -> all-gather H   [N,K,d]   (before scoring, BASELINE.json north_star),
                                             in C row chunks, asynchronously: pairs whose second endpoint is local are
                                             scored at once, pairs whose second endpoint lies in chunk c as soon as
                                             chunk c has landed — the scorer runs under the rest of the gather
  backward  all-gather (prob, g_prob)     [P]   (8 B per pair)
            scorer backward on local incidence rows -> dH_loc, dZ_loc
            all-gather dH [N,K,d]; phase 1 on local rows -> all-gather ds [N,K]; phase 2 -> dZ_loc
            MLP backward locally; all-reduce of the weight gradients

The kernels are reached through a small backend object so that the choreography can be exercised
on CPU with gloo in tests (tests/ supply an oracle-backed stand-in); the product default is the
HIP backend and there is no fallback.
"""
from __future__ import annotations

import os
import time
from dataclasses import dataclass, field

import numpy as np
import torch
import torch.distributed as dist

from .graph import Graph, PairList

DEFAULT_CHUNKS = int(os.environ.get("DL_GATHER_CHUNKS", "4"))


# --------------------------------------------------------------------------- partition
def block_size(n_nodes: int, world: int) -> int:
    return (n_nodes + world - 1) // world


def padded_nodes(n_nodes: int, world: int) -> int:
    return block_size(n_nodes, world) * world


def row_range(n_nodes: int, world: int, rank: int) -> tuple[int, int]:
    """Rows of rank `rank` in the padded node space of EQUAL node blocks (balance="nodes")."""
    b = block_size(n_nodes, world)
    return rank * b, (rank + 1) * b


def pair_slices(pu_sorted: np.ndarray, n_nodes: int, world: int):
    """Pairs (sorted by u) are scored by the owner of u: contiguous slices of the list (equal node blocks)."""
    b = block_size(n_nodes, world)
    cuts = np.searchsorted(pu_sorted, np.arange(world + 1) * b, side="left")
    cuts[-1] = pu_sorted.size
    return cuts


def balanced_cuts(weight: np.ndarray, world: int) -> np.ndarray:
    """Cut points [world + 1] of contiguous node blocks of (nearly) equal total weight: block r = [cuts[r], cuts[r+1]).
    A node goes to the block in which the MIDPOINT of its weight interval falls, so a hub heavier than a whole share
    takes a block of its own instead of dragging its neighbours along."""
    w = np.asarray(weight, dtype=np.float64)
    n = w.size
    if n == 0:
        return np.zeros(world + 1, dtype=np.int64)
    csum = np.cumsum(w)
    mid = csum - 0.5 * w
    owner = np.minimum((mid * world / csum[-1]).astype(np.int64), world - 1)
    owner = np.maximum.accumulate(owner)                     # monotone by construction; guard against rounding
    cuts = np.searchsorted(owner, np.arange(world + 1), side="left")
    cuts[-1] = n
    return cuts.astype(np.int64)


@dataclass
class Partition:
    """Relabelling of the n real nodes into `world` padded blocks of `block` ids each."""
    world: int
    n_nodes: int
    cuts: np.ndarray          # [world+1] real-node cut points
    block: int                # B: ids per rank in the padded space (a multiple of n_chunks)
    n_chunks: int

    @property
    def n_pad(self) -> int:
        return self.block * self.world

    @property
    def chunk_rows(self) -> int:
        return self.block // self.n_chunks

    def to_padded(self, ids) -> np.ndarray:
        ids = np.asarray(ids, dtype=np.int64)
        owner = np.searchsorted(self.cuts, ids, side="right") - 1
        return owner * self.block + (ids - self.cuts[owner])

    def real_rows(self, rank: int) -> tuple[int, int]:
        return int(self.cuts[rank]), int(self.cuts[rank + 1])

    @staticmethod
    def build(n_nodes: int, world: int, edge_src=None, edge_dst=None, balance: str = "nnz",
              n_chunks: int = 1) -> "Partition":
        if balance == "nodes" or edge_src is None:
            b = block_size(n_nodes, world)
            cuts = np.minimum(np.arange(world + 1, dtype=np.int64) * b, n_nodes)
        elif balance == "nnz":
            w = (np.bincount(np.asarray(edge_src, dtype=np.int64), minlength=n_nodes)
                 + np.bincount(np.asarray(edge_dst, dtype=np.int64), minlength=n_nodes) + 1)
            cuts = balanced_cuts(w, world)
        else:
            raise ValueError("balance must be 'nnz' or 'nodes'")
        rows = int(np.max(np.diff(cuts))) if n_nodes else 0
        block = max(1, -(-max(rows, 1) // n_chunks)) * n_chunks
        return Partition(world, n_nodes, cuts, block, n_chunks)


# --------------------------------------------------------------------------- collectives
def _gloo_on_device(t: torch.Tensor, group) -> bool:
    return t.is_cuda and dist.get_backend(group) == "gloo"          # one-GPU rehearsal: gloo moves host memory


def all_gather_rows(full: torch.Tensor, lo: int, hi: int, group=None, src: torch.Tensor | None = None) -> None:
    """Every rank contributes rows [lo, hi) of `full` (equal sizes on all ranks) and receives all rows.  The
    send buffer never aliases the receive buffer: `src` if the caller still holds the local rows elsewhere,
    else a copy of full[lo:hi] (a few microseconds against a collective of 8x the bytes)."""
    local = src if src is not None else full[lo:hi].clone()
    if _gloo_on_device(full, group):
        host = torch.empty(full.shape, dtype=full.dtype)
        dist.all_gather_into_tensor(host, local.contiguous().cpu(), group=group)
        full.copy_(host)
        return
    dist.all_gather_into_tensor(full, local.contiguous(), group=group)


class ChunkedRowGather:
    """All-gather of a node table in C row chunks, asynchronously: chunk c = rows [c Bc, (c+1) Bc) of EVERY rank's
    block.  `start` enqueues all C collectives (each into the views of `full` the chunk belongs to; the own block is
    already in place and is not rewritten — its slot goes to a scratch buffer); `wait(c)` makes the current stream wait
    for chunk c only, so kernels that need chunk c run under the transfer of chunks c+1 .."""

    def __init__(self, full: torch.Tensor, part: Partition, rank: int, group=None):
        self.full, self.part, self.rank, self.group = full, part, rank, group
        self.works, self._keep = [], []

    def start(self):
        B, Bc, W = self.part.block, self.part.chunk_rows, self.part.world
        full = self.full
        host_path = _gloo_on_device(full, self.group)
        for c in range(self.part.n_chunks):
            send = full[self.rank * B + c * Bc: self.rank * B + (c + 1) * Bc].clone()
            if host_path:
                outs = [torch.empty(send.shape, dtype=send.dtype) for _ in range(W)]
                work = dist.all_gather(outs, send.cpu(), group=self.group, async_op=True)
                self._keep.append((send, outs))
            else:
                scratch = torch.empty_like(send)
                outs = [scratch if q == self.rank else full[q * B + c * Bc: q * B + (c + 1) * Bc] for q in range(W)]
                work = dist.all_gather(outs, send, group=self.group, async_op=True)
                self._keep.append((send, scratch))
            self.works.append(work)
        return self

    def wait(self, c: int):
        self.works[c].wait()
        if _gloo_on_device(self.full, self.group):
            B, Bc = self.part.block, self.part.chunk_rows
            for q, o in enumerate(self._keep[c][1]):
                if q != self.rank:
                    self.full[q * B + c * Bc: q * B + (c + 1) * Bc].copy_(o)

    def wait_all(self):
        for c in range(len(self.works)):
            self.wait(c)
        self._keep.clear()


class PeerRowGather:
    """All-gather of a node table PEER BLOCK by peer block, asynchronously: W broadcasts (block q from its owner), so that
    `wait(q)` makes the current stream wait for peer q's rows only — the routing of the entries whose column lies in block
    q runs under the transfer of the blocks behind it (Shard.route_by_peer).  The own block is already in place."""

    def __init__(self, full: torch.Tensor, part: Partition, rank: int, group=None):
        self.full, self.part, self.rank, self.group = full, part, rank, group
        self.works, self._host = [], []

    def start(self):
        B, W = self.part.block, self.part.world
        host_path = _gloo_on_device(self.full, self.group)
        for q in range(W):
            src = dist.get_global_rank(self.group, q) if self.group is not None else q
            blk = self.full[q * B:(q + 1) * B]
            if host_path:
                h = blk.cpu() if q == self.rank else torch.empty(blk.shape, dtype=blk.dtype)
                self._host.append(h)
                self.works.append(dist.broadcast(h, src=src, group=self.group, async_op=True))
            else:
                self.works.append(dist.broadcast(blk, src=src, group=self.group, async_op=True))
        return self

    def wait(self, q: int):
        self.works[q].wait()
        if self._host and q != self.rank:
            B = self.part.block
            self.full[q * B:(q + 1) * B].copy_(self._host[q])

    def wait_all(self):
        for q in range(len(self.works)):
            self.wait(q)
        self._host.clear()


def route_in_arrival_order(backend, shard: "Shard", Z, t, s, gather):
    """p, a (per local entry) and this rank's rows of s, routing the entries peer block by peer block: first those whose
    neighbour is local, then, as `gather.wait(q)` returns, those whose neighbour lives on peer q.  Entry results do not
    depend on the order, and the row sums are taken over the finished arrays by the last call: the same bits as one
    routing pass after a blocking all-gather."""
    dev = Z.device
    p = torch.zeros(shard.graph.n_edges, dtype=torch.uint8, device=dev)
    a = torch.zeros(shard.graph.n_edges, dtype=torch.float32, device=dev)
    s[shard.lo:shard.hi] = 0                                   # (rows without any entry on any peer keep this)
    for q in [shard.rank] + [q for q in range(shard.world) if q != shard.rank]:
        if q != shard.rank:
            gather.wait(q)
        g = shard.route_by_peer[q]
        if g is not None:                                      # (a shard without any entry keeps the zero row sums)
            backend.route_fwd(g, Z, t, s, p_out=p, a_out=a)
    gather.wait_all()
    return p, a


# --------------------------------------------------------------------------- backends
class HipBackend:
    """The product backend: libdisenlink_hip.so through disenlink_amd.ops."""

    def __init__(self):
        from . import ops
        self.ops = ops

    def route_fwd(self, g, Z, t, s_out, p_out=None, a_out=None):
        return self.ops.route_fwd(g, Z, t, s_out=s_out, p_out=p_out, a_out=a_out)[:2]

    def aggregate_fwd(self, g, Z, beta, p, a, s, H_out):
        self.ops.aggregate_fwd(g, Z, beta, p, a, s, H_out=H_out)

    def score_pairs_fwd(self, Z, H, pairs, t):
        return self.ops.score_pairs_fwd(Z, H, pairs.pu, pairs.pv, t, pairs)

    def score_pairs_bwd(self, Z, H, inc, t, prob, g_prob, dZ_out, dH_out):
        self.ops.score_pairs_bwd(Z, H, inc, t, prob, g_prob, dZ_out=dZ_out, dH_out=dH_out)

    def score_pairs_train_supported(self, inc, K, d, table_dtype) -> bool:
        dt = self.ops._lib.DL_F32 if table_dtype == torch.float32 else self.ops._lib.DL_BF16
        return self.ops.score_pairs_train_supported(inc, K, d, dt)

    def score_pairs_train(self, Z, H, inc, t, label, weight):
        """-> prob [P total] (entries of the pairs touching the plan's rows), dZ, dH [n_pad,K,d] (the plan's rows)"""
        return self.ops.score_pairs_train(Z, H, inc, t, label, weight)

    def pair_bce_sum(self, prob, label, weight):
        """sum_q weight BCE(prob, label) with the reference's clamps (no autograd: the gradient came from the scorer)"""
        lib = self.ops._lib.load()
        loss = torch.empty(1, dtype=torch.float32, device=prob.device)
        g = torch.empty_like(prob)
        ws = self.ops._ws.get(8192, prob.device)
        self.ops._lib.check(lib.dl_pair_bce(prob.data_ptr(), label.data_ptr(), weight.data_ptr(), prob.numel(),
                                            loss.data_ptr(), g.data_ptr(), ws.data_ptr(), ws.numel(),
                                            torch.cuda.current_stream().cuda_stream), "dl_pair_bce")
        return loss[0]

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
    graph: Graph            # local rows of adj_sym, global (padded-space) columns
    pairs: PairList         # local slice of the pair list (forward), in list order
    inc: PairList           # incidence rows of local nodes over the WHOLE pair list (backward)
    pair_lo: int            # position of the local slice in the global pair list
    pair_hi: int
    n_pairs_total: int
    pair_block: int         # padded per-rank pair count used by the (prob, g_prob) all-gather
    pair_cuts: np.ndarray
    part: Partition | None = None
    # forward scoring in gather order: [(positions in the local slice, PairList)], first the pairs whose second endpoint
    # is local, then one list per row chunk of the H all-gather (second endpoint remote, in that chunk)
    pair_groups: list = field(default_factory=list)
    # routing under the Z gather: one Graph per peer block q (the same CSR arrays; its routing plan covers only the
    # entries whose column lies in block q — one contiguous run per row, columns being sorted), None where a peer has
    # none; empty list = one routing pass after a blocking all-gather
    route_by_peer: list = field(default_factory=list)

    @staticmethod
    def build(rank: int, world: int, n_nodes: int, edge_src, edge_dst, pu, pv, device,
              seg_len: int = 32, row_bytes: int = 2048, balance: str = "nnz", n_chunks: int = 1,
              with_backward: bool = True, z_by_peer: bool | None = None) -> "Shard":
        """edge rows = TRAIN edge rows (directed, duplicates ok); pu/pv = the global pair list, sorted by pu.
        balance: "nnz" (blocks of equal work) or "nodes" (equal node counts); n_chunks: row chunks of the asynchronous
        H all-gather (1 = one blocking all-gather before scoring).  z_by_peer: route the local-column entries first and
        every peer's entries when ITS block of Z has arrived (DL_Z_BY_PEER=0/1 forces it; default off: it trades one
        all-gather for W broadcasts and has not been timed on real links — and it only makes sense where a (row, peer) run
        still holds several entries, n_edges >= 4 * world * rows, shorter runs being a wavefront per entry or two)."""
        pu = np.asarray(pu, dtype=np.int64)
        pv = np.asarray(pv, dtype=np.int64)
        if pu.size and np.any(np.diff(pu) < 0):
            raise ValueError("the pair list must be sorted by pu (pairs are scored by the owner of u)")
        edge_src, edge_dst = np.asarray(edge_src, dtype=np.int64), np.asarray(edge_dst, dtype=np.int64)
        part = Partition.build(n_nodes, world, edge_src, edge_dst, balance=balance, n_chunks=max(1, n_chunks))
        B, n_pad = part.block, part.n_pad
        lo, hi = rank * B, (rank + 1) * B
        ts = torch.as_tensor(part.to_padded(edge_src), device=device)
        td = torch.as_tensor(part.to_padded(edge_dst), device=device)
        graph = Graph.from_edge_rows(ts, td, n_pad, symmetrise=True, seg_len=seg_len, row_range=(lo, hi),
                                     row_bytes=row_bytes)
        ppu, ppv = part.to_padded(pu), part.to_padded(pv)             # order-preserving: still sorted by u
        cuts = np.searchsorted(ppu, np.arange(world + 1) * B, side="left")
        cuts[-1] = ppu.size
        q0, q1 = int(cuts[rank]), int(cuts[rank + 1])
        tpu, tpv = torch.as_tensor(ppu, device=device), torch.as_tensor(ppv, device=device)
        pairs = PairList.build(tpu[q0:q1], tpv[q0:q1], n_pad, row_range=(lo, lo), by_u_range=(lo, hi),
                               row_bytes=row_bytes)
        groups = []
        if part.n_chunks > 1:
            lv = ppv[q0:q1]
            owner = lv // B
            chunk = (lv % B) // part.chunk_rows
            sel = [np.flatnonzero(owner == rank)] + [np.flatnonzero((owner != rank) & (chunk == c))
                                                     for c in range(part.n_chunks)]
            for idx in sel:
                ti = torch.as_tensor(idx, device=device)
                sub = PairList.build(tpu[q0:q1][ti], tpv[q0:q1][ti], n_pad, row_range=(lo, lo), by_u_range=(lo, hi),
                                     row_bytes=row_bytes) if idx.size else None
                groups.append((ti, sub))
        inc = _incidence_only(tpu, tpv, n_pad, lo, hi, row_bytes) if with_backward else None
        block = int(np.max(np.diff(cuts))) if pu.size else 0
        forced = os.environ.get("DL_Z_BY_PEER")
        if forced is not None and forced != "":
            z_by_peer = forced != "0"
        if z_by_peer is None:
            z_by_peer = False        # opt-in until it has been timed on real links (W broadcasts against one all-gather)
        by_peer = []
        if z_by_peer and world > 1:
            from .graph import CsrPlan, length_order, route_seg_len
            col = graph.col.to(torch.int64)
            for q in range(world):
                keep = (col >= q * B) & (col < (q + 1) * B)
                n_q = int(keep.sum())
                if n_q == 0:
                    by_peer.append(None)
                    continue
                route = CsrPlan.build(graph.rowptr.to(torch.int64), col, n_pad, row_offset=lo,
                                      seg_len=min(seg_len, route_seg_len(n_q)), n_slices=1, keep=keep, unit_segs=1,
                                      by_length=length_order(n_pad, row_bytes))
                route.rowptr, route.col = graph.plan.rowptr, graph.plan.col       # the SAME arrays: only the segments differ
                by_peer.append(Graph(graph.plan, None, route, False))
        return Shard(rank, world, n_nodes, n_pad, lo, hi, graph, pairs, inc, q0, q1, int(pu.size), block, cuts,
                     part, groups, by_peer)

    def pad_rows(self, x_local_real: torch.Tensor) -> torch.Tensor:
        """Feature rows of this rank's block, zero rows for padding nodes."""
        rows = self.hi - self.lo
        if x_local_real.shape[0] == rows:
            return x_local_real
        out = x_local_real.new_zeros((rows,) + tuple(x_local_real.shape[1:]))
        out[:x_local_real.shape[0]] = x_local_real
        return out

    def local_real_rows(self) -> tuple[int, int]:
        """Real node ids [r0, r1) owned by this rank (their rows are the first r1 - r0 of its padded block)."""
        return self.part.real_rows(self.rank)

    def work(self) -> dict:
        return dict(rows=self.local_real_rows()[1] - self.local_real_rows()[0], nnz=self.graph.n_edges,
                    pairs=self.pairs.n_pairs)


def score_local_pairs(backend, shard: Shard, Z, H, t, gather: "ChunkedRowGather | None"):
    """Probabilities of the local pair slice, in list order.  With a chunked gather in flight the pair groups are scored
    in arrival order (local second endpoints first), each under the transfer of the chunks behind it."""
    if gather is None or not shard.pair_groups:
        if gather is not None:
            gather.wait_all()
        return backend.score_pairs_fwd(Z, H, shard.pairs, t)
    prob = torch.empty(shard.pairs.n_pairs, dtype=torch.float32, device=Z.device)
    for gi, (idx, sub) in enumerate(shard.pair_groups):
        if gi >= 1:
            gather.wait(gi - 1)
        if sub is not None:
            prob.index_copy_(0, idx, backend.score_pairs_fwd(Z, H, sub, t))
    gather.wait_all()
    return prob


def _gather_and_route(sh: "Shard", backend, Z_loc, t, group, table_dtype):
    """The gathered Z table, this rank's rows of s (raw sums) and (p, a) of its entries: one blocking all-gather and one
    routing pass, or — Shard.route_by_peer — the per-peer gather with the routing in arrival order."""
    K, d = Z_loc.shape[1], Z_loc.shape[2]
    dev = Z_loc.device
    Z = torch.empty((sh.n_pad, K, d), dtype=table_dtype, device=dev)
    s = torch.empty((sh.n_pad, K), dtype=torch.float32, device=dev)
    if sh.route_by_peer:
        Z[sh.lo:sh.hi] = Z_loc.detach().to(table_dtype)
        p, a = route_in_arrival_order(backend, sh, Z, t, s, PeerRowGather(Z, sh.part, sh.rank, group).start())
    else:
        all_gather_rows(Z, sh.lo, sh.hi, group, src=Z_loc.detach().to(table_dtype))
        p, a = backend.route_fwd(sh.graph, Z, t, s)
    return Z, s, p, a


# --------------------------------------------------------------------------- autograd over the shard
class ShardedHotPath(torch.autograd.Function):
    """Z_loc [rows,K,d] -> (H_loc [rows,K,d], prob_loc [local pairs]) with the collectives inside."""

    @staticmethod
    def forward(ctx, Z_loc, shard: Shard, backend, beta: float, t: float, group, table_dtype=torch.float32):
        """table_dtype: storage type of the gathered Z / H tables (torch.bfloat16 halves the bytes of both all-gathers —
        the step is bound by them on large graphs; arithmetic and every gradient stay fp32)."""
        sh = shard
        K, d = Z_loc.shape[1], Z_loc.shape[2]
        dev = Z_loc.device
        Z, s, p, a = _gather_and_route(sh, backend, Z_loc, t, group, table_dtype)
        all_gather_rows(s, sh.lo, sh.hi, group)
        H = torch.empty_like(Z)
        backend.aggregate_fwd(sh.graph, Z, beta, p, a, s, H)
        if sh.pair_groups:
            prob = score_local_pairs(backend, sh, Z, H, t, ChunkedRowGather(H, sh.part, sh.rank, group).start())
        else:
            all_gather_rows(H, sh.lo, sh.hi, group)
            prob = backend.score_pairs_fwd(Z, H, sh.pairs, t)
        ctx.shard, ctx.backend, ctx.beta, ctx.t, ctx.group = sh, backend, beta, t, group
        ctx.save_for_backward(Z, H, s, a, prob)
        ctx.p = p
        return H[sh.lo:sh.hi].float(), prob

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
        dZ = torch.zeros(Z.shape, dtype=torch.float32, device=dev)
        dH = torch.zeros(Z.shape, dtype=torch.float32, device=dev)
        be.score_pairs_bwd(Z, H, sh.inc, t, prob_all, g_all, dZ, dH)
        if gH_loc is not None:
            dH[sh.lo:sh.hi] += gH_loc
        _route_aggregate_bwd_sharded(be, sh, Z, beta, t, ctx.p, a, s, dH, dZ, group)
        return dZ[sh.lo:sh.hi].clone(), None, None, None, None, None, None


def _route_aggregate_bwd_sharded(be, sh: Shard, Z, beta, t, p, a, s, dH, dZ, group) -> None:
    """dH holds this rank's rows of d loss / d H; dZ its rows of the scorer's d loss / d Z.  All-gather dH, phase 1 on the
    local rows, all-gather ds, phase 2 accumulating into dZ's local rows."""
    all_gather_rows(dH, sh.lo, sh.hi, group)
    ds = torch.zeros_like(s)
    dw, dwr = be.bwd_phase1(sh.graph, Z, beta, p, a, s, dH, ds)
    all_gather_rows(ds, sh.lo, sh.hi, group)
    be.bwd_phase2(sh.graph, Z, beta, t, p, a, s, dH, dw, dwr, ds, dZ, True)


class ShardedHotPathLoss(torch.autograd.Function):
    """Z_loc -> (H_loc, prob_loc, loss_loc) with the scorer's training step in ONE pass over this rank's incidence rows
    (dl_score_pairs_train): the wave that owns a node's pair slots forms prob itself, applies the weighted-BCE gradient
    and accumulates that node's dZ / dH rows — so the backward needs NO all-gather of (prob, g_prob), and every partner
    row is gathered once per direction instead of three times.  label / weight cover the WHOLE pair list (replicated,
    fixed for a run; weight carries the GLOBAL normaliser, metrics.pair_bce_weights).  loss_loc = this rank's share
    sum_{q: u local} w BCE — the caller all-reduces it for the value; its gradient factor must be the same on every rank."""

    @staticmethod
    def forward(ctx, Z_loc, shard: Shard, backend, beta: float, t: float, group, table_dtype, label, weight):
        ctx.set_materialize_grads(False)
        sh = shard
        K, d = Z_loc.shape[1], Z_loc.shape[2]
        dev = Z_loc.device
        Z, s, p, a = _gather_and_route(sh, backend, Z_loc, t, group, table_dtype)
        all_gather_rows(s, sh.lo, sh.hi, group)
        H = torch.empty_like(Z)
        backend.aggregate_fwd(sh.graph, Z, beta, p, a, s, H)
        all_gather_rows(H, sh.lo, sh.hi, group)
        prob_all, dZs, dHs = backend.score_pairs_train(Z, H, sh.inc, t, label, weight)
        q0, q1 = sh.pair_lo, sh.pair_hi
        prob = prob_all[q0:q1].clone()                            # the pairs this rank owns (first endpoint local)
        loss = backend.pair_bce_sum(prob, label[q0:q1].contiguous(), weight[q0:q1].contiguous()) if q1 > q0 else \
            torch.zeros((), dtype=torch.float32, device=dev)
        ctx.shard, ctx.backend, ctx.beta, ctx.t, ctx.group, ctx.p = sh, backend, beta, t, group, p
        ctx.save_for_backward(Z, s, a, dZs, dHs)
        return H[sh.lo:sh.hi].float(), prob, loss

    @staticmethod
    def backward(ctx, gH_loc, g_prob, g_loss):
        if g_prob is not None:
            raise RuntimeError("ShardedHotPathLoss: a gradient on prob needs the general path (ShardedHotPath)")
        sh, be = ctx.shard, ctx.backend
        Z, s, a, dZs, dHs = ctx.saved_tensors
        if g_loss is None:
            dZ, dH = torch.zeros_like(dZs), torch.zeros_like(dHs)
        else:
            dZ, dH = dZs * g_loss, dHs * g_loss
        if gH_loc is not None:
            dH[sh.lo:sh.hi] += gH_loc
        _route_aggregate_bwd_sharded(be, sh, Z, ctx.beta, ctx.t, ctx.p, a, s, dH, dZ, ctx.group)
        return dZ[sh.lo:sh.hi].clone(), None, None, None, None, None, None, None, None


def sharded_forward(model, x_local: torch.Tensor, shard: Shard, backend=None, group=None):
    """(emb_local [rows,K*d], prob_local [local pairs]) of the drop-in module on this rank's shard; the first
    r1 - r0 rows of emb_local are the real nodes shard.local_real_rows().  The gathered tables take the module's
    ``table_dtype`` (bf16: half the all-gather bytes)."""
    backend = backend or HipBackend()
    Z_loc = model.project(shard.pad_rows(x_local))
    H_loc, prob = ShardedHotPath.apply(Z_loc, shard, backend, float(model.beta), float(model.temperature), group,
                                       getattr(model, "table_dtype", torch.float32))
    return H_loc.reshape(H_loc.shape[0], -1), prob


def sharded_forward_loss(model, x_local: torch.Tensor, shard: Shard, label: torch.Tensor, weight: torch.Tensor,
                         backend=None, group=None):
    """(emb_local, prob_local, loss_local): the training step's forward with the one-pass scorer where the backend has
    it for this shape (else the general path + the same weighted BCE).  label / weight: the WHOLE pair list; the global
    loss is the all-reduced sum of loss_local."""
    backend = backend or HipBackend()
    tab = getattr(model, "table_dtype", torch.float32)
    Z_loc = model.project(shard.pad_rows(x_local))
    K, d = Z_loc.shape[1], Z_loc.shape[2]
    if shard.inc is not None and hasattr(backend, "score_pairs_train") and \
            backend.score_pairs_train_supported(shard.inc, K, d, tab):
        H_loc, prob, loss = ShardedHotPathLoss.apply(Z_loc, shard, backend, float(model.beta), float(model.temperature),
                                                     group, tab, label, weight)
        return H_loc.reshape(H_loc.shape[0], -1), prob, loss
    H_loc, prob = ShardedHotPath.apply(Z_loc, shard, backend, float(model.beta), float(model.temperature), group, tab)
    q0, q1 = shard.pair_lo, shard.pair_hi
    y, w = label[q0:q1], weight[q0:q1]
    bce = torch.nn.functional.binary_cross_entropy(prob, y, weight=w, reduction="sum") if q1 > q0 else prob.sum() * 0.0
    return H_loc.reshape(H_loc.shape[0], -1), prob, bce


def allreduce_gradients(model, group=None) -> None:
    """Sum the replicas' weight gradients (each rank's loss must already carry the GLOBAL normaliser)."""
    for prm in model.parameters():
        if prm.grad is None:
            prm.grad = torch.zeros_like(prm)
        dist.all_reduce(prm.grad, op=dist.ReduceOp.SUM, group=group)


# --------------------------------------------------------------------------- bench (bench.py --gpus N)
@dataclass
class BenchProblem:
    """What every rank needs of the benchmark graph: the train edge rows, the scored pairs sorted by (u, v) and the
    generator of the feature rows (any rank can produce exactly its own rows, data.SyntheticGraph.features)."""
    sg: object                 # SyntheticGraph (its src / dst are empty on the ranks that received the problem)
    edge_rows: int
    train_src: np.ndarray
    train_dst: np.ndarray
    pu: np.ndarray
    pv: np.ndarray
    scale: float
    prep_s: float = 0.0


def _build_problem(args, scale: float, device) -> BenchProblem:
    from .data import synthetic_graph
    from .splits import make_link_split
    sg = synthetic_graph(args.workload, seed=0, scale=scale)
    # the sorts and searches of the split run on the GPU when there is one (same split bit for bit, splits.py):
    # snap-patents full size 104 s -> seconds
    dev = device if (device is not None and torch.device(device).type == "cuda") else None
    split = make_link_split(sg.src, sg.dst, sg.n_nodes, m=5, seed=0, device=dev)
    pu = np.concatenate([split.pos_train.u, split.neg_train.u])
    pv = np.concatenate([split.pos_train.v, split.neg_train.v])
    if dev is not None:
        order = torch.argsort(torch.as_tensor(pu * sg.n_nodes + pv, device=dev), stable=True).cpu().numpy()
    else:
        order = np.lexsort((pv, pu))
    return BenchProblem(sg, int(sg.src.size), split.train_src, split.train_dst, pu[order], pv[order], scale)


def _bench_problem(args, world_for_scale: int, rank: int = 0, world: int = 1, device=None) -> BenchProblem:
    """The benchmark problem, built ONCE: rank 0 generates the graph, the split and the sorted pair list and writes the
    four index arrays to /dev/shm; the other ranks read them (round 2: every rank repeated the whole host preparation —
    104 s for snap-patents, times N ranks on one host).  Feature rows are generated per rank, for its own rows only."""
    from .data import SyntheticGraph
    scale = args.scale * (world_for_scale if args.scaling == "weak" else 1)
    t0 = time.perf_counter()
    shm = os.environ.get("DL_SHARE_DIR", "/dev/shm")
    if world == 1 or not dist.is_initialized() or not os.path.isdir(shm):
        prob = _build_problem(args, scale, device)
        prob.prep_s = time.perf_counter() - t0
        return prob
    token = [f"dl_bench_{os.getpid()}_{time.time_ns()}" if rank == 0 else None]
    dist.broadcast_object_list(token, src=0)
    base = os.path.join(shm, token[0])
    names = ("train_src", "train_dst", "pu", "pv")
    meta = [None]
    if rank == 0:
        prob = _build_problem(args, scale, device)
        for n in names:
            np.save(f"{base}_{n}.npy", getattr(prob, n))
        meta = [dict(name=prob.sg.name, n_nodes=prob.sg.n_nodes, n_feat=prob.sg.n_feat, seed=prob.sg.seed,
                     edge_rows=prob.edge_rows)]
    dist.broadcast_object_list(meta, src=0)                       # also the "files are complete" signal
    if rank != 0:
        m = meta[0]
        empty = np.zeros(0, dtype=np.int64)
        sg = SyntheticGraph(m["name"], m["n_nodes"], empty, empty, m["n_feat"], m["seed"])
        arrs = {n: np.load(f"{base}_{n}.npy") for n in names}
        prob = BenchProblem(sg, m["edge_rows"], arrs["train_src"], arrs["train_dst"], arrs["pu"], arrs["pv"], scale)
    try:
        dist.barrier()                                            # everyone has read: rank 0 removes the files
    finally:
        if rank == 0:
            for n in names:
                try:
                    os.remove(f"{base}_{n}.npy")
                except OSError:
                    pass
    prob.prep_s = time.perf_counter() - t0
    return prob


def bench_sharded(args, rank: int, world: int, device) -> dict:
    """One step = all-gather Z, route, all-gather s, aggregate, all-gather H (chunked, asynchronous), score the local pairs
    under it; value = (E_sym + P over all ranks) / max-over-ranks time.  --scaling weak: the graph grows with the GPU
    count (scale x world, same degree law); strong: the same graph on every GPU count.  --dtype bf16: the gathered Z / H
    tables — and the bytes of their all-gathers — are bf16, arithmetic fp32."""
    from . import _lib
    from .model import Disentangle
    lib = _lib.load()
    K, d, beta, t = args.K, args.d, 0.5, 1.0
    # DL_EMULATE_WORLD=8 on one GPU: build rank 0's shard of the 8-GPU problem and time its compute alone
    # (the collectives degenerate to no-ops) — a rehearsal of the per-rank work, not a scaling number.
    emu = int(os.environ.get("DL_EMULATE_WORLD", "0"))
    if emu > 1 and world == 1:
        return _bench_emulated(args, emu, device)
    tab = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    wb = 2 if args.dtype == "bf16" else 4
    t_prep = time.perf_counter()
    prob = _bench_problem(args, world, rank, world, device)
    sg, pu, pv, scale = prob.sg, prob.pu, prob.pv, prob.scale
    shard = Shard.build(rank, world, sg.n_nodes, prob.train_src, prob.train_dst, pu, pv, device, row_bytes=K * d * wb,
                        n_chunks=DEFAULT_CHUNKS, with_backward=False)
    torch.manual_seed(0)
    model = Disentangle(sg.n_feat, args.nhidden, d, nfactor=K, beta=beta, t=1).to(device)
    r0, r1 = shard.local_real_rows()
    x_loc = torch.from_numpy(sg.features(rows=(r0, r1))).to(device)      # this rank's rows only
    torch.cuda.synchronize()
    dist.barrier()
    prep_s = time.perf_counter() - t_prep                                 # launch -> every rank ready for its first collective
    backend = HipBackend()
    with torch.no_grad():
        Z_loc = model.project(shard.pad_rows(x_loc)).contiguous().to(tab)

    Z = torch.empty((shard.n_pad, K, d), dtype=tab, device=device)
    s = torch.empty((shard.n_pad, K), dtype=torch.float32, device=device)
    H = torch.empty_like(Z)
    ev = lambda: torch.cuda.Event(enable_timing=True)

    def step(timers=None):
        e = [ev() for _ in range(8)] if timers is not None else None
        rec = (lambda i: e[i].record()) if e else (lambda i: None)
        rec(0)
        if shard.route_by_peer:                                 # Z peer block by peer block, routing in arrival order
            Z[shard.lo:shard.hi] = Z_loc
            gz = PeerRowGather(Z, shard.part, shard.rank).start()
            rec(1)
            p, a = route_in_arrival_order(backend, shard, Z, t, s, gz)
        else:
            all_gather_rows(Z, shard.lo, shard.hi, src=Z_loc)
            rec(1)
            p, a = backend.route_fwd(shard.graph, Z, t, s)
        rec(2)
        all_gather_rows(s, shard.lo, shard.hi)
        rec(3)
        backend.aggregate_fwd(shard.graph, Z, beta, p, a, s, H)
        rec(4)
        gather = ChunkedRowGather(H, shard.part, shard.rank).start() if shard.pair_groups else None
        if gather is None:
            all_gather_rows(H, shard.lo, shard.hi)
        rec(5)
        prob = score_local_pairs(backend, shard, Z, H, t, gather) if gather is not None else \
            backend.score_pairs_fwd(Z, H, shard.pairs, t)
        rec(6)
        if timers is not None:
            timers.append(e)
        return prob

    for _ in range(args.warmup):
        step()
    blocks = []
    for _ in range(max(1, args.repeats)):
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        dist.barrier()
        blocks.append(time.perf_counter() - t0)
    red_dev = "cpu" if dist.get_backend() == "gloo" else device
    wall = torch.tensor(blocks, dtype=torch.float64, device=red_dev)
    dist.all_reduce(wall, op=dist.ReduceOp.MAX)                          # per block: the slowest rank
    wall_s = float(wall.median())
    counts = torch.tensor([shard.graph.n_edges, shard.pairs.n_pairs], dtype=torch.int64, device=red_dev)
    per_rank = [torch.zeros_like(counts) for _ in range(world)]
    dist.all_gather(per_rank, counts)
    E, P = int(sum(int(c[0]) for c in per_rank)), int(sum(int(c[1]) for c in per_rank))
    # where a step goes on THIS rank: HIP events on the launch stream around every collective and every kernel phase
    # (the scoring phase runs under the chunked H gather: "score" is the span from the first scoring launch to the last
    # chunk's arrival, "h_gather_exposed" what the gather adds before it)
    timers = []
    for _ in range(min(args.steps, 10)):
        step(timers)
    torch.cuda.synchronize()
    span = lambda i, j: float(np.median([e[i].elapsed_time(e[j]) for e in timers]))
    phases = dict(z_gather_ms=span(0, 1), route_ms=span(1, 2), s_gather_ms=span(2, 3), aggregate_ms=span(3, 4),
                  h_gather_start_ms=span(4, 5), score_and_h_gather_ms=span(5, 6))
    comm_ms = phases["z_gather_ms"] + phases["s_gather_ms"] + phases["h_gather_start_ms"]
    # kernels alone (no collectives), for the roofline entry and the compute / exposed-communication split
    reps = max(5, min(args.steps, 20))
    kev = [[ev() for _ in range(4)] for _ in range(reps)]
    for i in range(reps):
        kev[i][0].record()
        p, a = backend.route_fwd(shard.graph, Z, t, s)
        kev[i][1].record()
        backend.aggregate_fwd(shard.graph, Z, beta, p, a, s, H)
        kev[i][2].record()
        backend.score_pairs_fwd(Z, H, shard.pairs, t)
        kev[i][3].record()
    torch.cuda.synchronize()
    kt = [float(np.median([kev[i][j].elapsed_time(kev[i][j + 1]) for i in range(reps)])) * 1e-3 for j in range(3)]
    compute_ms = sum(kt) * 1e3
    step_ms = wall_s / args.steps * 1e3
    mine = torch.tensor([compute_ms, step_ms - compute_ms, comm_ms, phases["score_and_h_gather_ms"], kt[2] * 1e3],
                        dtype=torch.float64, device=red_dev)
    allr = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    import bench as _bench
    mb = _bench.moved_bytes(shard.graph, shard.pairs, K, d, w=wb)
    j = int(np.argmax(kt))
    name = ("route", "aggregate", "score")[j]
    table_bytes = 2 * shard.n_pad * K * d * wb
    bound, peak, bound_how = _bench.memory_bound(table_bytes, mb, None)     # no PMC passes exist for sharded runs
    roofline = {"bound": bound, "bound_decided_by": bound_how, "kernel": name, "achieved": mb[name] / kt[j] / 1e9,
                "peak": peak, "unit": "GB/s", "frac": mb[name] / kt[j] / 1e9 / peak, "traffic": None,
                "moved_bytes": mb[name], "avg_us": kt[j] * 1e6, "scope": "rank 0's shard, kernels only"}
    if bound == "hbm" and roofline["frac"] > _bench.HBM_ACHIEVABLE_FRAC:
        roofline["frac_unverified"] = "above what HBM can deliver and not backed by counters: part of the bytes are cache hits"
    nnz = np.array([int(c[0]) for c in per_rank], dtype=np.float64)
    gather_bytes = (world - 1) * shard.part.block * (2 * K * d * wb + K * 4)       # received per rank and step
    dist.barrier()
    return {
        "metric": "edges/sec (aggregate+score) at K=8 d=64",
        "value": (E + P) * args.steps / wall_s, "unit": "edges/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "repeats": len(blocks),
        "ms_per_step": step_ms, "ms_per_step_blocks": [float(b) / args.steps * 1e3 for b in wall.tolist()],
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "roofline": roofline,
        "per_rank": [dict(rank=r, nnz=int(per_rank[r][0]), pairs=int(per_rank[r][1]), compute_ms=float(allr[r][0]),
                          exposed_comm_ms=float(allr[r][1]), blocking_gathers_ms=float(allr[r][2]),
                          score_under_h_gather_ms=float(allr[r][3]), score_alone_ms=float(allr[r][4]))
                     for r in range(world)],
        "phases_rank0_ms": phases,
        "z_gather": ("by peer block, routing in arrival order (route_ms spans the gather)" if shard.route_by_peer
                     else "one blocking all-gather before the routing"),
        "partition": {"balance": "nnz", "block_rows": shard.part.block, "padded_nodes": shard.n_pad,
                      "nnz_max_over_mean": float(nnz.max() / max(nnz.mean(), 1.0)), "h_gather_chunks": shard.part.n_chunks,
                      "allgather_bytes_received_per_rank_per_step": int(gather_bytes)},
        "prep_s": {"problem_built_once_and_shared": prob.prep_s, "until_first_collective": prep_s},
        "config": {"workload": f"{args.workload}-synthetic x{scale:g} (seed 0): N={sg.n_nodes}, edge rows={prob.edge_rows}, "
                               f"85/5/10 split, E_sym={E}, scored train pairs P={P} (m=5), K={K}, d={d}, {args.dtype} tables; "
                               f"row-sharded over {world} GPUs by work (nnz), all-gather of Z, s and H over RCCL each step, "
                               f"the H gather in {shard.part.n_chunks} chunks under the scorer; forward route+aggregate+score",
                   "K": K, "d": d, "n_nodes": sg.n_nodes, "E_sym": E, "P": P,
                   "parallelism": f"row-shard x{world}", "fast_path": bool(lib.dl_has_fast_path_dtype(K, d, 1 if wb == 2 else 0))},
    }


def _bench_emulated(args, emu_world: int, device) -> dict:
    """Rank 0's shard of the `emu_world`-GPU problem on ONE GPU, compute only: per-phase kernel times, the partition's
    balance, and how much of the scoring can start before the H all-gather has delivered anything (pairs whose second
    endpoint is local) or has delivered chunk c."""
    K, d, beta, t = args.K, args.d, 0.5, 1.0
    tab = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    wb = 2 if args.dtype == "bf16" else 4
    t_prep = time.perf_counter()
    prob = _bench_problem(args, emu_world, 0, 1, device)
    sg, pu, pv, scale = prob.sg, prob.pu, prob.pv, prob.scale
    train_src, train_dst = prob.train_src, prob.train_dst
    shard = Shard.build(0, emu_world, sg.n_nodes, train_src, train_dst, pu, pv, device,
                        row_bytes=K * d * wb, n_chunks=DEFAULT_CHUNKS, with_backward=False)
    torch.cuda.synchronize()
    prep_s = time.perf_counter() - t_prep
    backend = HipBackend()
    Z = (torch.randn((shard.n_pad, K, d), device=device) * 0.24).to(tab)
    s = torch.empty((shard.n_pad, K), dtype=torch.float32, device=device)
    H = torch.empty_like(Z)
    ng = len(shard.pair_groups)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4 + ng)]
    acc = np.zeros(3 + ng)
    for it in range(args.warmup + args.steps):
        ev[0].record()
        p, a = backend.route_fwd(shard.graph, Z, t, s)
        ev[1].record()
        backend.aggregate_fwd(shard.graph, Z, beta, p, a, s, H)
        ev[2].record()
        backend.score_pairs_fwd(Z, H, shard.pairs, t)
        ev[3].record()
        for gi, (_idx, sub) in enumerate(shard.pair_groups):
            if sub is not None:
                backend.score_pairs_fwd(Z, H, sub, t)
            ev[4 + gi].record()
        torch.cuda.synchronize()
        if it >= args.warmup:
            acc += [ev[i].elapsed_time(ev[i + 1]) for i in range(3 + ng)]
    acc /= args.steps
    # the routing cut by peer block (Shard.route_by_peer): its kernels alone, every block already present
    by_peer = None
    if shard.route_by_peer:
        class _Arrived:
            def wait(self, q): pass
            def wait_all(self): pass
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(args.warmup + 1):
            e0.record()
            for _ in range(args.steps):
                route_in_arrival_order(backend, shard, Z, t, s, _Arrived())
            e1.record()
            torch.cuda.synchronize()
        peers = [g for g in shard.route_by_peer if g is not None]
        by_peer = {"route_us": e0.elapsed_time(e1) * 1e3 / args.steps, "passes": len(peers)}
    w = np.bincount(train_src, minlength=sg.n_nodes) + np.bincount(train_dst, minlength=sg.n_nodes) + 1
    cuts = shard.part.cuts
    share = np.array([w[cuts[r]:cuts[r + 1]].sum() for r in range(emu_world)], dtype=np.float64)
    groups = [dict(group="second endpoint local" if gi == 0 else f"chunk {gi - 1} of the H gather",
                   pairs=int(idx.numel()), score_us=float(acc[3 + gi] * 1e3)) for gi, (idx, _s) in enumerate(shard.pair_groups)]
    return {"emulated_world": emu_world, "scaling": args.scaling, "dtype": args.dtype, "workload": args.workload,
            "prep_s": {"problem": prob.prep_s, "problem_and_rank0_shard": prep_s,
                       "note": "graph + split + sorted pair list (sorts / searches on the GPU), then rank 0's shard plans"},
            "rank0_edges": shard.graph.n_edges, "rank0_pairs": shard.pairs.n_pairs,
            "n_nodes": sg.n_nodes, "block_rows": shard.part.block, "table_MB": shard.n_pad * K * d * wb / 1e6,
            "work_share_max_over_mean": float(share.max() / share.mean()),
            "route_us": acc[0] * 1e3, "aggregate_us": acc[1] * 1e3, "score_us": acc[2] * 1e3,
            "route_by_peer": by_peer,
            "score_in_gather_order": groups,
            "scored_before_any_chunk_lands": (groups[0]["pairs"] / max(1, shard.pairs.n_pairs)) if groups else 0.0,
            "allgather_bytes_per_rank_per_step": (emu_world - 1) * shard.part.block * (2 * K * d * wb + K * 4)}
