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
# Three ways to move a node table's blocks between the ranks, all of them writing every received row straight into its
# final place in the full table (no list all-gather, hence none of ProcessGroupNCCL's flatten-to-a-temporary-and-copy-out):
#   "allgather" — ONE all_gather_into_tensor (rank q's block IS rows [qB, (q+1)B): the output is the table itself, the
#                 input the rank's own rows in place — RCCL's in-place form, sendbuff = recvbuff + rank * count);
#   "p2p"       — the DIRECT fully connected exchange of SURVEY.md §8(e): one grouped batch of W-1 sends and W-1
#                 receives (one RCCL group call = one kernel driving all seven xGMI links at once), each message one
#                 contiguous block of rows; the only form that also serves ROW CHUNKS of every block (a chunk of rank q's
#                 block is contiguous where it lies), which is what lets the scorer run under the H gather;
#   "broadcast" — W broadcasts, one per owner; `wait(q)` per peer (routing in arrival order).
# Which of them is fastest on real links is decided by measurement in the run itself (bench_sharded: all three are timed
# on the Z table before the timed region and the fastest is used; DL_GATHER_MODE forces one).
GATHER_MODES = ("allgather", "p2p", "broadcast")
MESSAGES = {"collectives": 0, "p2p_ops": 0, "staging_copies": 0}     # counted per process; bench lines quote them per step


def _count(kind: str, n: int = 1) -> None:
    MESSAGES[kind] += n


def reset_message_counts() -> dict:
    out = dict(MESSAGES)
    for k in MESSAGES:
        MESSAGES[k] = 0
    return out


def _gloo_on_device(t: torch.Tensor, group) -> bool:
    return t.is_cuda and dist.get_backend(group) == "gloo"          # one-GPU rehearsal: gloo moves host memory


def _global_rank(group, q: int) -> int:
    return dist.get_global_rank(group, q) if group is not None else q


def all_gather_rows(full: torch.Tensor, lo: int, hi: int, group=None, src: torch.Tensor | None = None) -> None:
    """Every rank contributes rows [lo, hi) of `full` (equal sizes on all ranks, rank-major) and receives all rows: one
    all_gather_into_tensor whose output is `full` itself.  `src`: the local rows, if the caller holds them elsewhere
    (then `full[lo:hi]` need not be filled); otherwise they are taken where they lie — in place over RCCL, through one
    clone where the backend (gloo) does not promise the in-place form."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        # one rank: nothing to exchange — the launch sequence of the sharded step is then the unsharded one (a one-rank
        # RCCL all-gather still costs ~15 us of launch: 3 of them were 18 % of the squirrel step, profiles/r4w)
        if src is not None and (src.data_ptr() != full[lo:hi].data_ptr() or src.dtype != full.dtype):
            full[lo:hi].copy_(src)
        return
    _count("collectives")
    if _gloo_on_device(full, group):
        local = src if src is not None else full[lo:hi]
        host = torch.empty(full.shape, dtype=full.dtype)
        dist.all_gather_into_tensor(host, local.contiguous().cpu(), group=group)
        full.copy_(host)
        return
    if src is None:
        if dist.get_backend(group) == "gloo":
            src = full[lo:hi].clone()
            _count("staging_copies")
        else:
            src = full[lo:hi]                                     # in place: sendbuff == recvbuff + rank * count
    dist.all_gather_into_tensor(full, src.contiguous(), group=group)


class ChunkedRowGather:
    """All-gather of a node table in C row chunks, asynchronously and DIRECT: chunk c = rows [c Bc, (c+1) Bc) of EVERY
    rank's block.  `start` enqueues C grouped batches of point-to-point messages — per chunk W-1 sends of this rank's
    chunk and W-1 receives, each landing contiguously at full[q B + c Bc ...], its final place (the own block is already
    there) — and `wait(c)` makes the current stream wait for chunk c only, so kernels that need chunk c run under the
    transfer of chunks c+1 ..  Batches follow each other on the backend's stream: chunk c is complete before c+1
    starts, all W-1 links busy for each."""

    def __init__(self, full: torch.Tensor, part: Partition, rank: int, group=None):
        self.full, self.part, self.rank, self.group = full, part, rank, group
        self.works, self._host, self._done = [], [], set()

    def start(self):
        for c in range(self.part.n_chunks):
            self.start_chunk(c)
        return self

    def start_chunk(self, c: int):
        """Enqueue the exchange of chunk c alone (chunks must be started in order 0, 1, ...): for a producer that fills
        the own block chunk by chunk — the exchange of chunk c then runs under the production of chunk c + 1."""
        assert c == len(self.works), "chunks are started in order"
        B, Bc, W = self.part.block, self.part.chunk_rows, self.part.world
        full, me = self.full, self.rank
        if W == 1:
            self.works.append([])
            self._host.append(None)
            return self
        host_path = _gloo_on_device(full, self.group)
        peers = [(me + k) % W for k in range(1, W)]                # every rank starts with its right-hand neighbour
        rows = lambda q: slice(q * B + c * Bc, q * B + (c + 1) * Bc)
        send = full[rows(me)].cpu() if host_path else full[rows(me)]
        recv = {q: (torch.empty(send.shape, dtype=send.dtype) if host_path else full[rows(q)]) for q in peers}
        ops = []
        for q in peers:
            ops.append(dist.P2POp(dist.isend, send, _global_rank(self.group, q), self.group, tag=c))
            ops.append(dist.P2POp(dist.irecv, recv[q], _global_rank(self.group, q), self.group, tag=c))
        _count("p2p_ops", len(ops))
        self.works.append(dist.batch_isend_irecv(ops))
        self._host.append((send, recv) if host_path else None)
        return self

    def wait(self, c: int):
        if c in self._done:
            return
        self._done.add(c)
        for w in self.works[c]:
            w.wait()
        if self._host and self._host[c] is not None:
            B, Bc = self.part.block, self.part.chunk_rows
            for q, o in self._host[c][1].items():
                self.full[q * B + c * Bc: q * B + (c + 1) * Bc].copy_(o)

    def wait_all(self):
        for c in range(len(self.works)):
            self.wait(c)
        self._host.clear()


class PeerRowGather:
    """All-gather of a node table PEER BLOCK by peer block, asynchronously: W broadcasts (block q from its owner), so that
    `wait(q)` makes the current stream wait for peer q's rows only — the routing of the entries whose column lies in block
    q runs under the transfer of the blocks behind it (Shard.route_by_peer).  The own block is already in place."""

    def __init__(self, full: torch.Tensor, part: Partition, rank: int, group=None):
        self.full, self.part, self.rank, self.group = full, part, rank, group
        self.works, self._host, self._done = [], [], set()

    def start(self):
        B, W = self.part.block, self.part.world
        host_path = _gloo_on_device(self.full, self.group)
        for q in range(W):
            src = _global_rank(self.group, q)
            blk = self.full[q * B:(q + 1) * B]
            _count("collectives")
            if host_path:
                h = blk.cpu() if q == self.rank else torch.empty(blk.shape, dtype=blk.dtype)
                self._host.append(h)
                self.works.append(dist.broadcast(h, src=src, group=self.group, async_op=True))
            else:
                self.works.append(dist.broadcast(blk, src=src, group=self.group, async_op=True))
        return self

    def wait(self, q: int):
        if q in self._done:                                       # (a block already copied is not copied again)
            return
        self._done.add(q)
        self.works[q].wait()
        if self._host and q != self.rank:
            B = self.part.block
            self.full[q * B:(q + 1) * B].copy_(self._host[q])

    def wait_all(self):
        for q in range(len(self.works)):
            self.wait(q)
        self._host.clear()


def gather_table(full: torch.Tensor, part: Partition, rank: int, mode: str = "allgather", group=None,
                 src: torch.Tensor | None = None) -> None:
    """Blocking all-gather of a whole node table in one of GATHER_MODES (the own block must be in `full` for "p2p" and
    "broadcast"; `src` may stand in for it with "allgather")."""
    B = part.block
    if mode == "allgather" or part.world == 1:
        all_gather_rows(full, rank * B, (rank + 1) * B, group, src=src)
        return
    if src is not None:
        full[rank * B:(rank + 1) * B] = src
    if mode == "p2p":
        one = Partition(part.world, part.n_nodes, part.cuts, part.block, 1)       # the whole block as one chunk
        ChunkedRowGather(full, one, rank, group).start().wait_all()
    elif mode == "broadcast":
        PeerRowGather(full, part, rank, group).start().wait_all()
    else:
        raise ValueError(f"gather mode must be one of {GATHER_MODES}")


def default_gather_mode() -> str:
    mode = os.environ.get("DL_GATHER_MODE", "allgather")
    if mode not in GATHER_MODES:
        raise ValueError(f"DL_GATHER_MODE must be one of {GATHER_MODES}")
    return mode


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

    def honours_partial_route_plans(self, K, d, table_dtype) -> bool:
        """Whether dl_route_fwd walks only the entries of the graph's ROUTING plan (the tuned kernels do; the generic
        path — dl_set_force_generic or a (K, d) without a tuned instantiation — routes every entry whatever the plan
        says, so per-peer passes under an asynchronous gather would read blocks that have not arrived)."""
        dt = self.ops._lib.DL_F32 if table_dtype == torch.float32 else self.ops._lib.DL_BF16
        return self.ops.score_terms_available(K, d, dt)

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

    def score_pairs_fwd_terms(self, Z, H, pairs, t):
        """-> prob [P], coef ([2,P,K] per-factor terms for the coefficient-gather backward, or None where the tuned scorer
        does not hand them out)"""
        return self.ops.score_pairs_fwd(Z, H, pairs.pu, pairs.pv, t, pairs, want_coef=True)

    def score_pairs_bwd_terms(self, Z, H, pairs, t, prob, g_prob, coef, dZ_out, dH_out):
        self.ops.score_pairs_bwd(Z, H, pairs, t, prob, g_prob, dZ_out=dZ_out, dH_out=dH_out, coef=coef)

    def pair_bce_grad(self, prob, label, weight):
        """-> (sum_q weight BCE(prob, label), d that / d prob) from ONE kernel (dl_pair_bce)"""
        lib = self.ops._lib.load()
        loss = torch.empty(1, dtype=torch.float32, device=prob.device)
        g = torch.empty_like(prob)
        ws = self.ops._ws_bce(prob.device)
        self.ops._lib.check(lib.dl_pair_bce(prob.data_ptr(), label.data_ptr(), weight.data_ptr(), prob.numel(),
                                            loss.data_ptr(), g.data_ptr(), ws.data_ptr(), ws.numel(),
                                            torch.cuda.current_stream().cuda_stream), "dl_pair_bce")
        return loss[0], g

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
    row_bytes: int = 2048
    _touch: object = field(default=None, repr=False)      # touching(): built on first use
    _global_pairs: tuple | None = field(default=None, repr=False)     # (pu, pv) of the whole list, padded ids (device)

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
        # one rank owns every row: the unsharded graph (its routing plan walks each undirected edge once and mirrors the
        # result through the reverse-edge map — a shard has no reverse entries and routes every directed entry)
        graph = Graph.from_edge_rows(ts, td, n_pad, symmetrise=True, seg_len=seg_len,
                                     row_range=None if world == 1 else (lo, hi), row_bytes=row_bytes)
        ppu, ppv = part.to_padded(pu), part.to_padded(pv)             # order-preserving: still sorted by u
        cuts = np.searchsorted(ppu, np.arange(world + 1) * B, side="left")
        cuts[-1] = ppu.size
        q0, q1 = int(cuts[rank]), int(cuts[rank + 1])
        tpu, tpv = torch.as_tensor(ppu, device=device), torch.as_tensor(ppv, device=device)
        pairs = PairList.build(tpu[q0:q1], tpv[q0:q1], n_pad, row_range=(lo, lo), by_u_range=(lo, hi),
                               row_bytes=row_bytes)
        groups = []
        if part.n_chunks > 1 and world > 1:                        # (one rank: one scoring launch, nothing arrives in chunks)
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
                     part, groups, by_peer, row_bytes, None, (tpu, tpv) if with_backward else None)

    def with_pairs(self, pu, pv, with_backward: bool = False) -> "Shard":
        """The same partition and local graph with ANOTHER global pair list (sorted by pu; e.g. the test pairs, scored
        once with the best weights): only the pair plans are built."""
        pu = np.asarray(pu, dtype=np.int64)
        pv = np.asarray(pv, dtype=np.int64)
        if pu.size and np.any(np.diff(pu) < 0):
            raise ValueError("the pair list must be sorted by pu (pairs are scored by the owner of u)")
        part, B, dev = self.part, self.part.block, self.graph.device
        ppu, ppv = part.to_padded(pu), part.to_padded(pv)
        cuts = np.searchsorted(ppu, np.arange(self.world + 1) * B, side="left")
        cuts[-1] = ppu.size
        q0, q1 = int(cuts[self.rank]), int(cuts[self.rank + 1])
        tpu, tpv = torch.as_tensor(ppu, device=dev), torch.as_tensor(ppv, device=dev)
        pairs = PairList.build(tpu[q0:q1], tpv[q0:q1], self.n_pad, row_range=(self.lo, self.lo),
                               by_u_range=(self.lo, self.hi), row_bytes=self.row_bytes)
        inc = _incidence_only(tpu, tpv, self.n_pad, self.lo, self.hi, self.row_bytes) if with_backward else None
        block = int(np.max(np.diff(cuts))) if pu.size else 0
        return Shard(self.rank, self.world, self.n_nodes, self.n_pad, self.lo, self.hi, self.graph, pairs, inc, q0, q1,
                     int(pu.size), block, cuts, part, [], self.route_by_peer, self.row_bytes, None,
                     (tpu, tpv) if with_backward else None)

    def touching(self):
        """(idx, pairs, own_lo, own_hi): the pairs that TOUCH this rank's nodes — either endpoint local — as a pair list of
        their own (local pair ids; forward plan over all their first endpoints, incidence rows = the local nodes), `idx` =
        their positions in the global list (ascending: the list order is kept), and the contiguous range of it whose FIRST
        endpoint is local (the pairs this rank owns).  With it a rank can run the scorer's training step from kernels that
        need per-pair terms WITHOUT any exchange of per-pair data: it scores every touching pair itself (2 P / W of them
        on average, the same count the one-pass scorer walks) and feeds the terms to the coefficient-gather backward."""
        if self._touch is None:
            if self._global_pairs is None:
                raise ValueError("Shard.build(with_backward=True) is needed for the touching-pair list")
            tpu, tpv = self._global_pairs
            keep = ((tpu >= self.lo) & (tpu < self.hi)) | ((tpv >= self.lo) & (tpv < self.hi))
            idx = torch.nonzero(keep).reshape(-1)
            pu_t, pv_t = tpu[idx], tpv[idx]
            own_lo = int((pu_t < self.lo).sum())
            own_hi = int((pu_t < self.hi).sum())
            pl = PairList.build(pu_t, pv_t, self.n_pad, row_range=(self.lo, self.hi), by_u_range=(0, self.n_pad),
                                row_bytes=self.row_bytes)
            self._touch = (idx, pl, own_lo, own_hi)
        return self._touch

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


def _gather_and_route(sh: "Shard", backend, Z_loc, t, group, table_dtype, pre=None):
    """The gathered Z table, this rank's rows of s (raw sums) and (p, a) of its entries: one blocking all-gather and one
    routing pass, or — Shard.route_by_peer — the per-peer gather with the routing in arrival order.  pre = (Z, gather)
    from project_and_gather: the table with the own block in place and its chunked exchange already in flight."""
    K, d = Z_loc.shape[1], Z_loc.shape[2]
    dev = Z_loc.device
    s = torch.empty((sh.n_pad, K), dtype=torch.float32, device=dev)
    if pre is not None:
        Z, gather = pre
        gather.wait_all()
        p, a = backend.route_fwd(sh.graph, Z, t, s)
        return Z, s, p, a
    if sh.world == 1 and Z_loc.shape[0] == sh.n_pad:           # one rank: its rows ARE the table (no copy, no collective)
        Z = Z_loc.detach().to(table_dtype).contiguous()
        p, a = backend.route_fwd(sh.graph, Z, t, s)
        return Z, s, p, a
    Z = torch.empty((sh.n_pad, K, d), dtype=table_dtype, device=dev)
    by_peer = bool(sh.route_by_peer) and backend.honours_partial_route_plans(K, d, table_dtype)
    if by_peer:
        Z[sh.lo:sh.hi] = Z_loc.detach().to(table_dtype)
        p, a = route_in_arrival_order(backend, sh, Z, t, s, PeerRowGather(Z, sh.part, sh.rank, group).start())
    else:                                                      # one blocking gather, one routing pass over every entry
        gather_table(Z, sh.part, sh.rank, default_gather_mode(), group, src=Z_loc.detach().to(table_dtype))
        p, a = backend.route_fwd(sh.graph, Z, t, s)
    return Z, s, p, a


# --------------------------------------------------------------------------- autograd over the shard
class ShardedHotPath(torch.autograd.Function):
    """Z_loc [rows,K,d] -> (H_loc [rows,K,d], prob_loc [local pairs]) with the collectives inside."""

    @staticmethod
    def forward(ctx, Z_loc, shard: Shard, backend, beta: float, t: float, group, table_dtype=torch.float32, pre=None):
        """table_dtype: storage type of the gathered Z / H tables (torch.bfloat16 halves the bytes of both all-gathers —
        the step is bound by them on large graphs; arithmetic and every gradient stay fp32)."""
        sh = shard
        K, d = Z_loc.shape[1], Z_loc.shape[2]
        dev = Z_loc.device
        Z, s, p, a = _gather_and_route(sh, backend, Z_loc, t, group, table_dtype, pre)
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
        return dZ[sh.lo:sh.hi].clone(), None, None, None, None, None, None, None


def gather_grad_rows(dH: torch.Tensor, sh: "Shard", group, wire_dtype) -> None:
    """All-gather the ranks' rows of the fp32 gradient table dH.  wire_dtype = torch.bfloat16 sends the rows as bf16 —
    HALF the bytes of the largest message of the backward (snap-patents: 6 GB -> 3 GB received per rank and step) — and
    widens the REMOTE rows on arrival; a rank's OWN rows stay the exact fp32 values it computed.  The rounding of the
    remote rows (relative 2^-9 per element of dH) reaches the weight gradients at a few 1e-3 relative (tested over
    gloo); it rides on the same switch as the bf16 forward tables (Disentangle(table_dtype=torch.bfloat16)), whose own
    rounding is of the same size.  DL_DH_GATHER=f32 keeps the fp32 wire."""
    if wire_dtype == torch.float32 or sh.world == 1 or os.environ.get("DL_DH_GATHER", "") == "f32":
        all_gather_rows(dH, sh.lo, sh.hi, group)
        return
    wire = torch.empty(dH.shape, dtype=wire_dtype, device=dH.device)
    all_gather_rows(wire, sh.lo, sh.hi, group, src=dH[sh.lo:sh.hi].to(wire_dtype))
    if sh.lo > 0:
        dH[:sh.lo] = wire[:sh.lo]
    if sh.hi < dH.shape[0]:
        dH[sh.hi:] = wire[sh.hi:]


def _route_aggregate_bwd_sharded(be, sh: Shard, Z, beta, t, p, a, s, dH, dZ, group) -> None:
    """dH holds this rank's rows of d loss / d H; dZ its rows of the scorer's d loss / d Z.  All-gather dH (on a bf16
    wire when the forward tables are bf16: gather_grad_rows), phase 1 on the local rows, all-gather ds, phase 2
    accumulating into dZ's local rows."""
    gather_grad_rows(dH, sh, group, Z.dtype)
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
    def forward(ctx, Z_loc, shard: Shard, backend, beta: float, t: float, group, table_dtype, label, weight,
                scorer: str = "one_pass", pre=None):
        """scorer = "one_pass": dl_score_pairs_train over the rank's incidence rows; "terms": the rank scores every pair
        that touches its nodes with the forward scorer (keeping the per-factor terms), forms the BCE gradient of those
        pairs itself and runs the coefficient-gather backward (Shard.touching) — for shapes whose one-pass kernel is slow
        (K = 16, d = 128: one wave per SIMD) or missing; neither form exchanges per-pair data."""
        ctx.set_materialize_grads(False)
        sh = shard
        K, d = Z_loc.shape[1], Z_loc.shape[2]
        dev = Z_loc.device
        Z, s, p, a = _gather_and_route(sh, backend, Z_loc, t, group, table_dtype, pre)
        all_gather_rows(s, sh.lo, sh.hi, group)
        H = torch.empty_like(Z)
        backend.aggregate_fwd(sh.graph, Z, beta, p, a, s, H)
        all_gather_rows(H, sh.lo, sh.hi, group)
        q0, q1 = sh.pair_lo, sh.pair_hi
        if scorer == "terms":
            idx, touch, a0, a1 = sh.touching()
            y_t, w_t = label.index_select(0, idx).contiguous(), weight.index_select(0, idx).contiguous()
            prob_t, coef = backend.score_pairs_fwd_terms(Z, H, touch, t)
            _tot, g_t = backend.pair_bce_grad(prob_t, y_t, w_t)           # d (sum over the touching pairs) / d prob
            dZs = torch.zeros(Z.shape, dtype=torch.float32, device=dev)
            dHs = torch.zeros(Z.shape, dtype=torch.float32, device=dev)
            backend.score_pairs_bwd_terms(Z, H, touch, t, prob_t, g_t, coef, dZs, dHs)
            prob = prob_t[a0:a1].clone()                                  # the pairs this rank owns (first endpoint local)
            assert a1 - a0 == q1 - q0
        else:
            prob_all, dZs, dHs = backend.score_pairs_train(Z, H, sh.inc, t, label, weight)
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
        return dZ[sh.lo:sh.hi].clone(), None, None, None, None, None, None, None, None, None, None


def project_and_gather(model, x_pad: torch.Tensor, shard: Shard, group, table_dtype):
    """(Z_loc, pre): the projection of this rank's rows (model.py:106; autograd input of the sharded hot path) and — when
    the partition has row chunks (Partition.n_chunks > 1), there is more than one rank and DL_Z_OVERLAP=1 — the Z
    table with this rank's block in place and its exchange IN FLIGHT: chunk c of the local rows is projected, stored to its
    place in the table and handed to the direct exchange (ChunkedRowGather.start_chunk) while chunk c + 1 is projected, so
    only the last chunk's transfer is exposed.  The K MLPs act row by row, so the chunks'
    outputs are the rows of the whole projection; the weight gradients become a sum over the chunks' backward passes (a
    different summation order from the single launch: equal to rounding, not bit for bit).  pre = None otherwise (the hot
    path then gathers Z itself, in one collective, and routes the peers' blocks in arrival order where Shard.route_by_peer
    is set).  OFF unless asked for: this form gives up the arrival-order routing and the bit-identical weight gradients of
    the single launch, and it has never been timed on real links — `bench.py --gpus N` times both forms in the run and
    switches it on (DL_Z_OVERLAP=1 for its own timed steps) only where it wins by >= 5 % on every rank."""
    part = shard.part
    if shard.world == 1 or part.n_chunks <= 1 or os.environ.get("DL_Z_OVERLAP", "0") != "1" or \
            (dist.is_initialized() and dist.get_world_size(group) == 1):
        return model.project(x_pad), None
    Bc = part.chunk_rows
    pieces, Z, gather = [], None, None
    for c in range(part.n_chunks):
        Zc = model.project(x_pad[c * Bc:(c + 1) * Bc])
        if Z is None:
            Z = torch.empty((shard.n_pad,) + tuple(Zc.shape[1:]), dtype=table_dtype, device=Zc.device)
            gather = ChunkedRowGather(Z, part, shard.rank, group)
        with torch.no_grad():
            Z[shard.lo + c * Bc: shard.lo + (c + 1) * Bc] = Zc.detach().to(table_dtype)
        gather.start_chunk(c)
        pieces.append(Zc)
    return torch.cat(pieces, dim=0), (Z, gather)


def sharded_forward(model, x_local: torch.Tensor, shard: Shard, backend=None, group=None):
    """(emb_local [rows,K*d], prob_local [local pairs]) of the drop-in module on this rank's shard; the first
    r1 - r0 rows of emb_local are the real nodes shard.local_real_rows().  The gathered tables take the module's
    ``table_dtype`` (bf16: half the all-gather bytes)."""
    backend = backend or HipBackend()
    tab = getattr(model, "table_dtype", torch.float32)
    Z_loc, pre = project_and_gather(model, shard.pad_rows(x_local), shard, group, tab)
    H_loc, prob = ShardedHotPath.apply(Z_loc, shard, backend, float(model.beta), float(model.temperature), group, tab, pre)
    return H_loc.reshape(H_loc.shape[0], -1), prob


def sharded_forward_loss(model, x_local: torch.Tensor, shard: Shard, label: torch.Tensor, weight: torch.Tensor,
                         backend=None, group=None):
    """(emb_local, prob_local, loss_local): the training step's forward with the one-pass scorer where the backend has
    it for this shape (else the general path + the same weighted BCE).  label / weight: the WHOLE pair list; the global
    loss is the all-reduced sum of loss_local."""
    backend = backend or HipBackend()
    tab = getattr(model, "table_dtype", torch.float32)
    Z_loc, pre = project_and_gather(model, shard.pad_rows(x_local), shard, group, tab)
    K, d = Z_loc.shape[1], Z_loc.shape[2]
    # Which scorer: the module's own rule (ops.one_pass_scorer_wanted); shapes without a one-pass kernel — and wide bf16 rows
    # that have only the group-per-entry one — take the forward scorer with stored terms + the coefficient-gather backward
    # over the pairs that touch the rank's nodes.  DL_ONE_PASS_SCORER=0/1 forces.
    from .ops import one_pass_scorer_wanted
    want_one = one_pass_scorer_wanted(tab, shard.n_pad, K, d)
    has_one = shard.inc is not None and hasattr(backend, "score_pairs_train") and \
        backend.score_pairs_train_supported(shard.inc, K, d, tab)
    has_terms = shard._global_pairs is not None and hasattr(backend, "score_pairs_fwd_terms")
    scorer = "one_pass" if has_one and (want_one or not has_terms) else ("terms" if has_terms else None)
    if scorer is not None:
        H_loc, prob, loss = ShardedHotPathLoss.apply(Z_loc, shard, backend, float(model.beta), float(model.temperature),
                                                     group, tab, label, weight, scorer, pre)
        return H_loc.reshape(H_loc.shape[0], -1), prob, loss
    H_loc, prob = ShardedHotPath.apply(Z_loc, shard, backend, float(model.beta), float(model.temperature), group, tab, pre)
    q0, q1 = shard.pair_lo, shard.pair_hi
    y, w = label[q0:q1], weight[q0:q1]
    bce = torch.nn.functional.binary_cross_entropy(prob, y, weight=w, reduction="sum") if q1 > q0 else prob.sum() * 0.0
    return H_loc.reshape(H_loc.shape[0], -1), prob, bce


def allreduce_gradients(model, group=None) -> dict:
    """Sum the replicas' weight gradients (each rank's loss must already carry the GLOBAL normaliser) with as few
    messages and copies as the gradients' layout allows -> {"collectives": n, "copies": n, "how": ...}:

    * the drop-in module's gradients arrive as the slices of 4 stacked [K, ...] tensors that ops.project_bwd carves out
      of ONE allocation: ONE all-reduce over a flat view of it, no copy (round 3: 4 K all-reduces of 4 K small views —
      32 collectives per step on a model whose whole gradient is 3 MB);
    * stacked but not adjacent (padded feature widths): one all-reduce per stacked tensor (4);
    * anything else: the gradients are packed into one flat bucket, reduced once and copied back."""
    params = [p for p in model.parameters()]
    for prm in params:
        if prm.grad is None:
            prm.grad = torch.zeros_like(prm)
    stacked = None
    if getattr(model, "_stacked_params", None) is not None and model._stacked_params() is not None:
        from .optim import flat_view, stacked_grad
        groups = dict(model._param_groups())
        stacked = []
        for key, ps in groups.items():
            g0 = ps[0].grad
            st = stacked_grad(ps, model._stacked[key])
            if st.data_ptr() != g0.data_ptr():               # a stacking copy, not a view: not worth it, use the bucket
                stacked = None
                break
            stacked.append(st)
    if stacked:
        flat = flat_view(stacked)
        if flat is not None:
            _count("collectives")
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
            return {"collectives": 1, "copies": 0, "how": "one flat view over the stacked gradients"}
        for st in stacked:
            _count("collectives")
            dist.all_reduce(st, op=dist.ReduceOp.SUM, group=group)
        return {"collectives": len(stacked), "copies": 0, "how": "one all-reduce per stacked gradient"}
    grads = [p.grad for p in params]
    bucket = torch.cat([g.reshape(-1) for g in grads])
    _count("collectives")
    _count("staging_copies", 2)
    dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for g in grads:
        g.copy_(bucket[off:off + g.numel()].view_as(g))
        off += g.numel()
    return {"collectives": 1, "copies": 2 * len(grads), "how": "one packed bucket (gradients are not stacked views)"}
