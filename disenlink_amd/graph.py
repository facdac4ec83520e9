"""Graph and pair-list containers handed to libdisenlink_hip.so.

The reference keeps the training adjacency as a dense ``[N,N]`` fp32 matrix
(``main_disentangled.py:137-142``).  Here it is a CSR of the binarised, symmetrised adjacency
plus a row-segment plan that cuts skewed rows into pieces of at most ``seg_len`` entries (one
wavefront per piece).  A plan may cover only a contiguous block of rows (one shard per GPU);
column ids stay global.

All tensors are int32 and live on the device of the input.  Building is plain torch index
plumbing, done once per adjacency / pair list — it is not on the per-epoch path.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import torch

from . import _lib

DEFAULT_SEG_LEN = 32
DEFAULT_RUN_LEN = 64


def _i32(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.int32).contiguous()


@dataclass
class CsrPlan:
    """Rows [row_offset, row_offset + n_rows) of a CSR over n_total nodes + its segment plan."""
    n_rows: int
    row_offset: int
    n_total: int
    rowptr: torch.Tensor
    col: torch.Tensor
    seg_len: int
    seg_row: torch.Tensor
    seg_beg: torch.Tensor
    seg_slot: torch.Tensor
    row_seg0: torch.Tensor
    multi_row: torch.Tensor
    multi_slot0: torch.Tensor
    n_slots: int

    @property
    def n_entries(self) -> int:
        return int(self.col.numel())

    @property
    def n_seg(self) -> int:
        return int(self.seg_row.numel())

    @property
    def device(self) -> torch.device:
        return self.rowptr.device

    @staticmethod
    def build(rowptr: torch.Tensor, col: torch.Tensor, n_total: int, row_offset: int = 0,
              seg_len: int = DEFAULT_SEG_LEN) -> "CsrPlan":
        if seg_len < 1:
            raise ValueError("seg_len must be >= 1")
        rowptr = rowptr.to(torch.int64)
        n_rows = int(rowptr.numel()) - 1
        if row_offset < 0 or row_offset + n_rows > n_total:
            raise ValueError("row block outside [0, n_total)")
        dev = rowptr.device
        deg = rowptr[1:] - rowptr[:-1]
        nseg_row = torch.clamp((deg + seg_len - 1) // seg_len, min=1)
        row_seg0 = torch.zeros(n_rows + 1, dtype=torch.int64, device=dev)
        row_seg0[1:] = torch.cumsum(nseg_row, dim=0)
        seg_row = torch.repeat_interleave(torch.arange(n_rows, device=dev), nseg_row)
        seg_idx = torch.arange(seg_row.numel(), device=dev) - row_seg0[seg_row]
        seg_beg = rowptr[seg_row] + seg_idx * seg_len
        multi_row = torch.nonzero(nseg_row > 1).reshape(-1)
        # partial-sum slots: only segments of multi-segment rows get one, numbered consecutively per row
        multi_slot0 = torch.zeros(multi_row.numel() + 1, dtype=torch.int64, device=dev)
        multi_slot0[1:] = torch.cumsum(nseg_row[multi_row], dim=0)
        row_slot0 = torch.full((n_rows,), -1, dtype=torch.int64, device=dev)
        row_slot0[multi_row] = multi_slot0[:-1]
        rs = row_slot0[seg_row]
        seg_slot = torch.where(rs >= 0, rs + seg_idx, rs)
        return CsrPlan(n_rows, row_offset, n_total, _i32(rowptr), _i32(col), seg_len, _i32(seg_row), _i32(seg_beg),
                       _i32(seg_slot), _i32(row_seg0), _i32(multi_row), _i32(multi_slot0), int(multi_slot0[-1]))

    def to(self, device) -> "CsrPlan":
        mv = lambda t: t.to(device)
        return CsrPlan(self.n_rows, self.row_offset, self.n_total, mv(self.rowptr), mv(self.col), self.seg_len,
                       mv(self.seg_row), mv(self.seg_beg), mv(self.seg_slot), mv(self.row_seg0), mv(self.multi_row),
                       mv(self.multi_slot0), self.n_slots)

    def c_value(self) -> _lib.DlCsrPlan:
        return _lib.DlCsrPlan(
            self.n_rows, self.row_offset, self.n_total, self.n_entries, self.rowptr.data_ptr(), self.col.data_ptr(),
            self.seg_len, self.n_seg, self.seg_row.data_ptr(), self.seg_beg.data_ptr(), self.seg_slot.data_ptr(),
            int(self.multi_row.numel()), self.n_slots, self.multi_row.data_ptr(), self.multi_slot0.data_ptr())


@dataclass
class Graph:
    plan: CsrPlan
    rev: torch.Tensor | None = None      # reverse-edge permutation (unsharded builds only; used by tests/oracle)
    _struct: _lib.DlGraph | None = field(default=None, repr=False)

    # convenience views
    n_nodes = property(lambda self: self.plan.n_total)
    n_rows = property(lambda self: self.plan.n_rows)
    row_offset = property(lambda self: self.plan.row_offset)
    n_edges = property(lambda self: self.plan.n_entries)
    n_seg = property(lambda self: self.plan.n_seg)
    rowptr = property(lambda self: self.plan.rowptr)
    col = property(lambda self: self.plan.col)
    device = property(lambda self: self.plan.device)

    # ------------------------------------------------------------------ builders
    @staticmethod
    def from_edge_rows(src: torch.Tensor, dst: torch.Tensor, n_nodes: int, symmetrise: bool = True,
                       seg_len: int = DEFAULT_SEG_LEN, row_range: tuple[int, int] | None = None) -> "Graph":
        """Directed edge rows (duplicates allowed) -> CSR of the binarised adjacency.

        ``symmetrise=True`` reproduces ``adj_sym = (adj + adj.T) != 0`` (main_disentangled.py:141-142).
        ``row_range=(lo, hi)`` keeps only rows lo..hi-1 (one shard); column ids stay global.
        """
        if n_nodes < 0 or n_nodes >= 2 ** 31:
            raise ValueError(f"n_nodes={n_nodes} out of int32 range")
        src = src.reshape(-1).to(torch.int64)
        dst = dst.reshape(-1).to(torch.int64)
        if src.numel() != dst.numel():
            raise ValueError("src and dst differ in length")
        if src.numel() and (int(src.min()) < 0 or int(dst.min()) < 0 or
                            int(src.max()) >= n_nodes or int(dst.max()) >= n_nodes):
            raise ValueError("edge endpoint outside [0, n_nodes)")
        if symmetrise:
            src, dst = torch.cat([src, dst]), torch.cat([dst, src])
        key = torch.unique(src * n_nodes + dst)            # sorted, duplicates collapsed
        if key.numel() >= 2 ** 31:
            raise ValueError("more than 2^31-1 edges")
        r = torch.div(key, n_nodes, rounding_mode="floor")
        c = key - r * n_nodes
        tkey = c * n_nodes + r
        rev = torch.searchsorted(key, tkey)
        if key.numel():
            ok = (rev < key.numel()) & (key[rev.clamp(max=key.numel() - 1)] == tkey)
            if not bool(ok.all()):
                raise ValueError("adjacency is not symmetric (reverse edge missing); pass symmetrise=True")
        lo, hi = (0, n_nodes) if row_range is None else row_range
        if not (0 <= lo <= hi <= n_nodes):
            raise ValueError("row_range outside [0, n_nodes]")
        counts = torch.bincount(r, minlength=n_nodes) if key.numel() else torch.zeros(n_nodes, dtype=torch.int64,
                                                                                      device=key.device)
        full_ptr = torch.zeros(n_nodes + 1, dtype=torch.int64, device=key.device)
        full_ptr[1:] = torch.cumsum(counts, dim=0)
        e0, e1 = int(full_ptr[lo]), int(full_ptr[hi])
        plan = CsrPlan.build(full_ptr[lo:hi + 1] - e0, c[e0:e1], n_nodes, row_offset=lo, seg_len=seg_len)
        return Graph(plan, _i32(rev) if row_range is None else None)

    @staticmethod
    def from_dense(adj: torch.Tensor, seg_len: int = DEFAULT_SEG_LEN) -> "Graph":
        """Dense ``adj_sym`` as the reference passes it to ``model(x, adj_sym)`` (main_disentangled.py:194)."""
        if adj.dim() != 2 or adj.shape[0] != adj.shape[1]:
            raise ValueError("adj must be square")
        nz = torch.nonzero(adj)
        return Graph.from_edge_rows(nz[:, 0], nz[:, 1], adj.shape[0], symmetrise=False, seg_len=seg_len)

    def to(self, device) -> "Graph":
        return Graph(self.plan.to(device), None if self.rev is None else self.rev.to(device))

    def c_struct(self):
        if self._struct is None:
            self._struct = _lib.DlGraph(self.plan.c_value())
        return C.byref(self._struct)

    def c_plan(self):
        self.c_struct()
        return C.byref(self._struct.csr)


@dataclass
class PairList:
    """Scored pairs ``(pu[q], pv[q])``, the runs of equal ``pu`` the forward scorer stages in LDS, and
    the node-incidence plan the backward walks (rows = nodes ``[row_offset, row_offset+n_rows)``)."""
    n_nodes: int
    pu: torch.Tensor
    pv: torch.Tensor
    run_ptr: torch.Tensor
    inc: CsrPlan
    inc_pair: torch.Tensor
    _struct: _lib.DlPairIncidence | None = field(default=None, repr=False)

    @property
    def n_pairs(self) -> int:
        return int(self.pu.numel())

    @property
    def n_runs(self) -> int:
        return int(self.run_ptr.numel()) - 1

    @staticmethod
    def build(pu: torch.Tensor, pv: torch.Tensor, n_nodes: int, seg_len: int = DEFAULT_SEG_LEN,
              run_len: int = DEFAULT_RUN_LEN, row_range: tuple[int, int] | None = None,
              n_pairs_total: int | None = None) -> "PairList":
        """``row_range`` restricts the incidence rows to one shard's nodes; the pair ids in ``inc_pair``
        then index prob / g_prob arrays of length ``n_pairs_total`` (all shards' pairs)."""
        pu = pu.reshape(-1).to(torch.int64)
        pv = pv.reshape(-1).to(torch.int64)
        if pu.numel() != pv.numel():
            raise ValueError("pu and pv differ in length")
        P = pu.numel()
        if 2 * P >= 2 ** 31:
            raise ValueError("too many pairs for int32 incidence")
        if P and (int(torch.minimum(pu.min(), pv.min())) < 0 or int(torch.maximum(pu.max(), pv.max())) >= n_nodes):
            raise ValueError("pair endpoint outside [0, n_nodes)")
        dev = pu.device
        # runs of consecutive pairs with equal pu, cut into chunks of <= run_len pairs
        if P:
            change = torch.ones(P, dtype=torch.bool, device=dev)
            change[1:] = pu[1:] != pu[:-1]
            start_of = torch.cummax(torch.where(change, torch.arange(P, device=dev), 0), dim=0).values
            cut = change | (((torch.arange(P, device=dev) - start_of) % run_len) == 0)
            run_ptr = torch.cat([torch.nonzero(cut).reshape(-1), torch.tensor([P], device=dev)])
        else:
            run_ptr = torch.zeros(1, dtype=torch.int64, device=dev)
        # node-incidence CSR
        lo, hi = (0, n_nodes) if row_range is None else row_range
        node = torch.cat([pu, pv])
        other = torch.cat([pv, pu])
        pair = torch.arange(P, device=dev).repeat(2)
        keep = (node >= lo) & (node < hi)
        node, other, pair = node[keep], other[keep], pair[keep]
        order = torch.sort(node, stable=True).indices     # fixed order -> bitwise reproducible sums
        rowptr = torch.zeros(hi - lo + 1, dtype=torch.int64, device=dev)
        if node.numel():
            rowptr[1:] = torch.cumsum(torch.bincount(node - lo, minlength=hi - lo), dim=0)
        inc = CsrPlan.build(rowptr, other[order], n_nodes, row_offset=lo, seg_len=seg_len)
        return PairList(n_nodes, _i32(pu), _i32(pv), _i32(run_ptr), inc, _i32(pair[order]))

    def c_struct(self, n_pairs_total: int | None = None):
        if self._struct is None:
            self._struct = _lib.DlPairIncidence(self.inc.c_value(), self.inc_pair.data_ptr(),
                                                self.n_pairs if n_pairs_total is None else n_pairs_total)
        return C.byref(self._struct)

    def c_plan(self):
        self.c_struct()
        return C.byref(self._struct.csr)
