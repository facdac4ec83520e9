"""Graph and pair-list containers handed to libdisenlink_hip.so.

The reference keeps the training adjacency as a dense ``[N,N]`` fp32 matrix
(``main_disentangled.py:137-142``).  Here it is a CSR of the binarised, symmetrised adjacency
with a reverse-edge permutation (needed by the atomic-free backward)
and a row-segment plan that cuts skewed rows into pieces of at most ``seg_len`` edges.

All tensors are int32 and live on the device of the input.  Building is plain torch index
plumbing, done once per adjacency — it is not on the per-epoch path.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import torch

from . import _lib

DEFAULT_SEG_LEN = 32


def _i32(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.int32).contiguous()


@dataclass
class Graph:
    n_nodes: int
    rowptr: torch.Tensor
    col: torch.Tensor
    rev: torch.Tensor
    seg_len: int
    seg_row: torch.Tensor
    seg_beg: torch.Tensor
    seg_slot: torch.Tensor
    row_seg0: torch.Tensor
    multi_row: torch.Tensor
    multi_slot0: torch.Tensor
    _struct: _lib.DlGraph | None = field(default=None, repr=False)

    @property
    def n_edges(self) -> int:
        return int(self.col.numel())

    @property
    def n_seg(self) -> int:
        return int(self.seg_row.numel())

    @property
    def n_slots(self) -> int:
        return int(self.multi_slot0[-1]) if self.multi_slot0.numel() else 0

    @property
    def device(self) -> torch.device:
        return self.rowptr.device

    # ------------------------------------------------------------------ builders
    @staticmethod
    def from_edge_rows(src: torch.Tensor, dst: torch.Tensor, n_nodes: int, symmetrise: bool = True,
                       seg_len: int = DEFAULT_SEG_LEN) -> "Graph":
        """Directed edge rows (duplicates allowed) -> CSR of the binarised adjacency.

        ``symmetrise=True`` reproduces ``adj_sym = (adj + adj.T) != 0`` (main_disentangled.py:141-142).
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
        rowptr = torch.zeros(n_nodes + 1, dtype=torch.int64, device=key.device)
        if key.numel():
            rowptr[1:] = torch.cumsum(torch.bincount(r, minlength=n_nodes), dim=0)
        tkey = c * n_nodes + r
        rev = torch.searchsorted(key, tkey)
        if key.numel():
            ok = (rev < key.numel()) & (key[rev.clamp(max=key.numel() - 1)] == tkey)
            if not bool(ok.all()):
                raise ValueError("adjacency is not symmetric (reverse edge missing); pass symmetrise=True")
        return Graph._finish(n_nodes, rowptr, c, rev, seg_len)

    @staticmethod
    def from_dense(adj: torch.Tensor, seg_len: int = DEFAULT_SEG_LEN) -> "Graph":
        """Dense ``adj_sym`` as the reference passes it to ``model(x, adj_sym)`` (main_disentangled.py:194)."""
        if adj.dim() != 2 or adj.shape[0] != adj.shape[1]:
            raise ValueError("adj must be square")
        nz = torch.nonzero(adj)
        return Graph.from_edge_rows(nz[:, 0], nz[:, 1], adj.shape[0], symmetrise=False, seg_len=seg_len)

    @staticmethod
    def _finish(n_nodes, rowptr, col, rev, seg_len) -> "Graph":
        if seg_len < 1:
            raise ValueError("seg_len must be >= 1")
        dev = rowptr.device
        deg = rowptr[1:] - rowptr[:-1]
        nseg_row = torch.clamp((deg + seg_len - 1) // seg_len, min=1)
        row_seg0 = torch.zeros(n_nodes + 1, dtype=torch.int64, device=dev)
        row_seg0[1:] = torch.cumsum(nseg_row, dim=0)
        seg_row = torch.repeat_interleave(torch.arange(n_nodes, device=dev), nseg_row)
        seg_idx = torch.arange(seg_row.numel(), device=dev) - row_seg0[seg_row]
        seg_beg = rowptr[seg_row] + seg_idx * seg_len
        multi_row = torch.nonzero(nseg_row > 1).reshape(-1)
        # partial-sum slots: only segments of multi-segment rows get one, numbered consecutively per row
        multi_slot0 = torch.zeros(multi_row.numel() + 1, dtype=torch.int64, device=dev)
        multi_slot0[1:] = torch.cumsum(nseg_row[multi_row], dim=0)
        row_slot0 = torch.full((n_nodes,), -1, dtype=torch.int64, device=dev)
        row_slot0[multi_row] = multi_slot0[:-1]
        seg_slot = torch.where(row_slot0[seg_row] >= 0, row_slot0[seg_row] + seg_idx, row_slot0[seg_row])
        return Graph(n_nodes, _i32(rowptr), _i32(col), _i32(rev), seg_len, _i32(seg_row), _i32(seg_beg),
                     _i32(seg_slot), _i32(row_seg0), _i32(multi_row), _i32(multi_slot0))

    def to(self, device) -> "Graph":
        mv = lambda t: t.to(device)
        return Graph(self.n_nodes, mv(self.rowptr), mv(self.col), mv(self.rev), self.seg_len, mv(self.seg_row),
                     mv(self.seg_beg), mv(self.seg_slot), mv(self.row_seg0), mv(self.multi_row),
                     mv(self.multi_slot0))

    # ------------------------------------------------------------------ C view
    def c_struct(self) -> "C.POINTER(_lib.DlGraph)":
        if self._struct is None:
            self._struct = _lib.DlGraph(
                self.n_nodes, self.n_edges, self.rowptr.data_ptr(), self.col.data_ptr(), self.rev.data_ptr(),
                self.seg_len, self.n_seg, self.seg_row.data_ptr(), self.seg_beg.data_ptr(),
                self.seg_slot.data_ptr(), self.row_seg0.data_ptr(), int(self.multi_row.numel()), self.n_slots,
                self.multi_row.data_ptr(), self.multi_slot0.data_ptr())
        return C.byref(self._struct)


@dataclass
class PairList:
    """Scored pairs ``(pu[q], pv[q])`` plus the node-incidence list the backward walks."""
    n_nodes: int
    pu: torch.Tensor
    pv: torch.Tensor
    inc_ptr: torch.Tensor
    inc_other: torch.Tensor
    inc_pair: torch.Tensor
    _struct: _lib.DlPairIncidence | None = field(default=None, repr=False)

    @property
    def n_pairs(self) -> int:
        return int(self.pu.numel())

    @staticmethod
    def build(pu: torch.Tensor, pv: torch.Tensor, n_nodes: int) -> "PairList":
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
        node = torch.cat([pu, pv])
        other = torch.cat([pv, pu])
        pair = torch.arange(P, device=dev).repeat(2)
        order = torch.sort(node, stable=True).indices     # fixed order -> bitwise reproducible sums
        inc_ptr = torch.zeros(n_nodes + 1, dtype=torch.int64, device=dev)
        if P:
            inc_ptr[1:] = torch.cumsum(torch.bincount(node, minlength=n_nodes), dim=0)
        return PairList(n_nodes, _i32(pu), _i32(pv), _i32(inc_ptr), _i32(other[order]), _i32(pair[order]))

    def c_struct(self):
        if self._struct is None:
            self._struct = _lib.DlPairIncidence(self.n_nodes, self.n_pairs, self.inc_ptr.data_ptr(),
                                                self.inc_other.data_ptr(), self.inc_pair.data_ptr())
        return C.byref(self._struct)
