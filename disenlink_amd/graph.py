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

import os
import ctypes as C
from dataclasses import dataclass, field

import torch

from . import _lib

DEFAULT_SEG_LEN = int(os.environ.get("DL_SEG_LEN", "32"))        # adjacency rows (DL_SEG_LEN: experiments only)
# routing plan (no per-row reduction, hence no partials: shorter segments only add waves in flight — squirrel route
# phase 46.6 -> 42.9 us, real chameleon 33.4 -> 25.2, Penn94-sized unchanged); 0 = the adjacency plan's
DEFAULT_ROUTE_SEG_LEN = int(os.environ.get("DL_ROUTE_SEG_LEN", "0"))      # 0 = route_seg_len() decides (16, or 8 on small graphs)


def route_seg_len(n_entries: int) -> int:
    """Entries per wavefront of the routing plan (its kernel sums nothing across a row, so this is a pure grain choice;
    the per-entry results do not depend on it).  16 on graphs that fill the chip; 8 where 16 would leave fewer segments
    than the 4,096 wavefront slots of the 256 CUs (chameleon: routing phase 15.5 -> 13.6 us, Cora 14.3 -> 11.5)."""
    if DEFAULT_ROUTE_SEG_LEN:
        return DEFAULT_ROUTE_SEG_LEN
    return 16 if n_entries >= 16 * 4096 else 8
DEFAULT_INC_SEG_LEN = 64    # pair-incidence rows (backward of the scorer)
DEFAULT_RUN_LEN = 64        # pairs-by-first-endpoint rows (forward scorer)
DEFAULT_SLICES = 8          # XCDs of an MI355X
UNIT_SEGS = 4               # wavefronts per workgroup = segment positions per workgroup (DL_UNIT_SEGS)
L2_BYTES_PER_XCD = 4 << 20


CACHE_BYTES = 256 << 20     # Infinity Cache of an MI355X


def length_order(n_nodes: int, row_bytes: int, n_tables: int = 2) -> bool:
    """Whether a plan over these tables should place its units by LENGTH (longest first) inside a slice instead of
    in entry order.  A workgroup's four wavefronts end together (its LDS and its barrier), so four segments of 3, 60, 7
    and 31 entries leave most of the group idle; side by side by length they finish together, and longest-first is the
    better schedule for the tail.  The price is that consecutive workgroups no longer walk consecutive rows: where the
    row streams come from HBM that costs more than it gains (snap-patents-shaped, 3 GB: route / aggregate +11 %), where
    the gathered tables (Z and H: n_tables * n_nodes * row_bytes) fit the Infinity Cache it is free — real squirrel
    step 220 -> 208 us, chameleon 72 -> 62; squirrel-shaped graphs up to 170 MB of tables -5 %, 340 MB and up: even.
    Results do not depend on it (a unit's segments and its slot stay what they are).  DL_PLAN_SORT=0/1 forces it."""
    forced = os.environ.get("DL_PLAN_SORT")
    if forced is not None and forced != "":
        return forced != "0"
    return float(n_nodes) * row_bytes * n_tables <= CACHE_BYTES


def auto_slices(n_nodes: int, row_bytes: int, entries_per_row: float = 1e9, n_tables: int = 2) -> int:
    """Number of column slices of an XCD-aware plan (a multiple of 8: slice q belongs to XCD stream q % 8,
    the slices of a stream follow each other in time).  Aim: one slice of the gathered tables (Z and H rows
    of n_nodes / n_slices nodes) near an XCD's 4 MiB L2, while a (row, slice) group keeps >= ~4 entries so
    that the per-segment staging of the row stays amortised.  Measured (tools/): squirrel
    483 -> 196 us with 8 slices; a 41.6k-node table (170 MB) 600 -> 357 us with 32."""
    import os
    forced = os.environ.get("DL_FORCE_SLICES")               # experiments only
    if forced:
        return int(forced)
    table = float(n_nodes) * row_bytes * n_tables
    # small tables: the FEWEST slices that put a slice inside an L2 (round 6, wave-per-entry scorer: chameleon — 9.3 MB —
    # 34.0 / 32.4 / 34.3 / 40.9 us at 2 / 4 / 8 / 16 slices, profiles/r7q_fwd_slices.txt; squirrel — 21.3 MB — needs the 8)
    for few in (1, 2, 4):
        if table / few <= 0.7 * L2_BYTES_PER_XCD:
            return few if entries_per_row >= 4 * few else 1
    if entries_per_row < 2 * DEFAULT_SLICES:                     # rows too short to be cut 8 ways
        return 1
    t_cap = 1
    while 2 * t_cap * DEFAULT_SLICES * 4 <= entries_per_row:       # keep >= 4 entries per (row, slice)
        t_cap *= 2
    t = 1
    while t < t_cap and table / (DEFAULT_SLICES * t) > 1.5 * L2_BYTES_PER_XCD:
        t *= 2
    if table / (DEFAULT_SLICES * t) > 2 * L2_BYTES_PER_XCD:       # a slice that does not fit an L2 buys nothing and
        return 1                                                  # costs the per-(row, slice) staging (Penn94-sized,
                                                                  # bf16 tables: scorer 7.3 ms unsliced, 11.7 ms at 32)
    return DEFAULT_SLICES * t


def auto_inc_slices(n_nodes: int, row_bytes: int, entries_per_row: float, n_tables: int = 2) -> int:
    """Column slices of the INCIDENCE plan (one-pass training scorer, scorer backward).  Every (row, slice) group there
    is a unit with a partial slot — 2 K d 4 bytes written, then read by the combine launch, through memory — so slices
    cost per group what they buy per gathered row.  Measured (profiles/r7a_inc_slices_sweep.txt, r7b_train_scorer_ab.txt,
    interleaved same-process runs): squirrel (21 MB of Z + H, 388 entries per row) 379 us at 4, 363 at 8, 454 at 16;
    chameleon (9 MB, 148) 73 / 70 / 89 at 2 / 4 / 8; Penn94-sized K = 8 fp32 (170 MB, 331) 6.79 ms at 8, 5.78 at 16, 5.95
    at 32; K = 16 bf16 (340 MB) 15.4 ms unsliced, 14.5 at 16, 16.3 at 32.  The rule that picks the best of each: the
    fewest slices that put a slice of the tables inside an XCD's L2 (0.7 x 4 MiB), but never groups shorter than ~20
    entries; nothing where even that leaves slices 8x beyond the L2 (the hit rate it buys is then below ~1/8)."""
    forced = os.environ.get("DL_FORCE_INC_SLICES")                 # experiments only
    if forced:
        return int(forced)
    table = float(n_nodes) * row_bytes * n_tables
    s_len = 1
    while 2 * s_len * 20 <= entries_per_row:                        # groups of >= ~20 entries
        s_len *= 2
    s_fit = 1
    while table / s_fit > 0.7 * L2_BYTES_PER_XCD and s_fit < 256:
        s_fit *= 2
    s = max(1, min(s_fit, s_len))
    return s if table / s <= 8 * L2_BYTES_PER_XCD else 1


def slice_bounds(col: torch.Tensor, n_slices: int) -> torch.Tensor:
    """The n_slices - 1 boundaries of column_slices (quantiles of `col`): slice q = [bounds[q-1], bounds[q])."""
    E = int(col.numel())
    sorted_col = torch.sort(col).values
    cut = (torch.arange(1, n_slices, device=col.device, dtype=torch.int64) * E) // n_slices
    return sorted_col[cut]


def column_slices(col: torch.Tensor, n_slices: int) -> torch.Tensor:
    """Slice id of every entry: the column (node id) space is cut into n_slices contiguous ranges holding EQUAL
    NUMBERS OF ENTRIES (boundaries at the quantiles of `col`), not equal numbers of nodes — each XCD stream then gets
    the same amount of work whatever the degree distribution over node ids is (real squirrel: the heaviest of 8
    equal-width slices held 31 % more pairs than the average; scorer 226 -> see DESIGN.md).  Entries with the same
    column always share a slice.  dl_host_plan_build (dl_host.hip) computes the same boundaries."""
    E = int(col.numel())
    if E == 0 or n_slices <= 1:
        return torch.zeros_like(col)
    sorted_col = torch.sort(col).values
    cut = (torch.arange(1, n_slices, device=col.device, dtype=torch.int64) * E) // n_slices
    bounds = sorted_col[cut]                                   # slice q = [bounds[q-1], bounds[q])
    return torch.bucketize(col, bounds, right=True)


def _i32(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.int32).contiguous()


# Integer handles of live Graph / PairList objects (the registered operators of torch_ops.py take handles, not Python
# objects).  Held WEAKLY and assigned at construction: reading a handle is then a plain attribute access — traceable by
# torch.compile — and a caller that builds a new pair list per epoch does not accumulate GPU plans in a registry.
import itertools
import weakref

_HANDLES: "weakref.WeakValueDictionary[int, object]" = weakref.WeakValueDictionary()
_next_handle = itertools.count(1)


def _assign_handle(obj) -> None:
    obj._dl_handle = next(_next_handle)
    _HANDLES[obj._dl_handle] = obj


def by_handle(handle: int, kind):
    obj = _HANDLES.get(int(handle))
    return obj if isinstance(obj, kind) else None


@dataclass
class CsrPlan:
    """Rows [row_offset, row_offset + n_rows) of a CSR over n_total nodes + its segment plan.

    ``n_slices > 1`` makes the plan XCD-aware: the column space is cut into ``n_slices`` node ranges of equal entry
    count, no segment spans two ranges (needs ``col`` ascending inside each row) and segments are stored slice-major
    (``slice_seg0``).

    The ``seg_*`` arrays are indexed by POSITION: a workgroup of ``UNIT_SEGS`` wavefronts serves ``UNIT_SEGS``
    consecutive positions, and the segments of a row are placed in aligned runs (units) of at most ``UNIT_SEGS`` that
    the workgroup sums on chip; ``seg_row == -1`` marks a padding position, ``seg_slot`` is the partial slot of the
    segment's unit (-1: the row is a single unit and needs no partials).  See include/disenlink_hip.h."""
    n_rows: int
    row_offset: int
    n_total: int
    rowptr: torch.Tensor
    col: torch.Tensor
    seg_len: int
    seg_row: torch.Tensor
    seg_beg: torch.Tensor
    seg_end: torch.Tensor
    seg_slot: torch.Tensor
    n_slices: int
    slice_max_seg: int
    slice_seg0: torch.Tensor
    multi_row: torch.Tensor
    multi_slot0: torch.Tensor
    n_slots: int
    slot_multi: torch.Tensor | None = None      # [n_slots] index (into multi_row) of the row a partial slot belongs to
    unit_count: torch.Tensor | None = None      # [n_multi] zeroed int32, owned by the plan: the in-launch row sums count
                                                # the units of a row here and leave it all zero again (disenlink_hip.h)

    def __post_init__(self):
        dev = self.multi_slot0.device
        if self.slot_multi is None:
            n = self.multi_slot0[1:] - self.multi_slot0[:-1]
            self.slot_multi = torch.repeat_interleave(torch.arange(n.numel(), device=dev, dtype=torch.int32), n.long())
        if self.unit_count is None:
            self.unit_count = torch.zeros(int(self.multi_row.numel()), dtype=torch.int32, device=dev)

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
              seg_len: int = DEFAULT_SEG_LEN, n_slices: int = 1, keep: torch.Tensor | None = None,
              unit_segs: int = UNIT_SEGS, by_length: bool = False, bounds: torch.Tensor | None = None) -> "CsrPlan":
        """``keep`` (bool per entry, optional): segments cover only the kept entries, which must form one
        contiguous run inside every row (e.g. the upper triangle ``col >= row`` of a sorted row).
        ``unit_segs = 1``: every segment is its own unit — for plans whose kernels reduce nothing across segments
        (routing, the forward scorer): positions then simply follow the entry order.
        ``bounds`` (optional, n_slices - 1 ascending node ids): the column slices' boundaries given from outside instead of
        the quantiles of THIS plan's columns — a row shard passes the whole list's boundaries, so that its rows are cut
        into exactly the units the unsharded plan cuts them into (the units of a row are summed in slot order: same bits)."""
        if unit_segs not in (1, UNIT_SEGS):
            raise ValueError(f"unit_segs must be 1 or {UNIT_SEGS}")
        if seg_len < 1 or n_slices < 1:
            raise ValueError("seg_len and n_slices must be >= 1")
        rowptr = rowptr.to(torch.int64)
        col = col.to(torch.int64)
        n_rows = int(rowptr.numel()) - 1
        if row_offset < 0 or row_offset + n_rows > n_total:
            raise ValueError("row block outside [0, n_total)")
        dev = rowptr.device
        E_all = int(col.numel())
        deg_all = rowptr[1:] - rowptr[:-1]
        ar = lambda n: torch.arange(n, device=dev)
        # groups = maximal entry ranges with equal (row, column slice); entries are sorted by (row, col)
        row_all = torch.repeat_interleave(ar(n_rows), deg_all)
        if keep is None:
            orig, row_of, kcol = ar(E_all), row_all, col
        else:
            orig = torch.nonzero(keep.to(dev)).reshape(-1)          # original entry index of every kept entry
            row_of, kcol = row_all[orig], col[orig]
        E = int(orig.numel())
        deg = torch.bincount(row_of, minlength=n_rows) if E else torch.zeros(n_rows, dtype=torch.int64, device=dev)
        if n_slices > 1 and bounds is not None:
            if int(bounds.numel()) != n_slices - 1:
                raise ValueError("bounds must hold n_slices - 1 boundaries")
            sl_of = torch.bucketize(kcol, bounds.to(kcol.device), right=True)
        else:
            sl_of = column_slices(kcol, n_slices) if n_slices > 1 else 0
        gid = row_of * n_slices + sl_of
        if E and n_slices > 1 and bool((gid[1:] < gid[:-1]).any()):
            raise ValueError("sliced plans need col ascending inside every row")
        new = torch.ones(E, dtype=torch.bool, device=dev)
        if E:
            new[1:] = gid[1:] != gid[:-1]
            if keep is not None and bool(((orig[1:] != orig[:-1] + 1) & ~new[1:]).any()):
                raise ValueError("kept entries must be contiguous inside every row")
        gpos = torch.nonzero(new).reshape(-1)                       # positions in the kept list
        gpos_end = torch.cat([gpos[1:], torch.tensor([E], device=dev)]) if E else gpos
        gstart = orig[gpos] if E else gpos                          # ... and as original entry indices
        gend = (orig[gpos_end - 1] + 1) if E else gpos
        g_id = gid[gpos] if E else gpos
        nch = (gend - gstart + seg_len - 1) // seg_len
        ch0 = torch.cumsum(nch, 0) - nch
        seg_g = torch.repeat_interleave(ar(gstart.numel()), nch)
        seg_beg = gstart[seg_g] + (ar(seg_g.numel()) - ch0[seg_g]) * seg_len
        seg_end = torch.minimum(seg_beg + seg_len, gend[seg_g])
        seg_row = torch.div(g_id[seg_g], n_slices, rounding_mode="floor")
        seg_slice = g_id[seg_g] - seg_row * n_slices
        # empty rows still own one (empty) segment: their outputs must be written
        empty = torch.nonzero(deg == 0).reshape(-1)
        seg_row = torch.cat([seg_row, empty])
        seg_beg = torch.cat([seg_beg, rowptr[empty]])
        seg_end = torch.cat([seg_end, rowptr[empty]])
        seg_slice = torch.cat([seg_slice, torch.zeros_like(empty)])
        # entry order: (row, first entry); a (row, slice) group is a run of this order
        order = torch.argsort(seg_row * (E_all + 1) + seg_beg, stable=True)
        seg_row, seg_beg, seg_end, seg_slice = seg_row[order], seg_beg[order], seg_end[order], seg_slice[order]
        S = int(seg_row.numel())
        # UNITS: up to UNIT_SEGS consecutive segments of one (row, slice) group, counted from the group's first
        # segment — so the chunking of a row depends on that row alone (a shard cuts its rows exactly like the whole
        # graph does: results stay bitwise independent of the sharding).  A workgroup (UNIT_SEGS wavefronts) serves
        # UNIT_SEGS consecutive POSITIONS of a slice stream and sums each unit on chip; only rows with more than one
        # unit go through partial slots (one per UNIT, not per segment).
        grp_key = seg_row * n_slices + seg_slice
        gnew = torch.ones(S, dtype=torch.bool, device=dev)
        gnew[1:] = grp_key[1:] != grp_key[:-1]
        gstart_pos = torch.nonzero(gnew).reshape(-1)
        g_of = torch.cumsum(gnew, 0) - 1
        idx_in_grp = ar(S) - gstart_pos[g_of]
        unew = gnew | (idx_in_grp % unit_segs == 0)
        u_of = torch.cumsum(unew, 0) - 1                          # unit of every segment, units in entry order
        ufirst = torch.nonzero(unew).reshape(-1)
        n_units = int(ufirst.numel())
        u_size = torch.bincount(u_of, minlength=n_units)
        u_pad = torch.where(u_size == 3, torch.full_like(u_size, 4), u_size)   # 1, 2, 4: aligned when placed largest first
        u_row, u_slice = seg_row[ufirst], seg_slice[ufirst]
        # partial slots: one per unit of a multi-unit row, consecutive per row in entry order
        nunit_row = torch.bincount(u_row, minlength=n_rows)
        multi_row = torch.nonzero(nunit_row > 1).reshape(-1)
        multi_slot0 = torch.zeros(multi_row.numel() + 1, dtype=torch.int64, device=dev)
        multi_slot0[1:] = torch.cumsum(nunit_row[multi_row], dim=0)
        row_slot0 = torch.full((n_rows,), -1, dtype=torch.int64, device=dev)
        row_slot0[multi_row] = multi_slot0[:-1]
        row_unit0 = torch.cumsum(nunit_row, 0) - nunit_row
        u_idx_in_row = ar(n_units) - row_unit0[u_row]
        u_slot = torch.where(row_slot0[u_row] >= 0, row_slot0[u_row] + u_idx_in_row, row_slot0[u_row])
        # storage: one STREAM of positions per XCD.  Column slice q belongs to stream q % 8 and the slices of a stream
        # follow each other in time (q // 8), so an XCD works on one slice at a time.  Inside a slice the units are placed
        # by padded size, largest first (stable: entry order among equals — or, with by_length, most entries first: see
        # length_order), so no unit straddles a group of UNIT_SEGS positions; every slice region is padded to a multiple
        # of UNIT_SEGS.  Pad positions have seg_row = -1.
        n_streams = min(n_slices, DEFAULT_SLICES)
        u_stream = u_slice % n_streams
        key = (u_stream * (n_slices + 1) + u_slice) * 8 + (UNIT_SEGS - u_pad)
        if by_length and n_units:                                  # ... and by entries inside a size class, longest first
            u_entries = torch.zeros(n_units, dtype=torch.int64, device=dev).index_add_(0, u_of, seg_end - seg_beg)
            span = int(seg_len) * UNIT_SEGS
            key = key * (span + 1) + (span - u_entries)
        uperm = torch.argsort(key, stable=True)
        sl_size = torch.zeros(n_slices, dtype=torch.int64, device=dev).index_add_(0, u_slice, u_pad)
        sl_size = (sl_size + UNIT_SEGS - 1) // UNIT_SEGS * UNIT_SEGS
        sl_order = torch.argsort((ar(n_slices) % n_streams) * (n_slices + 1) + ar(n_slices), stable=True)   # (stream, slice)
        sl_base = torch.zeros(n_slices, dtype=torch.int64, device=dev)
        sl_base[sl_order] = torch.cumsum(sl_size[sl_order], 0) - sl_size[sl_order]
        pad_sorted = u_pad[uperm]
        run = torch.cumsum(pad_sorted, 0) - pad_sorted             # exclusive cumsum over all units in storage order ...
        sl_sorted = u_slice[uperm]
        first_of_slice = torch.ones(n_units, dtype=torch.bool, device=dev)
        first_of_slice[1:] = sl_sorted[1:] != sl_sorted[:-1]
        run0 = run[torch.nonzero(first_of_slice).reshape(-1)]      # ... minus its value at the slice's first unit
        slice_rank = torch.cumsum(first_of_slice, 0) - 1
        u_pos = torch.empty(n_units, dtype=torch.int64, device=dev)
        u_pos[uperm] = sl_base[sl_sorted] + run - run0[slice_rank]
        n_pos = int(sl_size.sum())
        pos = u_pos[u_of] + (ar(S) - ufirst[u_of])
        p_row = torch.full((n_pos,), -1, dtype=torch.int64, device=dev)
        p_beg = torch.zeros(n_pos, dtype=torch.int64, device=dev)
        p_end = torch.zeros(n_pos, dtype=torch.int64, device=dev)
        p_slot = torch.full((n_pos,), -1, dtype=torch.int64, device=dev)
        p_row[pos], p_beg[pos], p_end[pos], p_slot[pos] = seg_row, seg_beg, seg_end, u_slot[u_of]
        stream_size = torch.zeros(n_streams, dtype=torch.int64, device=dev).index_add_(0, ar(n_slices) % n_streams, sl_size)
        slice_seg0 = torch.zeros(n_streams + 1, dtype=torch.int64, device=dev)
        slice_seg0[1:] = torch.cumsum(stream_size, 0)
        return CsrPlan(n_rows, row_offset, n_total, _i32(rowptr), _i32(col), seg_len, _i32(p_row),
                       _i32(p_beg), _i32(p_end), _i32(p_slot), n_streams,
                       int(stream_size.max()), _i32(slice_seg0),
                       _i32(multi_row), _i32(multi_slot0), int(multi_slot0[-1]))

    def to(self, device) -> "CsrPlan":
        mv = lambda t: t.to(device)
        return CsrPlan(self.n_rows, self.row_offset, self.n_total, mv(self.rowptr), mv(self.col), self.seg_len,
                       mv(self.seg_row), mv(self.seg_beg), mv(self.seg_end), mv(self.seg_slot), self.n_slices,
                       self.slice_max_seg, mv(self.slice_seg0), mv(self.multi_row), mv(self.multi_slot0), self.n_slots)

    def c_value(self) -> _lib.DlCsrPlan:
        return _lib.DlCsrPlan(
            self.n_rows, self.row_offset, self.n_total, self.n_entries, self.rowptr.data_ptr(), self.col.data_ptr(),
            self.seg_len, self.n_seg, self.seg_row.data_ptr(), self.seg_beg.data_ptr(), self.seg_end.data_ptr(),
            self.seg_slot.data_ptr(), self.n_slices, self.slice_max_seg, self.slice_seg0.data_ptr(),
            int(self.multi_row.numel()), self.n_slots, self.multi_row.data_ptr(), self.multi_slot0.data_ptr(),
            self.slot_multi.data_ptr(), self.unit_count.data_ptr())


@dataclass
class Graph:
    plan: CsrPlan
    rev: torch.Tensor | None = None      # reverse-edge permutation (unsharded builds only)
    route: CsrPlan | None = None         # plan walked by the routing kernel: XCD-sliced, and (unsharded builds)
    route_mirror: bool = False           # covering col >= row only — routing is symmetric, so each undirected
                                         # edge is computed once and written to both entries through rev
    _struct: _lib.DlGraph | None = field(default=None, repr=False)

    def __post_init__(self):
        _assign_handle(self)

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
                       seg_len: int = DEFAULT_SEG_LEN, row_range: tuple[int, int] | None = None,
                       row_bytes: int = 2048) -> "Graph":
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
        by_len = length_order(n_nodes, row_bytes)
        plan = CsrPlan.build(full_ptr[lo:hi + 1] - e0, c[e0:e1], n_nodes, row_offset=lo, seg_len=seg_len, by_length=by_len)
        mirror = row_range is None
        lc, lr = c[e0:e1], r[e0:e1] - lo
        # XCD slicing of the routing plan was measured and rejected: hub rows already give the Z gathers a high
        # L2 hit rate, and the extra segments cost more than they save (squirrel 50 -> 72 us; 41.6k-node shard
        # 115 -> 107 us; re-measured in round 6 on the mirrored 22 us kernel: 31.5 / 31.1 / 32.5 / 35.0 us by events at
        # 1 / 2 / 4 / 8 slices).  `row_bytes` is kept for callers that pass the model shape.
        route = CsrPlan.build(full_ptr[lo:hi + 1] - e0, lc, n_nodes, row_offset=lo,
                              seg_len=min(seg_len, route_seg_len((e1 - e0 + 1) // 2 if mirror else e1 - e0)), n_slices=1,
                              keep=(lc >= lr + lo) if mirror else None, unit_segs=1, by_length=by_len)
        route.rowptr, route.col = plan.rowptr, plan.col        # the SAME arrays: only the segments differ
        return Graph(plan, _i32(rev) if mirror else None, route, mirror)

    @staticmethod
    def from_dense(adj: torch.Tensor, seg_len: int = DEFAULT_SEG_LEN, row_bytes: int = 2048) -> "Graph":
        """Dense ``adj_sym`` as the reference passes it to ``model(x, adj_sym)`` (main_disentangled.py:194)."""
        if adj.dim() != 2 or adj.shape[0] != adj.shape[1]:
            raise ValueError("adj must be square")
        # The reference multiplies by adj (model.py:62: p * adj, compared with the factor numbers): its layer is only
        # meaningful for entries that are 0 or 1.  Anything else would be an edge here and garbage there: refuse it.
        if adj.layout != torch.strided:                            # sparse COO / CSR: the stored entries (an extension)
            coo = adj.to_sparse_coo().coalesce()
            val, idx = coo.values(), coo.indices()
            if bool(((val != 0) & (val != 1)).any()):
                raise ValueError("adj must hold 0 / 1 entries (the reference's layer multiplies by it, model.py:62)")
            keep = val != 0
            return Graph.from_edge_rows(idx[0][keep], idx[1][keep], adj.shape[0], symmetrise=False, seg_len=seg_len,
                                        row_bytes=row_bytes)
        nz = torch.nonzero(adj)
        if nz.shape[0] and bool((adj[nz[:, 0], nz[:, 1]] != 1).any()):
            raise ValueError("adj must hold 0 / 1 entries (the reference's layer multiplies by it, model.py:62)")
        return Graph.from_edge_rows(nz[:, 0], nz[:, 1], adj.shape[0], symmetrise=False, seg_len=seg_len,
                                    row_bytes=row_bytes)

    def to(self, device) -> "Graph":
        plan = self.plan.to(device)
        route = None if self.route is None else self.route.to(device)
        if route is not None:
            route.rowptr, route.col = plan.rowptr, plan.col
        return Graph(plan, None if self.rev is None else self.rev.to(device), route, self.route_mirror)

    def c_struct(self):
        if self._struct is None:
            rp = self.route.c_value() if self.route is not None else _lib.DlCsrPlan()
            self._struct = _lib.DlGraph(self.plan.c_value(), rp, self.rev.data_ptr() if self.rev is not None else None,
                                        1 if self.route_mirror else 0)
        return C.byref(self._struct)

    def c_plan(self):
        self.c_struct()
        return C.byref(self._struct.csr)


@dataclass
class PairList:
    """Scored pairs ``(pu[q], pv[q])`` with the two CSR views the kernels walk:
    ``by_u`` — every pair once, in the row of its first endpoint (forward scorer), and
    ``inc``  — every pair twice, once per endpoint (backward; rows = nodes ``[row_offset, +n_rows)``).
    Both are XCD-sliced by the second endpoint by default."""
    n_nodes: int
    pu: torch.Tensor
    pv: torch.Tensor
    by_u: CsrPlan
    by_u_pair: torch.Tensor
    inc: CsrPlan
    inc_pair: torch.Tensor
    _struct: _lib.DlPairIncidence | None = field(default=None, repr=False)
    _struct_u: _lib.DlPairIncidence | None = field(default=None, repr=False)
    _yw: torch.Tensor | None = field(default=None, repr=False)      # per-entry (label, signed weight): bind_labels
    _yw_key: tuple | None = field(default=None, repr=False)
    _yw_seen: tuple | None = field(default=None, repr=False)

    def __post_init__(self):
        _assign_handle(self)

    def bind_labels(self, label: torch.Tensor, weight: torch.Tensor, n_pairs_total: int | None = None) -> None:
        """Lay the labels / loss weights of a training step out per incidence entry (dl_pair_incidence.entry_yw: a
        coalesced stream instead of two random reads per entry; the weight's sign picks the one entry of a pair that
        writes prob) — for the label and weight tensors the loop passes EVERY epoch (main_disentangled.py:195: the masks
        are fixed for a run).  Keyed on the tensors' identity and version counter; built the second time the same pair
        of tensor OBJECTS is seen, so a caller that draws new labels for every step never pays for it — and never gets the
        stream of an earlier tensor that lived at the same address.  Writes that bypass the
        version counter (``.data``, raw kernels) are not seen: call ``unbind_labels()`` after such a write."""
        if os.environ.get("DL_ENTRY_LABELS", "1") == "0" or label.is_inference() or weight.is_inference():   # (A/B runs; no version counter)
            self.unbind_labels()
            return
        # identity = the tensor OBJECTS (weak references) and their version counters: an address can be handed to a new
        # tensor with other contents — a caller that makes its labels afresh every step gets the same data_ptr and version 0
        # again and again — an object cannot
        def same(key):
            return key is not None and key[0]() is label and key[1]() is weight and key[2:] == (label._version, weight._version)
        if not same(self._yw_key):
            key = (weakref.ref(label), weakref.ref(weight), label._version, weight._version)
            if not same(self._yw_seen):                             # first sight: keep the gathers, remember the pair
                self._yw_seen = key
                self.unbind_labels()
                return
            q = self.inc_pair.long()
            deg = (self.inc.rowptr[1:] - self.inc.rowptr[:-1]).long()
            rows = torch.repeat_interleave(torch.arange(self.inc.n_rows, device=q.device), deg) + self.inc.row_offset
            first = rows == self.pu.long()[q]                        # the entry in the row of the pair's first endpoint
            w = weight.reshape(-1).float()[q]
            self._yw = torch.stack([label.reshape(-1).float()[q], torch.where(first, w, -w)], dim=1).contiguous()
            self._yw_key = key
        self.c_struct(n_pairs_total)
        self._struct.entry_yw = self._yw.data_ptr()

    def unbind_labels(self) -> None:
        self._yw = self._yw_key = None
        if self._struct is not None:
            self._struct.entry_yw = None

    @property
    def n_pairs(self) -> int:
        return int(self.pu.numel())

    @staticmethod
    def build(pu: torch.Tensor, pv: torch.Tensor, n_nodes: int, seg_len: int = DEFAULT_INC_SEG_LEN,
              run_len: int = DEFAULT_RUN_LEN, row_range: tuple[int, int] | None = None,
              n_slices: int | None = None, by_u_range: tuple[int, int] | None = None,
              build_by_u: bool = True, row_bytes: int = 2048, inc_slices: int | None = None) -> "PairList":
        """``row_range`` restricts the incidence rows to one shard's nodes (the pair ids in ``inc_pair``
        then index prob / g_prob arrays covering the whole pair list); ``by_u_range`` restricts the rows
        of the forward plan (every pu must lie inside it).  ``n_slices`` / ``inc_slices``: column slices of the forward
        plan / of the incidence plan (defaults: auto_slices / auto_inc_slices — the incidence plan pays a partial slot
        per (row, slice) group, the forward plan sums nothing across segments)."""
        pu = pu.reshape(-1).to(torch.int64)
        pv = pv.reshape(-1).to(torch.int64)
        if pu.numel() != pv.numel():
            raise ValueError("pu and pv differ in length")
        P = pu.numel()
        if n_slices is None:
            n_slices = auto_slices(n_nodes, row_bytes, entries_per_row=P / max(1, len(torch.unique(pu))))
        if inc_slices is None:
            # (from the WHOLE list, whatever row_range says: a shard must cut its rows exactly as the unsharded plan does)
            inc_slices = auto_inc_slices(n_nodes, row_bytes, entries_per_row=2.0 * P / max(1, n_nodes))
        if 2 * P >= 2 ** 31:
            raise ValueError("too many pairs for int32 incidence")
        if P and (int(torch.minimum(pu.min(), pv.min())) < 0 or int(torch.maximum(pu.max(), pv.max())) >= n_nodes):
            raise ValueError("pair endpoint outside [0, n_nodes)")
        dev = pu.device
        ids = torch.arange(P, device=dev)

        def csr(node, other, pair, lo, hi, seg, unit_segs=UNIT_SEGS, slices=None):
            slices = n_slices if slices is None else slices
            # slice boundaries from ALL entries of the list (before the row range cuts it): the same for every shard
            bnd = slice_bounds(other, slices) if slices > 1 and other.numel() else None
            keep = (node >= lo) & (node < hi)
            node, other, pair = node[keep], other[keep], pair[keep]
            order = torch.argsort((node - lo) * n_nodes + other, stable=True)   # fixed order -> reproducible sums
            rowptr = torch.zeros(hi - lo + 1, dtype=torch.int64, device=dev)
            if node.numel():
                rowptr[1:] = torch.cumsum(torch.bincount(node - lo, minlength=hi - lo), dim=0)
            plan = CsrPlan.build(rowptr, other[order], n_nodes, row_offset=lo, seg_len=seg, n_slices=slices,
                                 unit_segs=unit_segs, by_length=length_order(n_nodes, row_bytes), bounds=bnd)
            return plan, _i32(pair[order])

        ulo, uhi = (0, n_nodes) if by_u_range is None else by_u_range
        if P and build_by_u and (int(pu.min()) < ulo or int(pu.max()) >= uhi):
            raise ValueError("a first endpoint lies outside by_u_range")
        if build_by_u:
            by_u, by_u_pair = csr(pu, pv, ids, ulo, uhi, run_len, unit_segs=1)     # the forward scorer sums nothing across segments
        else:                                              # backward-only list (sharded runs): empty forward plan
            by_u, by_u_pair = csr(pu[:0], pv[:0], ids[:0], 0, 0, run_len, unit_segs=1)
        lo, hi = (0, n_nodes) if row_range is None else row_range
        inc, inc_pair = csr(torch.cat([pu, pv]), torch.cat([pv, pu]), ids.repeat(2), lo, hi, seg_len, slices=inc_slices)
        return PairList(n_nodes, _i32(pu), _i32(pv), by_u, by_u_pair, inc, inc_pair)

    def c_struct(self, n_pairs_total: int | None = None):
        if self._struct is None:
            self._struct = _lib.DlPairIncidence(self.inc.c_value(), self.inc_pair.data_ptr(),
                                                self.n_pairs if n_pairs_total is None else n_pairs_total)
        return C.byref(self._struct)

    def c_struct_by_u(self):
        if self._struct_u is None:
            self._struct_u = _lib.DlPairIncidence(self.by_u.c_value(), self.by_u_pair.data_ptr(), self.n_pairs)
        return C.byref(self._struct_u)

    def c_plan(self):
        self.c_struct()
        return C.byref(self._struct.csr)
