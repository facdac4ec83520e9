// Host-side graph preparation behind the C ABI (no GPU work): the counterpart of the reference's
// dense adjacency construction (main_disentangled.py:137-142) for callers that are not Python.
// disenlink_amd/graph.py builds the same arrays with torch index ops on the device; tests check the
// two against each other.  These functions malloc their outputs (freed by the matching *_free);
// the per-epoch entry points never allocate.
#include <algorithm>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "dl_common.h"

using namespace dl;

namespace {

template <typename T>
T* dup(const std::vector<T>& v) {
    T* p = (T*)malloc(std::max<size_t>(v.size(), 1) * sizeof(T));
    if (p && !v.empty()) memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}

}  // namespace

extern "C" {

int dl_host_csr_from_edges(const int64_t* src, const int64_t* dst, int64_t n_edge_rows, int32_t n_nodes,
                           int symmetrise, dl_host_csr* out) {
    DL_REQUIRE(out != nullptr, "out is NULL");
    memset(out, 0, sizeof(*out));
    DL_REQUIRE(n_nodes >= 0 && n_edge_rows >= 0, "negative size");
    DL_REQUIRE(n_edge_rows == 0 || (src && dst), "src/dst is NULL");
    std::vector<int64_t> key;
    key.reserve((size_t)n_edge_rows * (symmetrise ? 2 : 1));
    for (int64_t r = 0; r < n_edge_rows; ++r) {
        const int64_t a = src[r], b = dst[r];
        DL_REQUIRE(a >= 0 && a < n_nodes && b >= 0 && b < n_nodes, "edge row %lld: endpoint outside [0, %d)", (long long)r,
                   n_nodes);
        key.push_back(a * n_nodes + b);
        if (symmetrise) key.push_back(b * n_nodes + a);
    }
    std::sort(key.begin(), key.end());                                   // duplicates collapse (binarised adjacency)
    key.erase(std::unique(key.begin(), key.end()), key.end());
    DL_REQUIRE(key.size() < (size_t)1 << 31, "more than 2^31-1 edges");
    const size_t E = key.size();
    std::vector<int32_t> rowptr((size_t)n_nodes + 1, 0), col(E), rev(E);
    for (size_t e = 0; e < E; ++e) {
        rowptr[(size_t)(key[e] / n_nodes) + 1]++;
        col[e] = (int32_t)(key[e] % n_nodes);
    }
    for (int32_t i = 0; i < n_nodes; ++i) rowptr[i + 1] += rowptr[i];
    for (size_t e = 0; e < E; ++e) {
        const int64_t r = key[e] / n_nodes, c = key[e] % n_nodes;
        const auto it = std::lower_bound(key.begin(), key.end(), c * n_nodes + r);
        DL_REQUIRE(it != key.end() && *it == c * n_nodes + r, "adjacency is not symmetric (reverse of (%lld,%lld) missing)",
                   (long long)r, (long long)c);
        rev[e] = (int32_t)(it - key.begin());
    }
    out->n_nodes = n_nodes;
    out->n_entries = (int32_t)E;
    out->rowptr = dup(rowptr);
    out->col = dup(col);
    out->rev = dup(rev);
    DL_REQUIRE(out->rowptr && out->col && out->rev, "out of memory");
    return DL_OK;
}

void dl_host_csr_free(dl_host_csr* c) {
    if (!c) return;
    free(c->rowptr); free(c->col); free(c->rev);
    memset(c, 0, sizeof(*c));
}

int dl_host_plan_build(int32_t n_rows, int32_t n_total, const int32_t* rowptr, const int32_t* col, int32_t seg_len,
                       int32_t n_col_slices, const uint8_t* keep, int32_t unit_segs, int32_t by_length, dl_host_plan* out) {
    DL_REQUIRE(out != nullptr, "out is NULL");
    memset(out, 0, sizeof(*out));
    DL_REQUIRE(n_rows >= 0 && n_total >= n_rows && seg_len >= 1 && n_col_slices >= 1, "bad plan size");
    DL_REQUIRE(unit_segs == 1 || unit_segs == DL_UNIT_SEGS, "unit_segs must be 1 or %d", DL_UNIT_SEGS);
    DL_REQUIRE(n_rows == 0 || rowptr, "rowptr is NULL");
    const int32_t n_streams = std::min<int32_t>(n_col_slices, 8);
    // column slices = contiguous node ranges holding equal numbers of (kept) entries: boundaries at the quantiles of
    // the kept columns, as graph.column_slices does (slice of c = number of boundaries <= c)
    std::vector<int32_t> bounds;
    if (n_col_slices > 1 && n_rows > 0) {
        std::vector<int32_t> kept;
        kept.reserve((size_t)rowptr[n_rows]);
        for (int32_t e = 0; e < rowptr[n_rows]; ++e)
            if (!keep || keep[e]) kept.push_back(col[e]);
        std::sort(kept.begin(), kept.end());
        const int64_t E = (int64_t)kept.size();
        if (E > 0)
            for (int32_t q = 1; q < n_col_slices; ++q) bounds.push_back(kept[(size_t)((int64_t)q * E / n_col_slices)]);
    }
    auto slice_of = [&](int32_t c) {
        return (int32_t)(std::upper_bound(bounds.begin(), bounds.end(), c) - bounds.begin());
    };
    // segments in entry order; a unit = up to DL_UNIT_SEGS consecutive segments of one (row, slice) group, counted
    // from the group's first segment (graph.CsrPlan.build does the same with torch index ops)
    struct Seg { int32_t row, beg, end, unit; };
    struct Unit { int32_t row, slice, first_seg, size, pad, slot; int64_t pos; int32_t entries; };
    std::vector<Seg> segs;
    std::vector<Unit> units;
    std::vector<int32_t> nunit_row((size_t)n_rows, 0);
    auto add_group = [&](int32_t row, int32_t q, int32_t b, int32_t e, bool empty_row) {
        int32_t g = 0;
        for (int32_t sb = b; sb < e || (empty_row && g == 0); sb += seg_len, ++g) {
            if (g % unit_segs == 0) {
                units.push_back({row, q, (int32_t)segs.size(), 0, 0, -1, 0, 0});
                nunit_row[row]++;
            }
            segs.push_back({row, sb, std::min(sb + seg_len, e), (int32_t)units.size() - 1});
            units.back().size++;
            units.back().entries += std::min(sb + seg_len, e) - sb;
            if (empty_row) break;
        }
    };
    for (int32_t i = 0; i < n_rows; ++i) {
        // kept entries of the row: one contiguous run (checked)
        int32_t b = rowptr[i], e = rowptr[i + 1];
        if (keep) {
            while (b < e && !keep[b]) ++b;
            int32_t e2 = b;
            while (e2 < e && keep[e2]) ++e2;
            for (int32_t x = e2; x < e; ++x) DL_REQUIRE(!keep[x], "kept entries must be contiguous inside row %d", i);
            e = e2;
        }
        int32_t pos = b;
        bool any = false;
        while (pos < e) {
            const int32_t q = n_col_slices > 1 ? slice_of(col[pos]) : 0;
            int32_t gend = pos;
            while (gend < e && (n_col_slices == 1 || slice_of(col[gend]) == q)) {
                DL_REQUIRE(gend == pos || col[gend] >= col[gend - 1] || n_col_slices == 1,
                           "sliced plans need col ascending inside every row");
                ++gend;
            }
            add_group(i, q, pos, gend, false);
            any = true;
            pos = gend;
        }
        if (!any) add_group(i, 0, rowptr[i], rowptr[i], true);   // empty rows own one empty segment
    }
    // partial slots: one per unit of a multi-unit row, consecutive per row, in entry order
    std::vector<int32_t> multi_row, multi_slot0(1, 0), row_slot0((size_t)n_rows, -1), seen((size_t)n_rows, 0);
    for (int32_t i = 0; i < n_rows; ++i)
        if (nunit_row[i] > 1) {
            row_slot0[i] = multi_slot0.back();
            multi_row.push_back(i);
            multi_slot0.push_back(multi_slot0.back() + nunit_row[i]);
        }
    for (Unit& u : units) {
        u.pad = u.size == 3 ? 4 : u.size;
        u.slot = row_slot0[u.row] < 0 ? -1 : row_slot0[u.row] + seen[u.row];
        seen[u.row]++;
    }
    // storage order: one stream per XCD (slice q -> stream q % 8), slices of a stream in time order; inside a slice
    // the units by padded size, largest first (stable), so none straddles a group of DL_UNIT_SEGS positions; by_length:
    // inside a size class the units with the most entries first (a workgroup's wavefronts finish together; for graphs
    // whose tables are cache-resident: graph.length_order)
    std::vector<int32_t> uorder(units.size());
    for (size_t x = 0; x < units.size(); ++x) uorder[x] = (int32_t)x;
    std::stable_sort(uorder.begin(), uorder.end(), [&](int32_t a, int32_t b) {
        const Unit &ua = units[a], &ub = units[b];
        const int32_t sa = ua.slice % n_streams, sb = ub.slice % n_streams;
        if (sa != sb) return sa < sb;
        if (ua.slice != ub.slice) return ua.slice < ub.slice;
        if (ua.pad != ub.pad) return ua.pad > ub.pad;
        return by_length != 0 && ua.entries > ub.entries;
    });
    std::vector<int64_t> sl_size((size_t)n_col_slices, 0), sl_base((size_t)n_col_slices, 0), sl_fill((size_t)n_col_slices, 0);
    for (const Unit& u : units) sl_size[u.slice] += u.pad;
    std::vector<int32_t> slice_seg0((size_t)n_streams + 1, 0);
    int64_t total = 0;
    int32_t max_seg = 0;
    for (int32_t x = 0; x < n_streams; ++x) {
        slice_seg0[x] = (int32_t)total;
        for (int32_t q = x; q < n_col_slices; q += n_streams) {
            sl_size[q] = (sl_size[q] + DL_UNIT_SEGS - 1) / DL_UNIT_SEGS * DL_UNIT_SEGS;
            sl_base[q] = total;
            total += sl_size[q];
        }
        max_seg = std::max<int32_t>(max_seg, (int32_t)(total - slice_seg0[x]));
    }
    DL_REQUIRE(total < (1LL << 31), "too many segment positions");
    slice_seg0[n_streams] = (int32_t)total;
    for (int32_t x : uorder) {
        Unit& u = units[x];
        u.pos = sl_base[u.slice] + sl_fill[u.slice];
        sl_fill[u.slice] += u.pad;
    }
    const size_t S = (size_t)total;
    std::vector<int32_t> seg_row(S, -1), seg_beg(S, 0), seg_end(S, 0), seg_slot(S, -1);
    for (size_t x = 0; x < segs.size(); ++x) {
        const Unit& u = units[segs[x].unit];
        const size_t p = (size_t)(u.pos + ((int64_t)x - u.first_seg));
        seg_row[p] = segs[x].row; seg_beg[p] = segs[x].beg; seg_end[p] = segs[x].end; seg_slot[p] = u.slot;
    }
    out->seg_len = seg_len; out->n_seg = (int32_t)S; out->n_slices = n_streams; out->slice_max_seg = max_seg;
    out->n_multi = (int32_t)multi_row.size(); out->n_slots = multi_slot0.back();
    out->seg_row = dup(seg_row); out->seg_beg = dup(seg_beg); out->seg_end = dup(seg_end); out->seg_slot = dup(seg_slot);
    out->slice_seg0 = dup(slice_seg0); out->multi_row = dup(multi_row); out->multi_slot0 = dup(multi_slot0);
    std::vector<int32_t> slot_multi((size_t)multi_slot0.back());           // the row (index into multi_row) of every slot
    for (size_t m = 0; m + 1 < multi_slot0.size(); ++m)
        for (int32_t sl = multi_slot0[m]; sl < multi_slot0[m + 1]; ++sl) slot_multi[(size_t)sl] = (int32_t)m;
    out->slot_multi = dup(slot_multi);
    DL_REQUIRE(out->seg_row && out->seg_beg && out->seg_end && out->seg_slot && out->slice_seg0 && out->multi_row &&
                   out->multi_slot0 && out->slot_multi, "out of memory");
    return DL_OK;
}

void dl_host_plan_free(dl_host_plan* p) {
    if (!p) return;
    free(p->seg_row); free(p->seg_beg); free(p->seg_end); free(p->seg_slot);
    free(p->slice_seg0); free(p->multi_row); free(p->multi_slot0); free(p->slot_multi);
    memset(p, 0, sizeof(*p));
}

}  // extern "C"
