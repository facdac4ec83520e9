"""CPU stand-in for the HIP backend of disenlink_amd.dist, built on the oracle (numpy).  TEST ONLY.

It lets the world_size-2 gloo tests exercise the sharding choreography (partition, padding,
collectives, autograd glue, gradient all-reduce) without a GPU.  Same contract as
``disenlink_amd.dist.HipBackend``: node-indexed outputs are written at GLOBAL row ids for the
plan's rows only; per-edge arrays are local to the plan.
"""
import numpy as np
import torch

from oracle import sparse_ref
from oracle.sparse_ref import _edge_dots, f32


def _rows(plan):
    rowptr = plan.rowptr.numpy().astype(np.int64)
    src = np.repeat(np.arange(plan.n_rows, dtype=np.int64), np.diff(rowptr)) + plan.row_offset
    return src, plan.col.numpy().astype(np.int64)


def _np(t):
    """fp32 numpy view of a table in either storage type (bf16 tables: the rounded values, as the kernels read them)"""
    return t.float().numpy()


class OracleBackend:
    partial_route_plans = True            # tests flip this to stand in for the generic kernels (which ignore the plan)

    def honours_partial_route_plans(self, K, d, table_dtype) -> bool:
        return self.partial_route_plans

    def route_fwd(self, g, Z, t, s_out, p_out=None, a_out=None):
        """Like dl_route_fwd: the entries of the graph's ROUTING plan are routed (all of them, or — Shard.route_by_peer —
        those whose column lies in one peer's block), into p_out / a_out when given; the row sums are then taken over the
        per-entry arrays as they stand."""
        src, dst = _rows(g.plan)
        Zh = _np(Z)
        K = Zh.shape[1]
        E = src.size
        p_all = p_out.numpy() if p_out is not None else np.zeros(E, np.uint8)
        a_all = a_out.numpy() if a_out is not None else np.zeros(E, f32)
        todo = np.arange(E)
        if g.route is not None:
            r = g.route
            rows, beg, end = r.seg_row.numpy(), r.seg_beg.numpy(), r.seg_end.numpy()
            todo = np.concatenate([np.arange(b, e) for rw, b, e in zip(rows, beg, end) if rw >= 0 and e > b] or
                                  [np.zeros(0, np.int64)]).astype(np.int64)
        if todo.size:
            with np.errstate(over="ignore", invalid="ignore"):
                ex = np.exp(_edge_dots(Zh, src[todo], dst[todo]) / f32(t), dtype=f32)
                alpha = ex / ex.sum(axis=1, dtype=f32, keepdims=True)
            pk = np.argmax(alpha, axis=1)
            p_all[todo] = pk.astype(np.uint8)
            a_all[todo] = alpha[np.arange(pk.size), pk].astype(f32)
        s_loc = np.zeros((g.n_rows, K), dtype=f32)
        np.add.at(s_loc, (src - g.row_offset, p_all.astype(np.int64)), a_all)
        s_out[g.row_offset:g.row_offset + g.n_rows] = torch.from_numpy(s_loc)
        return (torch.from_numpy(p_all) if p_out is None else p_out), (torch.from_numpy(a_all) if a_out is None else a_out)

    def aggregate_fwd(self, g, Z, beta, p, a, s, H_out):
        src, dst = _rows(g.plan)
        Zh, sh = _np(Z), s.numpy()
        sh = np.where(sh == 0, f32(1), sh)
        pk = p.numpy().astype(np.int64)
        w = (a.numpy() / sh[dst, pk]).astype(f32)
        acc = np.zeros((g.n_rows,) + Zh.shape[1:], dtype=f32)
        np.add.at(acc, (src - g.row_offset, pk), w[:, None] * Zh[dst, pk])
        lo, hi = g.row_offset, g.row_offset + g.n_rows
        H_out[lo:hi] = torch.from_numpy((f32(beta) * Zh[lo:hi] + f32(1 - beta) * acc).astype(f32)).to(H_out.dtype)

    def score_pairs_fwd(self, Z, H, pairs, t):
        return torch.from_numpy(sparse_ref.score_pairs(_np(Z), _np(H), pairs.pu.numpy(), pairs.pv.numpy(), t))

    # ---- the training step of the scorer in one pass over the local incidence rows (dl_score_pairs_train)
    def score_pairs_train_supported(self, inc, K, d, table_dtype) -> bool:
        return True

    def score_pairs_train(self, Z, H, inc, t, label, weight):
        u, v = _rows(inc.inc)
        q = inc.inc_pair.numpy().astype(np.int64)
        Zh, Hh = _np(Z), _np(H)
        pr, qk, ex = sparse_ref.score_pairs(Zh, Hh, u, v, t, return_parts=True)
        y, w = label.numpy()[q], weight.numpy()[q]
        den = np.maximum(pr * (f32(1) - pr), f32(1e-12))
        gl = (w * (pr - y) / den * (pr * (f32(1) - pr))).astype(f32)      # dl_pair_bce's gradient times the sigmoid backward
        prob = np.full(label.numel(), np.nan, dtype=f32)                   # pairs without a local endpoint stay unwritten
        prob[q] = pr
        n, lo = inc.inc.n_rows, inc.inc.row_offset
        dZ = np.full(Zh.shape, np.nan, dtype=f32)
        dH = np.full(Zh.shape, np.nan, dtype=f32)
        accZ = np.zeros((n,) + Zh.shape[1:], dtype=f32)
        accH = np.zeros_like(accZ)
        np.add.at(accH, u - lo, (gl[:, None] * ex)[:, :, None] * Hh[v])
        np.add.at(accZ, u - lo, (gl[:, None] * qk * ex / f32(t))[:, :, None] * Zh[v])
        dZ[lo:lo + n], dH[lo:lo + n] = accZ, accH
        return torch.from_numpy(prob), torch.from_numpy(dZ), torch.from_numpy(dH)

    # ---- the same training step from the forward scorer + the backward over a pair list of its own (Shard.touching)
    def score_pairs_fwd_terms(self, Z, H, pairs, t):
        return self.score_pairs_fwd(Z, H, pairs, t), None

    def score_pairs_bwd_terms(self, Z, H, pairs, t, prob, g_prob, coef, dZ_out, dH_out):
        self.score_pairs_bwd(Z, H, pairs, t, prob, g_prob, dZ_out, dH_out)

    def pair_bce_grad(self, prob, label, weight):
        pr, y, w = prob.numpy().astype(f32), label.numpy(), weight.numpy()
        g = (w * (pr - y) / np.maximum(pr * (f32(1) - pr), f32(1e-12))).astype(f32)
        return self.pair_bce_sum(prob, label, weight), torch.from_numpy(g)

    def pair_bce_sum(self, prob, label, weight):
        from oracle import metrics_ref  # noqa: F401
        pr, y, w = prob.numpy().astype(f32), label.numpy(), weight.numpy()
        with np.errstate(divide="ignore"):
            lp = np.maximum(np.log(pr), f32(-100))
            l1p = np.maximum(np.log(f32(1) - pr), f32(-100))
        return torch.tensor(float((w * -(y * lp + (f32(1) - y) * l1p)).sum(dtype=np.float64)), dtype=torch.float32)

    def score_pairs_bwd(self, Z, H, inc, t, prob, g_prob, dZ_out, dH_out):
        u, v = _rows(inc.inc)
        q = inc.inc_pair.numpy().astype(np.int64)
        Zh, Hh = _np(Z), _np(H)
        pr, gq = prob.numpy()[q], g_prob.numpy()[q]
        _prob, qk, ex = sparse_ref.score_pairs(Zh, Hh, u, v, t, return_parts=True)
        gl = (gq * pr * (f32(1) - pr)).astype(f32)
        n = inc.inc.n_rows
        dH = np.zeros((n,) + Zh.shape[1:], dtype=f32)
        dZ = np.zeros_like(dH)
        np.add.at(dH, u - inc.inc.row_offset, (gl[:, None] * ex)[:, :, None] * Hh[v])
        np.add.at(dZ, u - inc.inc.row_offset, (gl[:, None] * qk * ex / f32(t))[:, :, None] * Zh[v])
        lo = inc.inc.row_offset
        dZ_out[lo:lo + n] = torch.from_numpy(dZ)
        dH_out[lo:lo + n] = torch.from_numpy(dH)

    def bwd_phase1(self, g, Z, beta, p, a, s, dH, ds_out):
        src, dst = _rows(g.plan)
        Zh, Dh, sh = _np(Z), dH.numpy(), s.numpy()
        pk = p.numpy().astype(np.int64)
        dw = f32(1 - beta) * np.einsum("ed,ed->e", Dh[src, pk], Zh[dst, pk], dtype=f32)
        dwr = f32(1 - beta) * np.einsum("ed,ed->e", Dh[dst, pk], Zh[src, pk], dtype=f32)
        acc = np.zeros((g.n_rows, Zh.shape[1]), dtype=f32)
        np.add.at(acc, (src - g.row_offset, pk), dwr * a.numpy())
        lo, hi = g.row_offset, g.row_offset + g.n_rows
        s_loc = sh[lo:hi]
        with np.errstate(divide="ignore", invalid="ignore"):
            ds = np.where(s_loc == 0, f32(0), -acc / (s_loc * s_loc)).astype(f32)
        ds_out[lo:hi] = torch.from_numpy(ds)
        return torch.from_numpy(dw.astype(f32)), torch.from_numpy(dwr.astype(f32))

    def bwd_phase2(self, g, Z, beta, t, p, a, s, dH, dw, dwr, ds, dZ_out, accumulate):
        src, dst = _rows(g.plan)
        Zh, Dh, sh, dsh = _np(Z), dH.numpy(), s.numpy(), ds.numpy()
        sh = np.where(sh == 0, f32(1), sh)
        pk = p.numpy().astype(np.int64)
        ah = a.numpy()
        E, K = src.size, Zh.shape[1]
        lo, hi = g.row_offset, g.row_offset + g.n_rows
        acc = (f32(beta) * Dh[lo:hi]).astype(f32)
        np.add.at(acc, (src - lo, pk), (f32(1 - beta) * ah / sh[src, pk])[:, None] * Dh[dst, pk])
        with np.errstate(over="ignore", invalid="ignore"):
            ex = np.exp(_edge_dots(Zh, src, dst) / f32(t), dtype=f32)
            alpha = ex / ex.sum(axis=1, dtype=f32, keepdims=True)
        onehot = np.zeros((E, K), dtype=f32)
        onehot[np.arange(E), pk] = 1
        da = dw.numpy() / sh[dst, pk] + dsh[src, pk]
        dar = dwr.numpy() / sh[src, pk] + dsh[dst, pk]
        c = ((da + dar) * ah)[:, None] * (onehot - alpha) / f32(t)
        np.add.at(acc, src - lo, c[:, :, None] * Zh[dst])
        out = torch.from_numpy(acc.astype(f32))
        if accumulate:
            dZ_out[lo:hi] += out
        else:
            dZ_out[lo:hi] = out
