"""One-pass training scorer, wave-per-entry kernel against the group-per-entry kernel (and the separate kernels) on a
bench workload: python tools/r4_train_debug.py <scale> <workload>.  prob differs by <= 1 ulp between the two kernels
(another dot-product tree); where that ulp decides whether a sigmoid saturates to exactly 1.0 the gradient of that pair is
w or 0 (SURVEY.md finding 4: the reference behaves the same way) — rows holding such a pair differ visibly, all others to 1e-7."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from disenlink_amd import ops
from disenlink_amd.metrics import pair_bce_weights
dev = torch.device("cuda:0")
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
wl = sys.argv[2] if len(sys.argv) > 2 else "snap_patents"
K, d, t, beta = 8, 64, 1.0, 0.5
sg, split, graph, pairs, model, x, Z = bench.build_workload(wl, dev, K, d, 512, scale=scale)
del model, x
H = ops.aggregate_fwd(graph, Z, beta, *ops.route_fwd(graph, Z, t))
P = pairs.n_pairs
rng = np.random.default_rng(3)
y = torch.from_numpy((rng.random(P) < 0.17).astype(np.float32)).to(dev)
w = pair_bce_weights(P // 6, P - P // 6, 5, dev)
w[-1000:] = 0
prob1, dZ1, dH1 = ops.score_pairs_train(Z, H, pairs, t, y, w)
os.environ["DL_TRAIN_GROUP_KERNEL"] = "1"
prob0, dZ0, dH0 = ops.score_pairs_train(Z, H, pairs, t, y, w)
torch.cuda.synchronize()
print("scale", scale, "N", graph.n_nodes, "P", P, "inc seg_len", pairs.inc.seg_len, "n_slices", pairs.inc.n_slices, "n_slots", pairs.inc.n_slots)
for name, a, b in (("prob", prob1, prob0), ("dZ", dZ1, dZ0), ("dH", dH1, dH0)):
    diff = (a - b).abs()
    print(name, "max|diff|", float(diff.max()), "max|ref|", float(b.abs().max()))
rowdiff = (dH1 - dH0).abs().amax(dim=(1, 2))
bad = torch.nonzero(rowdiff > 1e-3 * dH0.abs().max()).reshape(-1)
print("bad rows", bad.numel(), bad[:10].tolist(), bad[-5:].tolist())
if bad.numel():
    deg = (pairs.inc.rowptr[1:] - pairs.inc.rowptr[:-1]).long()
    print("their incidence degrees", deg[bad[:10]].tolist(), "zero rows in new:", int((dH1[bad].abs().amax(dim=(1, 2)) == 0).sum()))
    print("first bad row", int(bad[0]), "as bytes offset", int(bad[0]) * K * d * 4)
# which rows differ most in dZ, relative to the row's own magnitude
deg = (pairs.inc.rowptr[1:] - pairs.inc.rowptr[:-1]).long()
rd = (dZ1 - dZ0).abs().amax(dim=(1, 2))
top = torch.topk(rd, 10).indices
print("top dZ-diff rows", top.tolist())
print(" degrees", deg[top].tolist())
print(" diffs", [f"{v:.2e}" for v in rd[top].tolist()], "row max", [f"{v:.2e}" for v in dZ0[top].abs().amax(dim=(1, 2)).tolist()])
# the separate-kernel form as a third opinion
prob_c, coef = ops.score_pairs_fwd(Z, H, pairs.pu, pairs.pv, t, pairs, want_coef=True)
pr = prob0.detach().clone().requires_grad_(True)
(g_prob,) = torch.autograd.grad(ops.PairBCE.apply(pr, y, w), pr)
dZs, dHs = ops.score_pairs_bwd(Z, H, pairs, t, prob0, g_prob, coef=coef)
print("vs separate kernels: wave", float((dZ1 - dZs).abs().max()), "group", float((dZ0 - dZs).abs().max()), "| dH wave", float((dH1 - dHs).abs().max()), "group", float((dH0 - dHs).abs().max()))
r = int(top[0])
print("row", r, "deg", int(deg[r]), "dZ1", dZ1[r, 0, :4].tolist(), "dZ0", dZ0[r, 0, :4].tolist(), "dZs", dZs[r, 0, :4].tolist())
# per-entry reconstruction of the worst row in fp64
inc = pairs.inc
b, e = int(inc.rowptr[r]), int(inc.rowptr[r + 1])
cols = inc.col[b:e].long(); qs = pairs.inc_pair[b:e].long()
zu, hu = Z[r].double(), H[r].double()
zv, hv = Z[cols].double(), H[cols].double()
ps = (zu[None] * zv).sum(-1); pq = (hu[None] * hv).sum(-1)          # [deg, K]
E = torch.exp(ps / t); logit = (pq * E).sum(-1); p = torch.sigmoid(logit)
gl = w[qs].double() * (p - y[qs].double())
contrib = (gl[:, None] * pq * E / t)[:, :, None] * zv                # [deg, K, d] -> dZ row
want = contrib.sum(0)
print("fp64 row vs wave", float((want - dZ1[r].double()).abs().max()), "vs group", float((want - dZ0[r].double()).abs().max()))
miss = want - dZ1[r].double()
# which entry explains the miss?
err = [(float((miss - contrib[k]).abs().max()), k) for k in range(e - b)]
err2 = [(float((miss + contrib[k]).abs().max()), k) for k in range(e - b)]
print("best single-entry explanation (missing)", min(err), "(extra)", min(err2), "of", float(miss.abs().max()))
k = min(err)[1]
print("entry", k, "of", e - b, "col", int(cols[k]), "pair", int(qs[k]), "w", float(w[qs[k]]), "y", float(y[qs[k]]), "p", float(p[k]), "P", P)
print("cols", cols.tolist())
print("seg plan: rowseg", [(int(sb), int(se)) for sr, sb, se in zip(inc.seg_row.tolist(), inc.seg_beg.tolist(), inc.seg_end.tolist()) if sr == r])
