"""Does the order of the caller's pair list matter?  Scorer forward (with and without stored terms) and backward on
the bench workload: pairs as built ([pos | neg | val]) vs the same pairs sorted by first endpoint.
usage: python tools/score_pair_order.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disenlink_amd import ops
from disenlink_amd.data import synthetic_graph
from disenlink_amd.graph import PairList
from disenlink_amd.splits import make_link_split
dev = torch.device("cuda:0")
sg = synthetic_graph("squirrel", seed=0)
split = make_link_split(sg.src, sg.dst, sg.n_nodes, m=5, seed=0)
pu = np.concatenate([split.pos_train.u, split.neg_train.u, split.val.u])
pv = np.concatenate([split.pos_train.v, split.neg_train.v, split.val.v])
N, K, d, t = sg.n_nodes, 8, 64, 1.0
Z = torch.randn(N, K, d, device=dev) * 0.2
H = torch.randn(N, K, d, device=dev) * 0.2
def ev(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for name, order in (("as built", np.arange(pu.size)), ("sorted by u", np.argsort(pu, kind="stable")),
                    ("sorted by (u, v)", np.lexsort((pv, pu)))):
    u, v = torch.as_tensor(pu[order], device=dev), torch.as_tensor(pv[order], device=dev)
    pairs = PairList.build(u, v, N, row_bytes=K * d * 4)
    prob, coef = ops.score_pairs_fwd(Z, H, pairs.pu, pairs.pv, t, pairs, want_coef=True)
    g = torch.randn_like(prob)
    t_f = ev(lambda: ops.score_pairs_fwd(Z, H, pairs.pu, pairs.pv, t, pairs))
    t_fc = ev(lambda: ops.score_pairs_fwd(Z, H, pairs.pu, pairs.pv, t, pairs, want_coef=True))
    t_b = ev(lambda: ops.score_pairs_bwd(Z, H, pairs, t, prob, g, coef=coef))
    print(f"{name:18s} P={pu.size}: forward {t_f:6.1f} us   forward storing terms {t_fc:6.1f} us   backward {t_b:6.1f} us", flush=True)
