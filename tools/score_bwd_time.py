#!/usr/bin/env python3
"""Scorer backward in its recompute form (dl_score_pairs_bwd without stored terms: what the dense link_pred backward of the
drop-in module runs) on the squirrel-shaped pair list.  usage (GPU box): python tools/score_bwd_time.py [dataset]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from disenlink_amd import ops
from disenlink_amd.data import synthetic_graph
from disenlink_amd.graph import Graph, PairList
from disenlink_amd.splits import make_link_split
dev = torch.device("cuda:0")
sg = synthetic_graph(sys.argv[1] if len(sys.argv) > 1 else "squirrel", seed=0)
split = make_link_split(sg.src, sg.dst, sg.n_nodes, m=5, seed=0)
pu = np.concatenate([split.pos_train.u, split.neg_train.u]); pv = np.concatenate([split.pos_train.v, split.neg_train.v])
K, d = 8, 64
pairs = PairList.build(torch.from_numpy(pu).to(dev), torch.from_numpy(pv).to(dev), sg.n_nodes, row_bytes=K * d * 4)
torch.manual_seed(0)
Z = torch.randn(sg.n_nodes, K, d, device=dev) * 0.24
H = torch.randn(sg.n_nodes, K, d, device=dev) * 0.24
prob = ops.score_pairs_fwd(Z, H, pairs.pu, pairs.pv, 1.0, pairs)
gp = torch.full((pairs.n_pairs,), 1.0 / pairs.n_pairs, device=dev)
for _ in range(5):
    ops.score_pairs_bwd(Z, H, pairs, 1.0, prob, gp)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(30):
    ops.score_pairs_bwd(Z, H, pairs, 1.0, prob, gp)
e1.record(); e1.synchronize()
print(f"{sys.argv[1] if len(sys.argv) > 1 else 'squirrel'}: recompute backward {e0.elapsed_time(e1) / 30 * 1e3:.1f} us for {pairs.n_pairs} pairs")
