#!/usr/bin/env python3
"""The reference's own call, model(x, adj_sym) -> (emb, link_pred [N,N]) (main_disentangled.py:194), on the
drop-in module at squirrel size: wall time of forward and of forward + masked-BCE backward."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from disenlink_amd.data import synthetic_graph
from disenlink_amd.model import Disentangle
from disenlink_amd.splits import make_link_split
dev = torch.device("cuda:0")
sg = synthetic_graph(sys.argv[1] if len(sys.argv) > 1 else "squirrel", seed=0)
N = sg.n_nodes
split = make_link_split(sg.src, sg.dst, N, m=5, seed=0)
adj = torch.zeros(N, N, device=dev)
adj[torch.from_numpy(split.train_src).to(dev), torch.from_numpy(split.train_dst).to(dev)] = 1
adj = ((adj + adj.t()) != 0).float()
pos = torch.zeros(N, N, device=dev); pos[torch.from_numpy(split.pos_train.u).to(dev), torch.from_numpy(split.pos_train.v).to(dev)] = 1
neg = torch.zeros(N, N, device=dev); neg[torch.from_numpy(split.neg_train.u).to(dev), torch.from_numpy(split.neg_train.v).to(dev)] = 1
x = torch.from_numpy(sg.features()).to(dev)
torch.manual_seed(0)
model = Disentangle(sg.n_feat, 512, 64, nfactor=8, beta=0.5, t=1).to(dev)
F = torch.nn.functional
def step(backward):
    emb, a_pred = model(x, adj)
    if backward:
        loss = F.binary_cross_entropy(a_pred[pos == 1], adj[pos == 1]) + F.binary_cross_entropy(a_pred[neg == 1], adj[neg == 1]) / 5
        model.zero_grad(); loss.backward()
for bw in (False, True):
    for _ in range(2): step(bw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): step(bw)
    torch.cuda.synchronize()
    print(f"N={N}: dense drop-in {'forward+loss+backward' if bw else 'forward'}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")
