"""debug: wide training scorer (16,128,bf16) against the forward scorer on a small problem"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from disenlink_amd import ops
from disenlink_amd.graph import Graph, PairList
dev = "cuda:0"
K, d = 16, 128
rng = np.random.default_rng(0)
N, E, P = 200, 1000, 600
src, dst = rng.integers(0, N, E), rng.integers(0, N, E)
pu, pv = rng.integers(0, N, P), rng.integers(0, N, P)
G = Graph.from_edge_rows(torch.from_numpy(src), torch.from_numpy(dst), N).to(dev)
for dtype in (torch.bfloat16, torch.float32):
    pairs = PairList.build(torch.from_numpy(pu).to(dev), torch.from_numpy(pv).to(dev), N, row_bytes=K * d * (2 if dtype == torch.bfloat16 else 4))
    Z = (torch.randn(N, K, d, generator=torch.Generator().manual_seed(1)) * 0.1).to(dev).to(dtype)
    for t in (1.0, 2.0):
        H = ops.aggregate_fwd(G, Z, 0.6, *ops.route_fwd(G, Z, t))
        y = (torch.rand(P, device=dev) < 0.3).float()
        w = torch.full((P,), 1.0 / P, device=dev)
        prob1, dZ1, dH1 = ops.score_pairs_train(Z, H, pairs, t, y, w)
        prob0 = ops.score_pairs_fwd(Z, H, pairs.pu, pairs.pv, t, pairs)
        pr = prob1.detach().clone().requires_grad_(True)
        (g,) = torch.autograd.grad(ops.PairBCE.apply(pr, y, w), pr)
        dZ0, dH0 = ops.score_pairs_bwd(Z, H, pairs, t, prob0, g)
        print(dtype, t, "prob max diff", float((prob1 - prob0).abs().max()), "dZ rel", float((dZ1 - dZ0).abs().max() / dZ0.abs().max()),
              "dH rel", float((dH1 - dH0).abs().max() / dH0.abs().max()), flush=True)
        if float((prob1 - prob0).abs().max()) > 1e-3:
            lg1 = torch.log(prob1 / (1 - prob1)); lg0 = torch.log(prob0 / (1 - prob0))
            print("  logit ratio (first 8):", (lg1 / lg0)[:8].tolist())
