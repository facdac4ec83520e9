"""Loss trajectories of the eager and the HIP-graph epoch for both projection paths (they should agree)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disenlink_amd.data import synthetic_graph
from disenlink_amd.model import Disentangle
from disenlink_amd.splits import make_link_split
from disenlink_amd.train import prepare_run, run_link_prediction
dev = torch.device("cuda:0")
sg = synthetic_graph("chameleon", seed=0)
split = make_link_split(sg.src, sg.dst, sg.n_nodes, m=5, seed=0)
run = prepare_run(split, dev)
x = torch.from_numpy(sg.features()).to(dev)
for projection in ("library", "mfma"):
    for use_graph in (False, True):
        torch.manual_seed(0)
        model = Disentangle(sg.n_feat, 512, 64, nfactor=8, beta=0.5, t=1, projection=projection).to(dev)
        res = run_link_prediction(model, x, run, epochs=12, lr=1e-4, use_graph=use_graph)
        print(projection, "graph" if use_graph else "eager", " ".join(f"{v:.5f}" for v in res.losses), flush=True)
