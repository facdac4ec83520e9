#!/usr/bin/env python3
"""Wall time of one training epoch of disenlink_amd.train.run_link_prediction on the bench workload."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from disenlink_amd.data import synthetic_graph
from disenlink_amd.model import Disentangle
from disenlink_amd.splits import make_link_split
from disenlink_amd.train import prepare_run, run_link_prediction

# usage: epoch_time.py [dataset] [n_feat] [epochs]   (n_feat overrides the synthetic feature width, e.g. 2089)
name = sys.argv[1] if len(sys.argv) > 1 else "squirrel"
EPOCHS = int(sys.argv[3]) if len(sys.argv) > 3 else 200
dev = torch.device("cuda:0")
sg = synthetic_graph(name, seed=0)
if len(sys.argv) > 2 and int(sys.argv[2]) > 0:
    sg.n_feat = int(sys.argv[2])
split = make_link_split(sg.src, sg.dst, sg.n_nodes, m=5, seed=0)
run = prepare_run(split, dev)
x = torch.from_numpy(sg.features()).to(dev)
for projection, use_graph in (("library", False), ("mfma", False), ("library", True), ("mfma", True)):
    torch.manual_seed(0)
    model = Disentangle(sg.n_feat, 512, 64, nfactor=8, beta=0.5, t=1, projection=projection).to(dev)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    run_link_prediction(model, x, run, epochs=3, lr=1e-4, use_graph=use_graph)
    model.load_state_dict(sd)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = run_link_prediction(model, x, run, epochs=EPOCHS, lr=1e-4, use_graph=use_graph)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / EPOCHS      # incl. graph capture when use_graph
    print(f"{name} F={sg.n_feat} projection={projection} use_graph={use_graph}: {dt * 1e3:.2f} ms per epoch (train pairs {run.n_pos + run.n_neg}, val "
          f"{run.label_val.numel()}); loss {res.losses[0]:.4f} -> {res.losses[-1]:.4f}, val auc {res.val_aucs[-1]:.4f}, "
          f"test auc {res.test_auc:.4f}")
