#!/usr/bin/env python3
"""Per-epoch wall time of train.run_link_prediction with the end-of-epoch bookkeeping on the device (default) and on the
host (DL_DEVICE_EARLY_STOP=0), eager and replayed from a HIP graph.  usage: epoch_ab_early_stop.py [dataset] [epochs]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from disenlink_amd.data import synthetic_graph
from disenlink_amd.model import Disentangle
from disenlink_amd.splits import make_link_split
from disenlink_amd.train import prepare_run, run_link_prediction
name = sys.argv[1] if len(sys.argv) > 1 else "squirrel"
EPOCHS = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0")
sg = synthetic_graph(name, seed=0)
split = make_link_split(sg.src, sg.dst, sg.n_nodes, m=5, seed=0)
run = prepare_run(split, dev)
x = torch.from_numpy(sg.features()).to(dev)
# DL_AB_ONLY=graph: only the replayed, device-side rows (sweeps of DL_GRAPH_EPOCHS / DL_GRAPH_EXECS, one process each)
only_graph = os.environ.get("DL_AB_ONLY") == "graph"
for rep in range(2):
    for mode in (("1",) if only_graph else ("0", "1")):
        for use_graph in ((True,) if only_graph else (False, True)):
            os.environ["DL_DEVICE_EARLY_STOP"] = mode
            torch.manual_seed(0)
            model = Disentangle(sg.n_feat, 512, 64, nfactor=8, beta=0.5, t=1).to(dev)
            run_link_prediction(model, x, run, epochs=3, lr=1e-4, use_graph=use_graph)
            torch.manual_seed(0)
            model = Disentangle(sg.n_feat, 512, 64, nfactor=8, beta=0.5, t=1).to(dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = run_link_prediction(model, x, run, epochs=EPOCHS, lr=1e-4, use_graph=use_graph)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / max(res.epochs_run, 1)
            print(f"{name} device_early_stop={mode} use_graph={use_graph} graph_epochs={os.environ.get('DL_GRAPH_EPOCHS', '1')} execs={os.environ.get('DL_GRAPH_EXECS', '1')}: {dt * 1e3:.3f} ms per epoch; loss {res.losses[-1]:.6f} "
                  f"val {res.val_aucs[-1]:.6f} test {res.test_auc:.6f} epochs {res.epochs_run}", flush=True)
