"""Device and host memory after each of 4 runs of 1,500 epochs (chameleon-shaped), eager and graph-replayed: nothing may grow
with the epochs or with the runs (the reference's recipes are 2,000 epochs x 10 runs).  usage: python tools/memory_growth.py"""
import os, sys, gc, resource, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disenlink_amd.data import synthetic_graph
from disenlink_amd.model import Disentangle
from disenlink_amd.splits import make_link_split
from disenlink_amd.train import prepare_run, run_link_prediction
dev = torch.device("cuda")
sg = synthetic_graph("chameleon", seed=0)
x = torch.from_numpy(sg.features()).to(dev)
def rss(): return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024
for use_graph in (False, True):
    for run in range(4):
        split = make_link_split(sg.src, sg.dst, sg.n_nodes, m=5, seed=run)
        torch.manual_seed(run)
        model = Disentangle(x.shape[1], 512, 32, nfactor=5, beta=0.7, t=1).to(dev)
        res = run_link_prediction(model, x, prepare_run(split, dev, row_bytes=5 * 32 * 4), epochs=1500, lr=1e-4, patience=2000, use_graph=use_graph)
        del model, split, res
        gc.collect(); torch.cuda.synchronize()
        print(f"graph={use_graph} run {run}: 1500 epochs; device allocated {torch.cuda.memory_allocated() / 2**20:8.1f} MiB, reserved {torch.cuda.memory_reserved() / 2**20:8.1f} MiB, "
              f"host max RSS {rss():8.1f} MiB", flush=True)
