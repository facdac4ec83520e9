#!/usr/bin/env python3
"""Which Python lines launch the small torch kernels (copies, fills, index kernels) of a training epoch?
usage (GPU box): python tools/epoch_glue.py [dataset]  -> per-op table with source locations (torch.profiler, eager loop)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from disenlink_amd.data import synthetic_graph
from disenlink_amd.model import Disentangle
from disenlink_amd.splits import make_link_split
from disenlink_amd.train import prepare_run, run_link_prediction
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
sg = synthetic_graph(sys.argv[1] if len(sys.argv) > 1 else "squirrel", seed=0)
split = make_link_split(sg.src, sg.dst, sg.n_nodes, m=5, seed=0)
run = prepare_run(split, dev)
x = torch.from_numpy(sg.features()).to(dev)
torch.manual_seed(0)
model = Disentangle(sg.n_feat, 512, 64, nfactor=8, beta=0.5, t=1).to(dev)
run_link_prediction(model, x, run, epochs=3, lr=1e-4)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    run_link_prediction(model, x, run, epochs=10, lr=1e-4)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=6).table(sort_by="self_cuda_time_total", row_limit=40, max_src_column_width=110,
                                                 max_name_column_width=50))
