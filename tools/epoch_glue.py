#!/usr/bin/env python3
"""Which Python lines launch the small torch kernels (copies, fills, elementwise ops) of a training epoch?
usage (GPU box): python tools/epoch_glue.py [dataset] [graph]  -> per (op, source line) device time per epoch."""
import os, sys, collections
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
EPOCHS = 10
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    run_link_prediction(model, x, run, epochs=EPOCHS, lr=1e-4)
    torch.cuda.synchronize()
acc = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if not e.name.startswith("aten::") or e.device_time_total <= 0 or e.cpu_children and any(c.device_time_total > 0 and c.name.startswith("aten::") for c in e.cpu_children):
        continue
    where = next((f for f in e.stack if "/repo/" in f or "optim" in f), e.stack[0] if e.stack else "?")
    k = (e.name, where.split("/")[-1][:70])
    acc[k][0] += 1
    acc[k][1] += e.device_time_total
for (name, where), (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"{name:28s} {where:72s} {n / EPOCHS:6.1f} calls/epoch {t / EPOCHS:8.1f} us/epoch")
