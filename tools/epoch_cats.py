#!/usr/bin/env python3
"""Who calls torch.cat / torch.stack / Tensor.copy_ / clone / contiguous in a training epoch? (GPU box)"""
import os, sys, collections, traceback
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from disenlink_amd.data import synthetic_graph
from disenlink_amd.model import Disentangle
from disenlink_amd.splits import make_link_split
from disenlink_amd.train import prepare_run, run_link_prediction
dev = torch.device("cuda:0")
sg = synthetic_graph(sys.argv[1] if len(sys.argv) > 1 else "squirrel", seed=0)
split = make_link_split(sg.src, sg.dst, sg.n_nodes, m=5, seed=0)
run = prepare_run(split, dev)
x = torch.from_numpy(sg.features()).to(dev)
torch.manual_seed(0)
model = Disentangle(sg.n_feat, 512, 64, nfactor=8, beta=0.5, t=1).to(dev)
run_link_prediction(model, x, run, epochs=3, lr=1e-4)
from torch.utils._python_dispatch import TorchDispatchMode
acc = collections.Counter()
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(k in name for k in ("cat", "stack", "copy_", "clone", "fill_", "zero_", "mul", "div")):
            st = [f for f in traceback.extract_stack() if "/repo/" in f.filename and "epoch_cats" not in f.filename]
            where = f"{os.path.basename(st[-1].filename)}:{st[-1].lineno}" if st else "autograd/other"
            shape = tuple(args[0][0].shape) if isinstance(args[0], (list, tuple)) and len(args[0]) else (tuple(args[0].shape) if torch.is_tensor(args[0]) else ())
            acc[(name, where, shape)] += 1
        return func(*args, **(kwargs or {}))
E = 5
with Spy():
    run_link_prediction(model, x, run, epochs=E, lr=1e-4)
for (name, where, shape), n in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"{n / E:6.1f}/epoch  {name:32s} {where:28s} {shape}")
