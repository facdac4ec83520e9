"""aten-level op counts of one training epoch (torch.profiler), to find avoidable copies / fills."""
import os, sys
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disenlink_amd.data import synthetic_graph
from disenlink_amd.model import Disentangle
from disenlink_amd.splits import make_link_split
from disenlink_amd.train import prepare_run, run_link_prediction
dev = torch.device("cuda:0")
sg = synthetic_graph("squirrel", seed=0)
split = make_link_split(sg.src, sg.dst, sg.n_nodes, m=5, seed=0)
run = prepare_run(split, dev)
x = torch.from_numpy(sg.features()).to(dev)
torch.manual_seed(0)
model = Disentangle(sg.n_feat, 512, 64, nfactor=8, beta=0.5, t=1).to(dev)
run_link_prediction(model, x, run, epochs=3, lr=1e-4)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    run_link_prediction(model, x, run, epochs=10, lr=1e-4)
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.count)
for e in rows[:45]:
    print(f"{e.count / 10:7.1f}/epoch  cuda {getattr(e, 'device_time_total', getattr(e, 'cuda_time_total', 0)) / 10:8.1f} us  cpu {e.cpu_time_total / 10:8.1f} us  {e.key[:70]}")
