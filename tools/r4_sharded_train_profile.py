"""Where a sharded training step goes (one-rank RCCL group on one GPU): torch.profiler kernel table of
dist.sharded_forward_loss + backward + allreduce_gradients on a bench workload.
usage: python tools/r4_sharded_train_profile.py penn94 16 128 bf16 [scale]"""
import os, sys
import numpy as np, torch
import torch.distributed as tdist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disenlink_amd import dist as dd
from disenlink_amd import dist_bench
from disenlink_amd.model import Disentangle
wl = sys.argv[1] if len(sys.argv) > 1 else "penn94"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
d = int(sys.argv[3]) if len(sys.argv) > 3 else 128
dtype = sys.argv[4] if len(sys.argv) > 4 else "bf16"
scale = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
os.environ.setdefault("NCCL_DEBUG", "WARN")
tdist.init_process_group("nccl", init_method="tcp://127.0.0.1:29655", rank=0, world_size=1, device_id=dev)
tab = torch.bfloat16 if dtype == "bf16" else torch.float32
wb = 2 if dtype == "bf16" else 4
prob = dist_bench.build_problem(wl, scale, dev)
shard = dd.Shard.build(0, 1, prob.sg.n_nodes, prob.train_src, prob.train_dst, prob.pu, prob.pv, dev, row_bytes=K * d * wb, n_chunks=1)
torch.manual_seed(0)
model = Disentangle(prob.sg.n_feat, 512, d, nfactor=K, beta=0.5, t=1, table_dtype=tab).to(dev)
x = torch.from_numpy(prob.sg.features()).to(dev)
P = prob.pu.size
tpu, tpv = torch.as_tensor(prob.pu, device=dev), torch.as_tensor(prob.pv, device=dev)
label = ((tpu * 2654435761 + tpv) % 6 == 0).float()
weight = torch.full((P,), 1.0 / P, device=dev)

def step():
    model.zero_grad(set_to_none=True)
    _e, _p, loss = dd.sharded_forward_loss(model, x, shard, label, weight)
    loss.backward()
    dd.allreduce_gradients(model)

for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=22, max_name_column_width=70))
tdist.destroy_process_group()
