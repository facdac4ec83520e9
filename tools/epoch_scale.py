"""One-off: a few training epochs at a large synthetic shape (does it run, how long, how much memory).
usage: epoch_scale.py <workload> <K> <d> <nhid> <f32|bf16> [epochs] [eager]   (eager: skip the graph-replay arm)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disenlink_amd.data import synthetic_graph
from disenlink_amd.model import Disentangle
from disenlink_amd.splits import make_link_split
from disenlink_amd.train import prepare_run, run_link_prediction
name, K, d, nhid, dt = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
epochs = int(sys.argv[6]) if len(sys.argv) > 6 else 10
dev = torch.device("cuda:0")
t0 = time.perf_counter()
sg = synthetic_graph(name, seed=0)
split = make_link_split(sg.src, sg.dst, sg.n_nodes, m=5, seed=0, device=dev)     # sorts / searches on the GPU: same split
run = prepare_run(split, dev, row_bytes=K * d * (2 if dt == "bf16" else 4))
x = torch.from_numpy(sg.features()).to(dev)
print(f"{name}: N={sg.n_nodes} F={sg.n_feat} train pairs {run.n_pos + run.n_neg} prep {time.perf_counter() - t0:.1f} s", flush=True)
torch.manual_seed(0)
model = Disentangle(sg.n_feat, nhid, d, nfactor=K, beta=0.5, t=1, table_dtype=torch.bfloat16 if dt == "bf16" else torch.float32).to(dev)
for use_graph in ((False,) if len(sys.argv) > 7 and sys.argv[7] == "eager" else (False, True)):
    run_link_prediction(model, x, run, epochs=2, lr=1e-4, use_graph=use_graph)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = run_link_prediction(model, x, run, epochs=epochs, lr=1e-4, use_graph=use_graph)
    torch.cuda.synchronize()
    print(f"  K={K} d={d} nhid={nhid} {dt} graph={use_graph}: {(time.perf_counter() - t0) / epochs * 1e3:.2f} ms/epoch, loss {res.losses[0]:.4f} -> {res.losses[-1]:.4f}, "
          f"val auc {res.val_aucs[-1]:.4f}, peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
