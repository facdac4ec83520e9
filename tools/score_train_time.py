"""Training step of the scorer on the bench workload: one pass (dl_score_pairs_train) vs forward storing terms +
fused loss + two backward passes.  usage: python tools/score_train_time.py [workload] [K] [d] [f32|bf16]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from disenlink_amd import ops
from disenlink_amd.metrics import pair_bce_weights
dev = torch.device("cuda:0")
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d = int(sys.argv[3]) if len(sys.argv) > 3 else 64
bf16 = len(sys.argv) > 4 and sys.argv[4] == "bf16"
sg, split, graph, pairs, model, x, Z = bench.build_workload(sys.argv[1] if len(sys.argv) > 1 else "squirrel", dev, K, d, 512,
                                                            elem_bytes=2 if bf16 else 4)
if bf16:
    Z = Z.to(torch.bfloat16)
t, beta = 1.0, 0.5
H = ops.aggregate_fwd(graph, Z, beta, *ops.route_fwd(graph, Z, t))
P = pairs.n_pairs
y = (torch.rand(P, device=dev) < 0.2).float()
w = pair_bce_weights(P // 6, P - P // 6, 5, dev)
def one_pass():
    return ops.score_pairs_train(Z, H, pairs, t, y, w)
def separate():
    prob, coef = ops.score_pairs_fwd(Z, H, pairs.pu, pairs.pv, t, pairs, want_coef=True)
    pr = prob.detach().requires_grad_(True)
    (g,) = torch.autograd.grad(ops.PairBCE.apply(pr, y, w), pr)
    return (prob,) + tuple(ops.score_pairs_bwd(Z, H, pairs, t, prob, g, coef=coef))
for name, fn in (("one pass", one_pass), ("separate", separate), ("one pass", one_pass), ("separate", separate)):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); e1.synchronize()
    print(f"{sys.argv[1:]} {name}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us", flush=True)
a, b = one_pass(), separate()
print("max|prob diff|", float((a[0] - b[0]).abs().max()), "max|dZ diff|/max", float((a[1] - b[1]).abs().max() / b[1].abs().max()),
      "max|dH diff|/max", float((a[2] - b[2]).abs().max() / b[2].abs().max()))
