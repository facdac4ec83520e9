"""One-pass training scorer with the labels / weights laid out per incidence entry (PairList.bind_labels) against the
per-entry gathers through inc_pair (DL_ENTRY_LABELS=0): same bits, time of both.
usage: python tools/entry_labels_ab.py <workload> <K> <d> <f32|bf16> [inc_slices]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from disenlink_amd import ops
from disenlink_amd.graph import PairList
from disenlink_amd.metrics import pair_bce_weights
dev = torch.device("cuda:0")
name, K, d = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
bf16 = sys.argv[4] == "bf16"
sg, split, graph, pairs, model, x, Z = bench.build_workload(name, dev, K, d, 512, elem_bytes=2 if bf16 else 4)
if len(sys.argv) > 5:
    pairs = PairList.build(pairs.pu, pairs.pv, sg.n_nodes, row_bytes=K * d * (2 if bf16 else 4), inc_slices=int(sys.argv[5]), build_by_u=False)
if bf16:
    Z = Z.to(torch.bfloat16)
t, beta = 1.0, 0.5
H = ops.aggregate_fwd(graph, Z, beta, *ops.route_fwd(graph, Z, t))
P = pairs.n_pairs
y = torch.from_numpy(bench.build_workload.__globals__["np"].asarray(getattr(pairs, "bench_label", None))).to(dev) if hasattr(pairs, "bench_label") else (torch.rand(P, device=dev) < 0.17).float()
w = pair_bce_weights(int(y.sum().item()), P - int(y.sum().item()), 5, dev)
w[::7] = 0.0                                                        # some weight-0 pairs (validation pairs ride along like this)
fn = lambda: ops.score_pairs_train(Z, H, pairs, t, y, w)

def timed():
    for _ in range(5): fn()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    return best
os.environ["DL_ENTRY_LABELS"] = "0"
ref = [v.clone() for v in fn()]
t0 = timed()
os.environ["DL_ENTRY_LABELS"] = "1"
fn(); out = fn()
bound = pairs._yw is not None
same = all(torch.equal(a, b) for a, b in zip(out, ref))
t1 = timed()
os.environ["DL_ENTRY_LABELS"] = "0"
t2 = timed()
print(f"{name} K={K} d={d} {'bf16' if bf16 else 'f32'} inc slices {pairs.inc.n_slices}: gathers {t0:.1f} / {t2:.1f} us, per-entry labels {t1:.1f} us "
      f"(bound: {bound}); same bits: {same}", flush=True)
sys.exit(0 if same and bound else 1)
