"""One-pass training scorer: interleaved A/B of its runtime choices in ONE process — incidence slices, per-entry labels
(PairList.bind_labels), rows of several units summed inside the launch (DL_INKERNEL_COMBINE) — rounds of 10 launches per
configuration, the configurations taken in turn, median / min over the rounds (clock and thermal drift hit all of them).
usage: python tools/train_scorer_ab.py <workload> <K> <d> <f32|bf16> <slices,slices,...> [rounds]"""
import itertools, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from disenlink_amd import _lib, ops
from disenlink_amd.graph import PairList
from disenlink_amd.metrics import pair_bce_weights
dev = torch.device("cuda:0")
name, K, d = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
bf16 = sys.argv[4] == "bf16"
slices = [int(v) for v in sys.argv[5].split(",")]
rounds = int(sys.argv[6]) if len(sys.argv) > 6 else 12
sg, split, graph, pairs, model, x, Z = bench.build_workload(name, dev, K, d, 512, elem_bytes=2 if bf16 else 4)
if bf16:
    Z = Z.to(torch.bfloat16)
t, beta = 1.0, 0.5
H = ops.aggregate_fwd(graph, Z, beta, *ops.route_fwd(graph, Z, t))
P = pairs.n_pairs
y = torch.from_numpy(pairs.bench_label).to(dev)
w = pair_bce_weights(int(y.sum().item()), P - int(y.sum().item()), 5, dev)
plans = {s: PairList.build(pairs.pu, pairs.pv, sg.n_nodes, row_bytes=K * d * (2 if bf16 else 4), inc_slices=s, build_by_u=False) for s in slices}
configs = list(itertools.product(slices, (0, 1), (0, 1)))          # (slices, entry labels, in-launch row sums)

def setup(c):
    os.environ["DL_ENTRY_LABELS"] = str(c[1])
    os.environ["DL_INKERNEL_COMBINE"] = str(c[2])
    _lib.config_reload()
    return lambda: ops.score_pairs_train(Z, H, plans[c[0]], t, y, w)
ref = None
for c in configs:                                                   # warm every configuration (and bind the labels), check the bits per plan
    fn = setup(c)
    for _ in range(3): out = fn()
    if c[1] == 0 and c[2] == 0:
        ref = [v.clone() for v in out]
    else:
        assert all(torch.equal(a, b) for a, b in zip(out, ref)), ("bits differ", c)
times = {c: [] for c in configs}
for r in range(rounds):
    for c in (configs if r % 2 == 0 else configs[::-1]):
        fn = setup(c)
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); e1.synchronize()
        times[c].append(e0.elapsed_time(e1) / 10 * 1e3)
print(f"{name} K={K} d={d} {'bf16' if bf16 else 'f32'} P={P}: median / min us over {rounds} interleaved rounds of 10 launches (same bits within a plan: checked)")
for c in configs:
    v = np.array(times[c])
    print(f"  inc slices {c[0]:3d}  per-entry labels {c[1]}  in-launch row sums {c[2]}:  {np.median(v):8.1f} / {v.min():8.1f}", flush=True)
