"""Forward scorer: wave-per-entry kernel (round 6) against the group-per-entry kernel (DL_FWD_GROUP_KERNEL=1), interleaved in
one process on the bench workload; probabilities compared with each other, with the one-pass training scorer's (same bits
expected from the wave kernel) and the stored per-factor terms.  usage: python tools/fwd_scorer_ab.py [workload] [K] [d] [t]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from disenlink_amd import _lib, ops
from disenlink_amd.metrics import pair_bce_weights
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "squirrel_real"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d = int(sys.argv[3]) if len(sys.argv) > 3 else 64
t = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
sg, split, graph, pairs, model, x, Z = bench.build_workload(name, dev, K, d, 512)
H = ops.aggregate_fwd(graph, Z, 0.5, *ops.route_fwd(graph, Z, t))
P = pairs.n_pairs

def mode(group):
    if group: os.environ["DL_FWD_GROUP_KERNEL"] = "1"
    else: os.environ.pop("DL_FWD_GROUP_KERNEL", None)
    _lib.config_reload()
fwd = lambda: ops.score_pairs_fwd(Z, H, pairs.pu, pairs.pv, t, pairs)
res = {}
for group in (True, False):
    mode(group)
    res[group] = (fwd().clone(), [v.clone() for v in ops.score_pairs_fwd(Z, H, pairs.pu, pairs.pv, t, pairs, want_coef=True)])
y = torch.from_numpy(pairs.bench_label).to(dev)
w = pair_bce_weights(int(y.sum().item()), P - int(y.sum().item()), 5, dev)
p_train = ops.score_pairs_train(Z, H, pairs, t, y, w)[0]
pg, pw = res[True][0], res[False][0]
print(f"{name} K={K} d={d} t={t}: max|prob wave - prob group| {float((pg - pw).abs().max()):.2e}; wave == training scorer's prob: {torch.equal(pw, p_train)}; "
      f"coef pass prob == plain: {torch.equal(res[False][1][0], pw)}; max rel |coef wave - coef group| "
      f"{float(((res[False][1][1] - res[True][1][1]).abs() / res[True][1][1].abs().clamp_min(1e-30)).max()):.2e}")
times = {True: [], False: []}
for r in range(12):
    for group in ((True, False) if r % 2 == 0 else (False, True)):
        mode(group)
        fwd()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fwd()
        e1.record(); e1.synchronize()
        times[group].append(e0.elapsed_time(e1) / 20 * 1e3)
mode(False)
print(f"  group per entry {np.median(times[True]):7.1f} / {min(times[True]):7.1f} us   wave per entry {np.median(times[False]):7.1f} / {min(times[False]):7.1f} us   (median / min of 12 interleaved rounds of 20)")
