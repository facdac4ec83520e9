"""One-pass training scorer against the column slicing of the incidence plan (graph.PairList.build(inc_slices=...)).
usage: python tools/inc_slices_sweep.py <workload> <K> <d> <f32|bf16> <slices,slices,...> [seg_len,seg_len,...]
Every plan is built on the same pair list; the outputs of every plan are compared with the first one's (placement must
change speed only)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from disenlink_amd import ops
from disenlink_amd.graph import PairList, DEFAULT_INC_SEG_LEN
from disenlink_amd.metrics import pair_bce_weights
dev = torch.device("cuda:0")
name, K, d = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
bf16 = sys.argv[4] == "bf16"
slices = [int(s) for s in sys.argv[5].split(",")]
segs = [int(s) for s in sys.argv[6].split(",")] if len(sys.argv) > 6 else [DEFAULT_INC_SEG_LEN]
sg, split, graph, pairs, model, x, Z = bench.build_workload(name, dev, K, d, 512, elem_bytes=2 if bf16 else 4)
if bf16:
    Z = Z.to(torch.bfloat16)
t, beta = 1.0, 0.5
H = ops.aggregate_fwd(graph, Z, beta, *ops.route_fwd(graph, Z, t))
P = pairs.n_pairs
y = torch.from_numpy(pairs.bench_label).to(dev)
w = pair_bce_weights(int(y.sum().item()), P - int(y.sum().item()), 5, dev)
print(f"{name} K={K} d={d} {'bf16' if bf16 else 'f32'}: N={sg.n_nodes} P={P} default inc slices={pairs.inc.n_slices} fwd slices={pairs.by_u.n_slices}", flush=True)
ref = None
for seg in segs:
    for s in slices:
        pl = PairList.build(pairs.pu, pairs.pv, sg.n_nodes, seg_len=seg, row_bytes=K * d * (2 if bf16 else 4), inc_slices=s,
                            build_by_u=False)
        fn = lambda: ops.score_pairs_train(Z, H, pl, t, y, w)
        for _ in range(5): fn()
        best = 1e9
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fn()
            e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
        out = fn()
        if ref is None:
            ref = out
        dz = float((out[1].float() - ref[1].float()).abs().max() / ref[1].float().abs().max())
        dp = float((out[0] - ref[0]).abs().max())
        nseg, nslot = pl.inc.n_seg, pl.inc.n_slots
        print(f"  seg_len {seg:3d} inc_slices {s:3d}: {best:8.1f} us   segments {nseg} partial slots {nslot}   "
              f"max|dprob| {dp:.1e} max|ddZ|/max {dz:.1e}", flush=True)
        del pl
