"""AucPlan.auc on the GPU at the bench workload's validation-set size (and DL_AUC_TARGET sweeps).
usage: python tools/auc_time.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disenlink_amd.metrics import AucPlan
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for n_pos, n_neg in ((10804, 54016), (1800, 9000), (68000, 340000)):
    y = torch.cat([torch.ones(n_pos), torch.zeros(n_neg)])[torch.randperm(n_pos + n_neg)].to(dev)
    sc = torch.rand(n_pos + n_neg, device=dev)
    plan = AucPlan(y)
    out = []
    for tgt in (None, 128, 256, 512, 1024, 2048):
        if tgt is None: os.environ.pop("DL_AUC_TARGET", None)
        else: os.environ["DL_AUC_TARGET"] = str(tgt)
        for _ in range(3): plan.auc(sc)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): plan.auc(sc)
        e1.record(); e1.synchronize()
        out.append(f"{'auto' if tgt is None else tgt}:{e0.elapsed_time(e1) / 20 * 1e3:.1f}")
    os.environ.pop("DL_AUC_TARGET", None)
    print(f"n_pos={n_pos} n_neg={n_neg}: us per AUC (incl. the division)  " + "  ".join(out), flush=True)
