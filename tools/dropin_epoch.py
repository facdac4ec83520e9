"""The reference's loop (main_disentangled.py:192-214, dense masks) around the drop-in module for a number of epochs — the
subject of `rocprofv3 --kernel-trace --stats` for the drop-in path (tools/prof_stats.sh <tag> tools/dropin_epoch.py ...).
usage: python tools/dropin_epoch.py [workload] [epochs] [default|static]
Prints ms per epoch (wall) and the launches per epoch seen by the host."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "squirrel_real"
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 30
static = len(sys.argv) > 3 and sys.argv[3] == "static"
sg, split, graph, pairs, model, x, Z = bench.build_workload(name, dev, 8, 64, 512)
del graph, pairs, model, Z
out = bench.dropin_epoch_section(dev, 8, 64, 512, split, x, (sg.src, sg.dst), name, epochs=epochs, static_masks=(static,))
key = "static_masks" if static else "default"
print(f"{name} {key}: {out[key]['ms_per_epoch']:.3f} ms per epoch, pair plan built {out[key]['pair_plan_builds_in_the_timed_epochs']}x; "
      f"dl_score_allpairs_bwd {out['dl_score_allpairs_bwd_us']:.1f} us over {out['dl_score_allpairs_bwd_pairs']} pairs")
for k, v in out[key]["stages_ms_synchronised"].items():
    print(f"   {v:7.3f} ms  {k}")
