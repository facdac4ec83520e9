#!/bin/bash
# usage (GPU box): bash tools/r3_train_ab.sh "<flagsA>" "<flagsB>" ...  -> the training scorer with each hipcc flag set (rebuilds the
# library on the box), on the default workload and on the HBM-sized one
for fl in "$@"; do
  DL_CXXFLAGS="$fl" python3 -m disenlink_amd.build --force > /dev/null 2> gpurun_out/train_ab_build.err || { tail -3 gpurun_out/train_ab_build.err; exit 1; }
  echo "flags: '$fl'"
  bash tools/r3_train.sh 1
  python3 bench.py --workload snap_patents --scale 0.25 --sections headline,fwd_bwd,scorer_train --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['scorer_training_step']
print('snap x0.25: one_pass %.0f us separate %.0f forward %.0f | fwd_bwd %.2f ms' % (s['one_pass_us'], s['separate_us'], s['forward_us'], d['fwd_bwd']['ms_per_step']))"
done
python3 -m disenlink_amd.build --force > /dev/null
