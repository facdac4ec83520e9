#!/bin/bash
# usage (GPU box): bash tools/r3_flags_ab.sh "<flagsA>" "<flagsB>" ...  -> headline + training sections with each hipcc flag set
for fl in "$@"; do
  DL_CXXFLAGS="$fl" python3 -m disenlink_amd.build --force > /dev/null 2> gpurun_out/flags_ab_build.err || { tail -3 gpurun_out/flags_ab_build.err; exit 1; }
  for rep in 1 2; do
  python3 bench.py --sections headline,fwd_bwd,scorer_train --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['scorer_training_step']; k=d['kernels']
print('%-28s step %.1f us: route %.1f agg %.1f score %.1f | one_pass %.1f separate %.1f | fwd_bwd %.4f ms' % ('$fl', d['ms_per_step']*1e3, k['route']['avg_us'], k['aggregate']['avg_us'], k['score']['avg_us'], s['one_pass_us'], s['separate_us'], d['fwd_bwd']['ms_per_step']))"
  done
done
python3 -m disenlink_amd.build --force > /dev/null
