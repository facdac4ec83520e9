#!/bin/bash
# usage (GPU box): bash tools/r3_train.sh [runs]  -> the scorer's training step and forward+backward of the default workload
for i in $(seq 1 ${1:-3}); do
python3 bench.py --sections headline,fwd_bwd,scorer_train --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['scorer_training_step']
print('one_pass %.1f us  separate %.1f  forward %.1f | fwd_bwd %.4f ms' % (s['one_pass_us'], s['separate_us'], s['forward_us'], d['fwd_bwd']['ms_per_step']))"
done
