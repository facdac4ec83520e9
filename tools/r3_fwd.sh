#!/bin/bash
# usage (GPU box): bash tools/r3_fwd.sh [runs] -> forward step of squirrel_real and chameleon
for i in $(seq 1 ${1:-3}); do for w in squirrel_real chameleon; do
python3 bench.py --workload $w --sections headline --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('%-14s step %.1f us: route %.1f agg %.1f score %.1f' % ('$w', d['ms_per_step']*1e3, k['route']['avg_us'], k['aggregate']['avg_us'], k['score']['avg_us']))"
done; done
