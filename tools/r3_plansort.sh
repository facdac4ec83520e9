#!/bin/bash
# A/B of the unit order inside a slice (graph.length_order): DL_PLAN_SORT=0 entry order, =1 by length (most entries first)
for rep in 1 2; do for v in 0 1; do for w in squirrel_real chameleon; do
DL_PLAN_SORT=$v python3 bench.py --workload $w --sections headline,fwd_bwd,scorer_train --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['scorer_training_step']; k=d['kernels']
print('sort=%-2s %-14s step %.1f us: route %.1f agg %.1f score %.1f | one_pass %.1f separate %.1f | fwd_bwd %.4f ms' % ('$v', '$w', d['ms_per_step']*1e3, k['route']['avg_us'], k['aggregate']['avg_us'], k['score']['avg_us'], s['one_pass_us'], s['separate_us'], d['fwd_bwd']['ms_per_step']))"
done; done; done
DL_PLAN_SORT=1 python3 bench.py --sections hbm_bound --no-cpu-baseline --hbm-steps 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); hk=d['hbm_bound']['kernels']
print('sort=1 hbm_bound (default: entry order): route %.0f agg %.0f score %.0f' % (hk['route']['avg_us'], hk['aggregate']['avg_us'], hk['score']['avg_us']))"
