#!/bin/bash
# A/B of the unit order inside a slice (DL_PLAN_SORT=0: by length, unset: entry order), alternating on one box
for rep in 1 2 3; do
for v in "" 0; do
for w in squirrel_real chameleon; do
env ${v:+DL_PLAN_SORT=$v} python3 bench.py --workload $w --sections headline,fwd_bwd,scorer_train --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['scorer_training_step']; k=d['kernels']
print('sort=%-2s %-14s step %.1f us: route %.1f agg %.1f score %.1f | one_pass %.1f separate %.1f | fwd_bwd %.4f ms' % ('$v', '$w', d['ms_per_step']*1e3, k['route']['avg_us'], k['aggregate']['avg_us'], k['score']['avg_us'], s['one_pass_us'], s['separate_us'], d['fwd_bwd']['ms_per_step']))"
done; done; done
for v in "" 0; do
env ${v:+DL_PLAN_SORT=$v} python3 bench.py --workload penn94 --K 16 --d 128 --dtype bf16 --sections headline --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('sort=%-2s penn94 bf16 step %.1f us: route %.1f agg %.1f score %.1f' % ('$v', d['ms_per_step']*1e3, k['route']['avg_us'], k['aggregate']['avg_us'], k['score']['avg_us']))"
done
