#!/bin/bash
# round 5, verdict item 4: would the edge scatter gain from the row-sum pass writing the normalised weight a / s~ through the
# reverse-edge map (so that the aggregation reads it in its per-entry stream instead of gathering s[col][p])?  Timing bounds
# from experiment builds (wrong numbers on purpose, tools/build_variant.py): nosgather = the aggregation without the
# gather (its best case), rowscatter = the row-sum pass with the extra read + scattered store (its cost), both.
# Alternating runs on the same box; per-phase microseconds from the bench line.
for rep in 1 2; do
for v in default nosgather rowscatter both; do
  for wl in squirrel_real chameleon; do
    if [ $v = default ]; then unset DL_LIB_PATH; else export DL_LIB_PATH=variants/libdisenlink_hip_$v.so; fi
    python3 bench.py --workload $wl --sections headline --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=l['kernels']
print('$v $wl rep$rep: step %.1f us  route %.1f  aggregate %.1f  score %.1f  edge_scatter %.1f' % (l['ms_per_step']*1e3, k['route']['avg_us'], k['aggregate']['avg_us'], k['score']['avg_us'], k['route']['avg_us']+k['aggregate']['avg_us']))"
  done
done
done
