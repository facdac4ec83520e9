#!/bin/bash
# routing kernel: (value, index) exchange arg-max vs ballot arg-max (DL_ROUTE_BALLOT=1), same box, alternating
for i in 1 2; do
for wl in squirrel_real chameleon; do
  for b in 0 1; do
    echo "== $wl ballot=$b"; DL_ROUTE_BALLOT=$b python bench.py --workload $wl --sections headline --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read()); print({k:round(v['avg_us'],2) for k,v in b['kernels'].items()}, round(b['ms_per_step']*1e3,1))"
  done
done
done
