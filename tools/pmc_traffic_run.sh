#!/bin/bash
# usage: tools/pmc_traffic_run.sh <tag> <key> <bench.py args...>  ->  gpurun_out/<tag>_pmc_traffic.json (entry <key>)
#   e.g. tools/pmc_traffic_run.sh r2a squirrelx1_K8_d64_f32 --sections headline --steps 5 --warmup 2
#        tools/pmc_traffic_run.sh r2a snap_patentsx0.25_K8_d64_f32 --sections hbm_bound
# Two separate rocprofv3 PMC passes (FETCH_SIZE, then WRITE_SIZE; --kernel-trace only) of the bench command,
# summarised by tools/pmc_traffic.py.  Run from the repo root on the GPU box; the program follows `--` directly.
set -u
tag=$1; key=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o out -- python3 "$root/bench.py" "$@" --no-cpu-baseline --warm-s 0 --min-region-s 0 > /tmp/pmc_$c.log 2>&1 < /dev/null || { echo "pass $c failed"; tail -n 5 /tmp/pmc_$c.log; exit 1; }
done
f=$(find /tmp/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -n 1)
w=$(find /tmp/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -n 1)
if [ -z "$f" ] || [ -z "$w" ]; then echo "PMC output missing"; tail -n 5 /tmp/pmc_FETCH_SIZE.log; exit 1; fi
python3 "$root/tools/pmc_traffic.py" "$f" "$w" "$root/gpurun_out/${tag}_pmc_traffic.json" "$key" \
  "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py $* --no-cpu-baseline --warm-s 0 --min-region-s 0"
