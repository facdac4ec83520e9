#!/bin/bash
# usage: tools/pmc_l2_run.sh <tag> <key> <bench.py args...>  ->  gpurun_out/<tag>_pmc_l2.json (entry <key>)
# L2-SIDE request counters of the kernels (what the compute units ask of the XCD L2s — against which bench.py's
# `moved_bytes` model is checked; FETCH_SIZE / WRITE_SIZE of pmc_traffic_run.sh count what LEAVES the L2s):
#   pass 1: TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum     pass 2: TCC_HIT_sum TCC_MISS_sum
# rocprofv3 with --kernel-trace only, the program directly after `--`.  Units (profiles/r3_pmc_calibration.txt, measured
# on this kernel family's access shapes): one read request = one 128-B line, one write request = 64 B.
set -u
tag=$1; key=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for c in "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmc_l2_$i
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_l2_$i -o out -- python3 "$root/bench.py" "$@" --no-cpu-baseline --warm-s 0 --min-region-s 0 > /tmp/pmc_l2_$i.log 2>&1 < /dev/null || { echo "pass $i ($c) failed"; tail -n 5 /tmp/pmc_l2_$i.log; exit 1; }
done
a=$(find /tmp/pmc_l2_1 -name '*counter_collection.csv' | head -n 1)
b=$(find /tmp/pmc_l2_2 -name '*counter_collection.csv' | head -n 1)
if [ -z "$a" ] || [ -z "$b" ]; then echo "PMC output missing"; tail -n 5 /tmp/pmc_l2_1.log; exit 1; fi
python3 "$root/tools/pmc_l2.py" "$a" "$b" "$root/gpurun_out/${tag}_pmc_l2.json" "$key" \
  "rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum | TCC_HIT_sum TCC_MISS_sum (separate passes) -- python3 bench.py $* --no-cpu-baseline --warm-s 0 --min-region-s 0"
