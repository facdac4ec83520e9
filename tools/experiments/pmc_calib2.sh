#!/bin/bash
# read-request sizes of the calibration kernels: gpurun_out/pmc_calib2.txt
root=${GRAFT_REPO_ROOT:-$(pwd)}
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 "$root/tools/experiments/pmc_calib.hip" -o /tmp/pmc_calib || exit 1
cd /tmp && export TMPDIR=/tmp
out="$root/gpurun_out/pmc_calib2.txt"
: > "$out"
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_WR[A-Z0-9_]*\|TCC_EA0_RD[A-Z0-9_]*" | sort -u | tr '\n' ' ' >> "$out"; echo >> "$out"
for c in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_sum"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-60)
  rm -rf /tmp/cal_$tag
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/cal_$tag -o out -- /tmp/pmc_calib > /tmp/cal_$tag.log 2>&1 < /dev/null || { echo "pass $c failed" >> "$out"; tail -n 3 /tmp/cal_$tag.log >> "$out"; continue; }
  f=$(find /tmp/cal_$tag -name '*counter_collection.csv' | head -n 1)
  python3 - "$f" >> "$out" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    if not k.startswith("__amd"): print(f"{k:12s} {c:28s} mean {sum(v)/len(v):16.1f}  (x{len(v)})")
PY
done
cat "$out"
