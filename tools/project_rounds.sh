#!/bin/bash
# projection forward: kernel-only durations (rocprofv3 kernel trace) at a node count that makes whole rounds of workgroups
# (N = 4096: 32 node tiles x K = 8 x G groups), for G = 1, 2, 4 -> per-workgroup fixed cost and cost per pipeline step
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
for G in 1 2 4; do
  rm -rf /tmp/pr_$G
  DL_FWD_GROUPS=$G timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_$G -o out -- python3 $root/tools/project_fwd_quick.py 4096 128 8 512 64 5201 128 8 512 64 8192 128 8 512 64 > /tmp/pr_$G.log 2>&1 < /dev/null
  f=$(find /tmp/pr_$G -name '*kernel_trace.csv' | head -n 1)
  python3 - "$f" $G <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "project2_fwd" in n or "split_fwd" in n or "z_slab" in n:
        acc[(n.split("(")[0][-40:], r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X","?"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items(), key=lambda kv: kv[0][1]):
    v = sorted(v)
    print("G=%s %-42s grid %-8s calls %3d median %.1f us min %.1f" % (sys.argv[2], k[0], k[1], len(v), v[len(v)//2], v[0]))
PY
done
