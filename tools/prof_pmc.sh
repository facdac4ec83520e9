#!/bin/bash
# usage: tools/prof_pmc.sh <tag> "<COUNTER ...>" <script.py> [args...] -> gpurun_out/<tag>_pmc.csv (per-kernel means printed)
# One rocprofv3 --pmc pass (with --kernel-trace only) over one python script; run from the repo root on the GPU box.
set -u
tag=$1; ctrs=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=/tmp/pmc_$tag
rm -rf "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$out" -o out -- python3 "$root/$1" "${@:2}" > "$out.log" 2>&1 < /dev/null
f=$(find "$out" -name '*counter_collection.csv' | head -n 1)
if [ -z "$f" ]; then echo "no counter_collection.csv for $tag"; tail -n 8 "$out.log"; exit 1; fi
cp "$f" "$root/gpurun_out/${tag}_pmc.csv"
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    name = row["Kernel_Name"].split("(")[0][-60:]
    acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, cs in acc.items():
    if "dl::" not in name: continue
    print(name, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "calls", len(next(iter(cs.values()))))
PY
