#!/bin/bash
# usage: tools/epoch_sequence.sh <tag> [dataset]  ->  gpurun_out/<tag>_epoch_sequence.txt
# Kernel trace of tools/epoch_once.py; prints the launches of ONE steady eager epoch (between two consecutive launches of the
# training scorer, late in the run) in order: start offset, duration, gap to the previous kernel's end, name.
set -u
tag=$1; ds=${2:-squirrel}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=/tmp/seq_$tag
rm -rf "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$out" -o out -- python3 "$root/tools/epoch_once.py" "$ds" > "$out.log" 2>&1 < /dev/null
f=$(find "$out" -name '*kernel_trace.csv' | head -n 1)
if [ -z "$f" ]; then echo "no kernel_trace.csv"; tail -n 5 "$out.log"; exit 1; fi
python3 - "$f" > "$root/gpurun_out/${tag}_epoch_sequence.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"]
marks = [i for i, r in enumerate(rows) if "score_train" in name(r)]
a, b = marks[-6], marks[-5]                     # one epoch, late in the run
t0 = int(rows[a]["Start_Timestamp"]); prev_end = None; busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0 if prev_end is None else s - prev_end
    busy += e - s
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap / 1e3:7.1f}  {name(r)[:110]}")
    prev_end = e
t1 = int(rows[b]["Start_Timestamp"])
print(f"epoch {(t1 - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us, {b - a} launches")
PY
tail -n 3 "$root/gpurun_out/${tag}_epoch_sequence.txt"
