#!/bin/bash
# usage: tools/prof_stats.sh <tag> <script.py> [args...]   ->  gpurun_out/<tag>_kernel_stats.csv (top rows printed)
# rocprofv3 --kernel-trace --stats over one python script; run from the repo root on the GPU box.
set -u
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=/tmp/prof_$tag
rm -rf "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o out -- python3 "$root/$1" "${@:2}" > "$out.log" 2>&1 < /dev/null
f=$(find "$out" -name '*kernel_stats.csv' | head -n 1)
if [ -z "$f" ]; then echo "no kernel_stats.csv for $tag"; tail -n 5 "$out.log"; exit 1; fi
cp "$f" "$root/gpurun_out/${tag}_kernel_stats.csv"
grep -v "amdgpu.ids" "$out.log" | tail -n 40 > "$root/gpurun_out/${tag}_stdout.txt"      # what the script itself printed
echo "== $tag: $*"
cut -d, -f1-4 "$f" < /dev/null | sed -n 1,9p
