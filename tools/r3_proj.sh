#!/bin/bash
# kernel-level times of the projection forward at the bench shape and the wide-feature shape
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for shape in "5201 128 8 512 64" "5201 2088 8 512 64"; do
  tag=$(echo $shape | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/proj_$tag -o p -- python3 $R/tools/project_once.py $shape 30 > $R/gpurun_out/proj_$tag.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/proj_$tag/**/p_kernel_stats.csv",recursive=True)[0]
print("$shape")
for r in csv.DictReader(open(f)):
    if float(r['Percentage'])>0.5: print("  ",r['Name'][:60].ljust(60), r['Calls'], "%.1f us"%(float(r['AverageNs'])/1e3), "min %.1f"%(float(r['MinNs'])/1e3))
PY
  tail -1 $R/gpurun_out/proj_$tag.log
done
for g in 1 2 4; do DL_FWD_GROUPS=$g python3 $R/tools/project_once.py 5201 128 8 512 64 50 | sed "s/^/G=$g /"; done
