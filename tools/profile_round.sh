#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
# Everything the bench line's numbers are checked against, into gpurun_out/<tag>_*:
#   <tag>_bench_line.json                          python bench.py (the driver's command, default flags)
#   <tag>_headline_kernel_stats.csv                rocprofv3 --kernel-trace --stats -- python3 bench.py --sections headline
#   <tag>_hbm_bound_kernel_stats.csv               ... --sections hbm_bound   (one workload per trace: averages stay attributable)
#   <tag>_pmc_traffic.json                         FETCH_SIZE / WRITE_SIZE, separate passes, per section (tools/pmc_traffic.py)
# Copy the files into profiles/ afterwards (tools/stats_md.py turns a csv into the markdown table).
set -u
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$root"
bash tools/prof_stats.sh ${tag}_headline bench.py --sections headline --steps 20 --warmup 5 --no-cpu-baseline || exit 1
bash tools/prof_stats.sh ${tag}_hbm_bound bench.py --sections hbm_bound --no-cpu-baseline || exit 1
bash tools/pmc_traffic_run.sh $tag squirrelx1_K8_d64_f32 --sections headline --steps 5 --warmup 2 || exit 1
bash tools/pmc_traffic_run.sh $tag snap_patentsx0.25_K8_d64_f32 --sections hbm_bound --hbm-steps 2 --repeats 2 || exit 1
# the bench line last, with the fresh PMC summary in place so that its `traffic` fields are this build's
cp gpurun_out/${tag}_pmc_traffic.json profiles/pmc_traffic_latest.json
python3 bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.err || { tail -n 5 gpurun_out/${tag}_bench.err; exit 1; }
cp profiles/pmc_traffic_latest.json gpurun_out/${tag}_pmc_traffic_latest.json
echo "profile_round $tag done"
