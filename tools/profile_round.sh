#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
# Everything the bench line's numbers are checked against, into gpurun_out/<tag>_*:
#   <tag>_bench_line.json                          python bench.py (the driver's command, default flags)
#   <tag>_<workload>_kernel_stats.csv              rocprofv3 --kernel-trace --stats -- python3 bench.py --sections headline ...
#                                                  (one workload per trace: a kernel's average belongs to one problem size)
#   <tag>_pmc_traffic.json                         FETCH_SIZE / WRITE_SIZE, separate passes, per workload (tools/pmc_traffic.py)
# Workloads: the default headline (squirrel_real), the hbm_bound block (snap_patents x0.25), and the other BASELINE.json
# configurations on one GPU: chameleon K=8 d=64 fp32 (configs[1]) and Penn94-shaped K=16 d=128 bf16 (configs[4]).
# Copy the files into profiles/ afterwards (tools/stats_md.py turns a csv into the markdown table).
# A gpurun call is capped at 20 minutes: PARTS="1 2 3 4" (default: all) selects what runs; the PMC summaries of earlier parts
# are picked up from profiles/_partial/ (copy gpurun_out/<tag>_pmc_*.json there between calls — gpurun_out/ does not travel).
set -u
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$root"
PARTS=${PARTS:-"1 2 3 4"}
mkdir -p gpurun_out
for f in pmc_traffic pmc_l2; do
  [ -f profiles/_partial/${tag}_$f.json ] && [ ! -f gpurun_out/${tag}_$f.json ] && cp profiles/_partial/${tag}_$f.json gpurun_out/${tag}_$f.json
done
part() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }
if part 1; then
bash tools/prof_stats.sh ${tag}_headline bench.py --sections headline --steps 20 --warmup 5 --no-cpu-baseline || exit 1
bash tools/prof_stats.sh ${tag}_hbm_bound bench.py --sections hbm_bound --no-cpu-baseline || exit 1
bash tools/prof_stats.sh ${tag}_chameleon bench.py --workload chameleon --sections headline --steps 20 --warmup 5 --no-cpu-baseline || exit 1
bash tools/prof_stats.sh ${tag}_penn94_K16_d128_bf16 bench.py --workload penn94 --K 16 --d 128 --dtype bf16 --sections headline --steps 10 --warmup 3 --no-cpu-baseline || exit 1
bash tools/prof_stats.sh ${tag}_training bench.py --sections fwd_bwd,scorer_train --steps 20 --warmup 5 --no-cpu-baseline --warm-s 0 --min-region-s 0 || exit 1
bash tools/prof_stats.sh ${tag}_penn94_training bench.py --workload penn94 --K 16 --d 128 --dtype bf16 --sections fwd_bwd --steps 10 --warmup 3 --no-cpu-baseline --warm-s 0 --min-region-s 0 || exit 1
# the path the one-line swap gives: the reference's dense-mask loop around the drop-in module (tools/dropin_epoch.py)
bash tools/prof_stats.sh ${tag}_dropin_static tools/dropin_epoch.py squirrel_real 30 static || exit 1
bash tools/prof_stats.sh ${tag}_dropin_default tools/dropin_epoch.py squirrel_real 30 default || exit 1
fi
if part 2; then
bash tools/pmc_traffic_run.sh $tag squirrel_realx1_K8_d64_f32 --sections headline --steps 5 --warmup 2 || exit 1
bash tools/pmc_traffic_run.sh $tag squirrel_realx1_K8_d64_f32_train --sections fwd_bwd --steps 5 --warmup 2 || exit 1
# L2-side request counters (what `moved_bytes` is checked against): headline, training step, hbm_bound
bash tools/pmc_l2_run.sh $tag squirrel_realx1_K8_d64_f32 --sections headline --steps 5 --warmup 2 || exit 1
bash tools/pmc_l2_run.sh $tag squirrel_realx1_K8_d64_f32_train --sections fwd_bwd --steps 5 --warmup 2 || exit 1
bash tools/pmc_l2_run.sh $tag snap_patentsx0.25_K8_d64_f32 --sections hbm_bound --hbm-steps 2 --repeats 2 || exit 1
bash tools/pmc_l2_run.sh $tag chameleonx1_K8_d64_f32 --workload chameleon --sections headline --steps 5 --warmup 2 || exit 1
fi
if part 3; then
bash tools/pmc_l2_run.sh $tag penn94x1_K16_d128_bf16 --workload penn94 --K 16 --d 128 --dtype bf16 --sections headline --steps 3 --warmup 1 --repeats 2 || exit 1
bash tools/pmc_l2_run.sh $tag penn94x1_K16_d128_bf16_train --workload penn94 --K 16 --d 128 --dtype bf16 --sections fwd_bwd --steps 3 --warmup 1 || exit 1
bash tools/pmc_traffic_run.sh $tag penn94x1_K16_d128_bf16_train --workload penn94 --K 16 --d 128 --dtype bf16 --sections fwd_bwd --steps 3 --warmup 1 || exit 1
bash tools/pmc_traffic_run.sh $tag snap_patentsx0.25_K8_d64_f32 --sections hbm_bound --hbm-steps 2 --repeats 2 || exit 1
bash tools/pmc_traffic_run.sh $tag chameleonx1_K8_d64_f32 --workload chameleon --sections headline --steps 5 --warmup 2 || exit 1
bash tools/pmc_traffic_run.sh $tag penn94x1_K16_d128_bf16 --workload penn94 --K 16 --d 128 --dtype bf16 --sections headline --steps 3 --warmup 1 --repeats 2 || exit 1
fi
if part 4; then
# the bench lines last, with the fresh PMC summary in place so that their `traffic` fields are this build's
cp gpurun_out/${tag}_pmc_traffic.json profiles/pmc_traffic_latest.json
cp gpurun_out/${tag}_pmc_l2.json profiles/pmc_l2_latest.json
python3 bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.err || { tail -n 5 gpurun_out/${tag}_bench.err; exit 1; }
python3 bench.py --workload chameleon --sections headline,cpu --steps 20 --warmup 5 > gpurun_out/${tag}_chameleon_bench_line.json 2>> gpurun_out/${tag}_bench.err || exit 1
# configs[4] and configs[3] on one GPU, WITH Baseline B (the C edge-list restatement on the host cores) and the parity of the timed step against it
python3 bench.py --workload penn94 --K 16 --d 128 --dtype bf16 --sections headline,fwd_bwd,cpu --steps 10 --warmup 3 > gpurun_out/${tag}_penn94_K16_d128_bf16_bench_line.json 2>> gpurun_out/${tag}_bench.err || exit 1
python3 bench.py --workload snap_patents --sections headline,cpu --steps 5 --warmup 2 --repeats 3 > gpurun_out/${tag}_snap_patents_bench_line.json 2>> gpurun_out/${tag}_bench.err || exit 1
cp profiles/pmc_traffic_latest.json gpurun_out/${tag}_pmc_traffic_latest.json
cp profiles/pmc_l2_latest.json gpurun_out/${tag}_pmc_l2_latest.json
fi
echo "profile_round $tag parts $PARTS done"
