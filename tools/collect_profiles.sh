#!/bin/bash
# usage (in the build container, after `gpurun -- bash tools/profile_round.sh <tag>`): bash tools/collect_profiles.sh <tag>
# copies the judged summaries from gpurun_out/ (scratch) into profiles/ (tracked) and writes <tag>_kernel_stats.md
set -e
tag=$1
cd "$(dirname "$0")/.."
for f in bench_line.json chameleon_bench_line.json penn94_K16_d128_bf16_bench_line.json chameleon_kernel_stats.csv \
         hbm_bound_kernel_stats.csv headline_kernel_stats.csv penn94_K16_d128_bf16_kernel_stats.csv pmc_traffic.json \
         training_kernel_stats.csv pmc_l2.json; do
  cp gpurun_out/${tag}_$f profiles/${tag}_$f
done
cp gpurun_out/${tag}_pmc_traffic_latest.json profiles/pmc_traffic_latest.json
cp gpurun_out/${tag}_pmc_l2_latest.json profiles/pmc_l2_latest.json
(cd profiles && for w in headline training hbm_bound chameleon penn94_K16_d128_bf16; do
   echo "## $w"; echo; python3 ../tools/stats_md.py ${tag}_${w}_kernel_stats.csv 10; echo; done) > profiles/${tag}_kernel_stats.md
python3 - <<PY
import json
d = json.loads(open("profiles/${tag}_bench_line.json").read().strip().splitlines()[-1])
print("headline", round(d["value"] / 1e9, 3), "G edges/s", round(d["ms_per_step"], 4), "ms; score frac", round(d["roofline"]["frac"], 3),
      "traffic", d["roofline"]["traffic"], "; edge scatter", round(d["edge_scatter"]["avg_us"], 1), "us; fwd_bwd",
      round(d["fwd_bwd"]["ms_per_step"], 3), "ms; one pass", round(d["scorer_training_step"]["one_pass_us"], 1), "us")
hb = d["hbm_bound"]
print("hbm_bound", round(hb["ms_per_step"], 2), "ms", {k: (round(v["frac"], 3), round(v["avg_us"], 1)) for k, v in hb["kernels"].items()})
print("projection fwd", round(d["projection"]["fwd"]["avg_us"], 1), "bwd", round(d["projection"]["bwd"]["avg_us"], 1))
PY
