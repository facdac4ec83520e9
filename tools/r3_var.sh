#!/bin/bash
# usage (GPU box): bash tools/r3_var.sh <tag> "<ENV=.. ENV=..>" [sections]  -> one bench line under the given environment
tag=$1; envs=$2; sec=${3:-headline,hbm_bound}
env $envs python3 bench.py --sections $sec --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/${tag}.json 2> gpurun_out/${tag}.err || { tail -5 gpurun_out/${tag}.err; exit 1; }
python3 - <<PY
import json
d=json.load(open("gpurun_out/${tag}.json"))
k=d["kernels"]; line=f"${tag} [${envs}] squirrel: step {d['ms_per_step']*1e3:.1f} us  route {k['route']['avg_us']:.1f} agg {k['aggregate']['avg_us']:.1f} score {k['score']['avg_us']:.1f}"
h=d.get("hbm_bound")
if h:
    hk=h["kernels"]; line+=f" | hbm: route {hk['route']['avg_us']:.0f} agg {hk['aggregate']['avg_us']:.0f} (frac {hk['aggregate']['frac']:.2f}) score {hk['score']['avg_us']:.0f}"
print(line)
PY
