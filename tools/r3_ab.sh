#!/bin/bash
# usage (GPU box): bash tools/r3_ab.sh <tag> [sections]  -> bench lines of the aggregate kernel forms, side by side
tag=$1; sec=${2:-headline,hbm_bound}
python3 bench.py --sections $sec --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/${tag}_cls4.json 2> gpurun_out/${tag}_cls4.err || { tail -5 gpurun_out/${tag}_cls4.err; exit 1; }
DL_AGG_DEPTH=2 python3 bench.py --sections $sec --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/${tag}_cls2.json 2> gpurun_out/${tag}_cls2.err || exit 1
DL_AGG_FORM=groups python3 bench.py --sections $sec --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/${tag}_groups.json 2> gpurun_out/${tag}_groups.err || exit 1
python3 - <<PY
import json
for f in ("cls4","cls2","groups"):
    d=json.load(open("gpurun_out/${tag}_%s.json"%f))
    k=d["kernels"]; line=f"{f:7s} squirrel: step {d['ms_per_step']*1e3:.1f} us  route {k['route']['avg_us']:.1f} agg {k['aggregate']['avg_us']:.1f} score {k['score']['avg_us']:.1f}"
    h=d.get("hbm_bound")
    if h:
        hk=h["kernels"]; line+=f" | hbm: route {hk['route']['avg_us']:.0f} agg {hk['aggregate']['avg_us']:.0f} (frac {hk['aggregate']['frac']:.2f}) score {hk['score']['avg_us']:.0f}"
    print(line)
PY
