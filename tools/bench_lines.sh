#!/bin/bash
# usage (GPU box, repo root): bash tools/bench_lines.sh <tag>  ->  gpurun_out/<tag>_*bench_line.json
# The three bench lines of a round (default command, chameleon, Penn94-shaped K=16 d=128 bf16) with the PMC summaries already
# under profiles/ — the tail of tools/profile_round.sh, for when only bench.py changed since the counter passes.
set -u
tag=$1
python3 bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.err || { tail -n 5 gpurun_out/${tag}_bench.err; exit 1; }
python3 bench.py --workload chameleon --sections headline,cpu --steps 20 --warmup 5 > gpurun_out/${tag}_chameleon_bench_line.json 2>> gpurun_out/${tag}_bench.err || exit 1
# configs[4] and configs[3] on one GPU, WITH Baseline B (the C edge-list restatement on the host cores) and the parity of the timed step against it
python3 bench.py --workload penn94 --K 16 --d 128 --dtype bf16 --sections headline,fwd_bwd,cpu --steps 10 --warmup 3 > gpurun_out/${tag}_penn94_K16_d128_bf16_bench_line.json 2>> gpurun_out/${tag}_bench.err || exit 1
python3 bench.py --workload snap_patents --sections headline,cpu --steps 5 --warmup 2 --repeats 3 > gpurun_out/${tag}_snap_patents_bench_line.json 2>> gpurun_out/${tag}_bench.err || exit 1
echo "bench_lines $tag done"
