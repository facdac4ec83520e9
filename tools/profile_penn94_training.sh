#!/bin/bash
# round 5: Penn94-shaped K=16 d=128 bf16 (configs[4] on one GPU): full-size parity of the wide one-pass scorer, the training step's line, its kernel trace
set -o pipefail
tag=${1:-r5c}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_fullsize.py -x -q -k penn94 > gpurun_out/${tag}_fullsize_penn94.log 2>&1
rc=$?; tail -3 gpurun_out/${tag}_fullsize_penn94.log; [ $rc -eq 0 ] || exit $rc
python3 bench.py --workload penn94 --K 16 --d 128 --dtype bf16 --sections headline,fwd_bwd --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/${tag}_penn94_K16_d128_bf16_train_line.json 2> gpurun_out/${tag}_bench.err || { tail -5 gpurun_out/${tag}_bench.err; exit 1; }
python3 -c "
import json,sys
l=json.loads(open('gpurun_out/${tag}_penn94_K16_d128_bf16_train_line.json').read().strip().splitlines()[-1])
print('ms_per_step', l['ms_per_step'], 'fwd_bwd', json.dumps(l.get('fwd_bwd'))[:1500])"
bash tools/prof_stats.sh ${tag}_penn94_training bench.py --workload penn94 --K 16 --d 128 --dtype bf16 --sections fwd_bwd --steps 10 --warmup 3 --no-cpu-baseline --warm-s 0 --min-region-s 0
