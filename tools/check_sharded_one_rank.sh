#!/bin/bash
# round 5: multi-GPU hardening checked on the one-GPU box — tests of the sharded path, then the three blocks at one rank over RCCL
set -o pipefail
tag=${1:-r5f}
python -m pytest tests/test_gpu_parity.py -x -q -k "sharded or gpus_2 or back_to_back or routing_by_peer" > gpurun_out/${tag}_tests.log 2>&1
rc=$?; tail -3 gpurun_out/${tag}_tests.log; [ $rc -eq 0 ] || exit $rc
DL_FORCE_SHARDED=1 python3 bench.py --gpus 1 --steps 10 --warmup 3 > gpurun_out/${tag}_sharded_1rank_line.json 2> gpurun_out/${tag}_sharded.err || { tail -20 gpurun_out/${tag}_sharded.err; exit 1; }
python3 - <<PY
import json
l=json.loads(open('gpurun_out/${tag}_sharded_1rank_line.json').read().strip().splitlines()[-1])
print(l['config']['workload'][:260])
for n,b in l['blocks'].items():
    t=b.get('training_step',{})
    print(n,'ms',round(b['ms_per_step'],4),'n1',round(b['n1_same_problem_ms'],4),'speedup_vs_n1',round(b['speedup_vs_n1'],4),'| train',round(t.get('ms_per_step',0),3),t.get('n1_same_problem_ms'),t.get('speedup_vs_n1'),t.get('gradient_allreduce'),t.get('error'))
    print('   ', b['gather_ab']['z_gather_used'], b['gather_ab']['h_phase_used'], b['messages_per_step'])
PY
