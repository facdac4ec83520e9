#!/bin/bash
# round 5: the wide wave-per-entry training scorer (K=16, d=128): parity tests, then same-box timings of the variants
set -o pipefail
mkdir -p gpurun_out/r5b
python -m pytest tests/test_gpu_parity.py -x -q -k "one_pass_training_scorer" > gpurun_out/r5b/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/r5b/tests.log
[ $rc -eq 0 ] || exit $rc
for dt in bf16 f32; do
  echo "== penn94 16 128 $dt: default build"; timeout -k 10 300 python tools/score_train_time.py penn94 16 128 $dt 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r5b/penn94_$dt.txt
done
echo "== penn94 16 128 bf16: U=2, 3 waves"; DL_LIB_PATH=variants/libdisenlink_hip_wideU2.so timeout -k 10 300 python tools/score_train_time.py penn94 16 128 bf16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5b/penn94_bf16_U2.txt
echo "== penn94 16 128 bf16: group kernel"; DL_TRAIN_GROUP_KERNEL=1 timeout -k 10 300 python tools/score_train_time.py penn94 16 128 bf16 2>&1 | grep -v amdgpu.ids | head -2
