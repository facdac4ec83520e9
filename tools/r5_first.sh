#!/bin/bash
# round 5, first GPU trip: the new one-pass-scorer parity tests + same-box reference timings of the Penn94 training scorer
set -o pipefail
mkdir -p gpurun_out/r5a
python -m pytest tests/test_gpu_parity.py -x -q -k "one_pass_training_scorer or compiled or trajectory_at_t2" > gpurun_out/r5a/tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r5a/tests.log
tail -5 gpurun_out/r5a/tests.log
python tools/score_train_time.py penn94 16 128 bf16 > gpurun_out/r5a/penn94_bf16_train.txt 2>&1 && tail -6 gpurun_out/r5a/penn94_bf16_train.txt
python tools/score_train_time.py penn94 16 128 f32 > gpurun_out/r5a/penn94_f32_train.txt 2>&1 && tail -6 gpurun_out/r5a/penn94_f32_train.txt
