#!/bin/bash
# one-pass training scorer: group-per-entry kernel (round 3) vs wave-per-entry kernel (round 4), plain and pipelined
for wl in squirrel_real chameleon; do
  echo "== $wl: group kernel"; DL_TRAIN_GROUP_KERNEL=1 python tools/score_train_time.py $wl 2>&1 | grep -v separate
  echo "== $wl: wave kernel, plain"; DL_TRAIN_PIPE=0 python tools/score_train_time.py $wl 2>&1 | grep -v separate
  echo "== $wl: wave kernel, pipelined"; DL_TRAIN_PIPE=1 python tools/score_train_time.py $wl 2>&1 | grep -v separate
done
