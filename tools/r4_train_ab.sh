#!/bin/bash
# one-pass training scorer: group-per-entry kernel (round 3, DL_TRAIN_GROUP_KERNEL=1) vs wave-per-entry kernel (round 4)
for wl in squirrel_real chameleon; do
  echo "== $wl: group kernel"; DL_TRAIN_GROUP_KERNEL=1 python tools/score_train_time.py $wl 2>&1 | grep -v separate
  echo "== $wl: wave kernel"; python tools/score_train_time.py $wl 2>&1 | grep -v separate
done
