#!/bin/bash
# round 6, call 33: which of the two switches breaks the captured training step
for cfg in "0 0" "1 0" "0 1"; do
  set -- $cfg
  echo "heavy_first=$1 tz_tail=$2"
  WHMR_TRAIN_HEAVY_FIRST=$1 WHMR_TRAIN_TZ_TAIL=$2 timeout 600 python -m pytest tests/test_train_gpu.py -x -q -m gpu -k "hip_graph_replay" 2>&1 | grep -v "^  File\|^$" | tail -4
done
