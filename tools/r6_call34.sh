#!/bin/bash
mkdir -p gpurun_out
for cfg in "1 0" "0 1"; do
  set -- $cfg
  echo "heavy_first=$1 tz_tail=$2"
  WHMR_TRAIN_HEAVY_FIRST=$1 WHMR_TRAIN_TZ_TAIL=$2 timeout 600 python -m pytest tests/test_train_gpu.py -x -q -s -m gpu -k "hip_graph_replay" > gpurun_out/r6_graph_probe_$1$2.log 2>&1
  grep -v "^  File\|^$\|Extension modules" gpurun_out/r6_graph_probe_$1$2.log | head -30
done
