#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_train_gpu.py -m gpu -q -x -s -k "gate_flips" > $OUT/r3k.log 2>&1
echo "rc=$?"; grep -E "passed|failed|skipped|Error|assert|deconv chain" $OUT/r3k.log | tail -8
