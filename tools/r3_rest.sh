#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 2400 python -m pytest tests/test_raster_gpu.py tests/test_train_gpu.py tests/test_vit_gpu.py tests/test_x3_gpu.py -m gpu -q -x -s > $OUT/r3d_tests.log 2>&1
echo "tests rc=$?"; grep -E "passed|failed|Error|error|assert|B=64 train|max-rel" $OUT/r3d_tests.log | tail -40
