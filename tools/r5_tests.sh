#!/bin/bash
# Round 5: the GPU tests added or touched this round (fast subset), then the default bench line.   usage: bash tools/r5_tests.sh
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -k "default_numerics or sync_batchnorm or reducer or mat_to_aa or fp32_matches_oracle_autograd or gemm_tn_group or hip_graph_replay or smpl" > $OUT/r05_new_tests.log 2>&1
echo "new tests rc=$?"; tail -15 $OUT/r05_new_tests.log
