#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 600 python tools/r6_cam_graphs.py bf16 2>&1 | grep -v amdgpu.ids
timeout 600 python tools/r6_cam_graphs.py bf16x3 2>&1 | grep -v amdgpu.ids
timeout 1200 python -m pytest tests/test_train_gpu.py -m gpu -q -x -s -k "ddp_wrap" 2>&1 | grep -v "^\[W\|^$" | tail -12
timeout 600 python -m pytest tests/test_blocked_gpu.py -m gpu -q -x -k "layernorm_blk" 2>&1 | tail -3
