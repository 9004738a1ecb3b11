#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests/test_train_gpu.py -m gpu -q -x -k "whmr_train_step or smpl_backward or regressor_post or downsample or conv_linear" 2>&1 | tail -4
for i in 1 2 3; do python bench.py --workload whmr_train --no-cpu --steps 30 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train ms', round(d['ms_per_step'],3))"; done
timeout 600 python tools/train_launch_census.py 70 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tail -75
