#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_blocked_gpu.py -m gpu -q -x -k "chain" 2>&1 | tail -5
timeout 900 python tools/r6_chain_ab.py 2>&1 | grep -v amdgpu.ids | tee $OUT/r6_chain_ab.txt
timeout 900 python -m pytest tests/test_hotpath_gpu.py -m gpu -q -x -k "side_stream" 2>&1 | tail -3
