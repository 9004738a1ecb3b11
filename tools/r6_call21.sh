#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "big_tile and 260" 2>&1 | tail -3
timeout 600 python tools/deconv_probe.py 2>&1 | grep -v amdgpu.ids
