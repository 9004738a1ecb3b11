#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_hotpath_gpu.py -m gpu -q -x > $OUT/r3h_glue.log 2>&1
echo "rc=$?"; tail -6 $OUT/r3h_glue.log
python tools/forward_census.py 64 2>/dev/null | head -30
python tools/forward_census.py 1 2>/dev/null | head -12
for B in 1 8 64; do python bench.py --workload whmr --batch $B --no-cpu --no-parity --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('whmr batch $B ms', round(d['ms_per_step'],4))"; done
