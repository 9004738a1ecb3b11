#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_x3_gpu.py tests/test_blocked_gpu.py -m gpu -q -x -s -k "x3 or fold" > $OUT/r3i.log 2>&1
echo "rc=$?"; grep -E "passed|failed|Error|assert|folded|bf16x3 ViT" $OUT/r3i.log | tail -12
timeout 900 python -m pytest tests/test_hotpath_gpu.py -m gpu -q -x -k "bf16x3" > $OUT/r3i2.log 2>&1; echo "hot rc=$?"; tail -2 $OUT/r3i2.log
python bench.py --no-cpu --no-secondary --numerics bf16x3 --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('x3 ms', round(d['ms_per_step'],4), 'issue', round(d['roofline']['mfma_issue_frac'],4))"
python tools/forward_census.py 64 bf16x3 2>/dev/null | head -32
