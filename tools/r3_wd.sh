#!/bin/bash
# A/B of the blocked GEMM main loop: schedule 1 (W through LDS) vs 2 (W direct), bf16 and bf16x3, + the bit-identity test
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_blocked_gpu.py -m gpu -q -x -k "w_direct" > $OUT/r3g_wd.log 2>&1
echo "wd rc=$?"; tail -5 $OUT/r3g_wd.log
for S in 1 2 1 2; do
  WHMR_BLK_SCHED=$S timeout 300 python bench.py --no-cpu --no-secondary --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sched $S bf16  ms', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],4), 'avg_us', round(d['roofline']['avg_launch_us'],2))"
done
for S in 1 2; do
  WHMR_BLK_SCHED=$S timeout 300 python bench.py --no-cpu --no-secondary --numerics bf16x3 --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sched $S x3    ms', round(d['ms_per_step'],4), 'issue frac', round(d['roofline']['mfma_issue_frac'],4))"
done
