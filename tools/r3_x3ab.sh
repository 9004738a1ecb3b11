#!/bin/bash
# A/B of the LayerNorm fold in the bf16x3 pipeline (WHMR_X3_FOLD=1/0)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for F in 1 0 1 0; do
  WHMR_X3_FOLD=$F python bench.py --no-cpu --no-secondary --numerics bf16x3 --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('x3 fold=$F ms', round(d['ms_per_step'],4), 'issue', round(d['roofline']['mfma_issue_frac'],4), 'avg gemm us', round(d['roofline']['avg_launch_us'],2))"
done
