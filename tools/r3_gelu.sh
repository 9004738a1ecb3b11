#!/bin/bash
# bf16x3 fc1 epilogue: A&S erf GELU -- parity tests of the x3 path + step time
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_x3_gpu.py -m gpu -q -x -s > $OUT/r3u.log 2>&1; echo "rc=$?"; grep -E "passed|failed|assert|bf16x3 ViT" $OUT/r3u.log | tail -8
timeout 600 python -m pytest tests/test_hotpath_gpu.py -m gpu -q -x -k "bf16x3" 2>&1 | tail -1
for i in 1 2; do python bench.py --no-cpu --no-secondary --numerics bf16x3 --steps 30 --warmup 5 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('x3 ms', round(d['ms_per_step'],4), 'issue', round(d['roofline']['mfma_issue_frac'],4))"; done
python bench.py --no-cpu --no-secondary --steps 30 --warmup 5 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16 ms', round(d['ms_per_step'],4))"
