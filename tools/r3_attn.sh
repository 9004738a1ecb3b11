#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_blocked_gpu.py tests/test_kernels_gpu.py tests/test_x3_gpu.py -m gpu -q -x -k "attention" > $OUT/r3p.log 2>&1
echo "rc=$?"; tail -3 $OUT/r3p.log
for i in 1 2; do python bench.py --no-cpu --no-secondary --steps 40 --warmup 10 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16 ms', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],4))"; done
python bench.py --no-cpu --no-secondary --numerics bf16x3 --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('x3 ms', round(d['ms_per_step'],4))"
python tools/attn_probe.py 2>/dev/null | tail -8
