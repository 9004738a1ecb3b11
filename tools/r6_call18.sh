#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout 1500 python -m pytest tests/test_vit_gpu.py tests/test_hotpath_gpu.py tests/test_blocked_gpu.py tests/test_x3_gpu.py -m gpu -q -x 2>&1 | tail -3
for i in 1 2; do
python bench.py --no-cpu --no-secondary --steps 30 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('vit224 ms', round(d['ms_per_step'],4), round(d['roofline']['frac'],4))"
python bench.py --workload whmr --no-cpu --no-parity --steps 30 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('whmr ms', round(d['ms_per_step'],4))"
done
python tools/forward_census.py 2>&1 | tail -12
