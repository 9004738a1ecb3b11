#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_hotpath_gpu.py tests/test_kernels_gpu.py -m gpu -q -x -k "maf or sampler or whmr_forward_bf16 or side_stream" > $OUT/r3n.log 2>&1
echo "rc=$?"; tail -3 $OUT/r3n.log
python bench.py --workload whmr --no-cpu --no-parity --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('whmr ms', round(d['ms_per_step'],4), {k: (round(v['avg_us'],2), round(v['frac_of_8TBps'],4)) for k,v in d['hbm_rows'].items()})"
