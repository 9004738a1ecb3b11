#!/bin/bash
# Round 6, first GPU call: composed Tz convolution probe (three numerics), deconv tile sweep, the new unit tests, full-forward A/B.
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
{ for m in bf16 bf16x3 fp32; do timeout 300 python tools/r6_tz_probe.py $m; done; } > $OUT/r6_tz_probe.txt 2>&1
tail -45 $OUT/r6_tz_probe.txt
timeout 600 python tools/deconv_probe.py > $OUT/r6_deconv_probe.txt 2>&1; tail -25 $OUT/r6_deconv_probe.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "tz_" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_hotpath_gpu.py -m gpu -q -x 2>&1 | tail -5
for i in 1 2; do
  WHMR_COMPOSE_TZ=0 python bench.py --workload whmr --no-cpu --no-parity --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('two-conv  whmr bf16 ms', d['ms_per_step'])"
  WHMR_COMPOSE_TZ=1 python bench.py --workload whmr --no-cpu --no-parity --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('composed  whmr bf16 ms', d['ms_per_step'])"
done
WHMR_COMPOSE_TZ=0 python bench.py --workload whmr --numerics bf16x3 --no-cpu --no-parity --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('two-conv  whmr x3 ms', d['ms_per_step'])"
WHMR_COMPOSE_TZ=1 python bench.py --workload whmr --numerics bf16x3 --no-cpu --no-parity --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('composed  whmr x3 ms', d['ms_per_step'])"
