#!/bin/bash
# round 6, call 32: heavy chain first in autograd's ready queue + the Tz tail on its own stream -- training tests, same-box A/B of the step
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_train_gpu.py -x -q -m gpu 2>&1 | tail -5
for r in 1 2 3; do
  for cfg in "0 0" "1 0" "0 1" "1 1"; do
    set -- $cfg
    WHMR_TRAIN_HEAVY_FIRST=$1 WHMR_TRAIN_TZ_TAIL=$2 python bench.py --workload whmr_train --no-cpu --no-ceilings --steps 30 --warmup 30 2>/dev/null | python -c "import sys,json; [print('heavy_first=$1 tz_tail=$2', json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]"
  done
done
