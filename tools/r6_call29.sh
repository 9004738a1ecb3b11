#!/bin/bash
# round 6, call 29: the stage-3 sampler's deferred map gradient (MapForkFn) -- unit tests, training tests, same-box A/B of the step
mkdir -p gpurun_out
python -m pytest tests/test_train_gpu.py -x -q -m gpu -k "map_fork or maf_sampler or passthrough" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_train_gpu.py -x -q -m gpu 2>&1 | tail -5
for r in 1 2 3; do
  for f in 0 1; do
    WHMR_TRAIN_FORK3=$f python bench.py --workload whmr_train --no-cpu --no-ceilings --steps 30 --warmup 30 2>/dev/null | python -c "import sys,json; [print('fork3=$f', json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]"
  done
done
