#!/bin/bash
# round 6, call 66: the composed Tz convolution in the training graph -- unit test, training tests, same-box A/B
python -m pytest tests/test_train_gpu.py -x -q -m gpu -k "tz_composed_convolution_node" 2>&1 | grep -v "^$" | tail -15
timeout 1200 python -m pytest tests/test_train_gpu.py -x -q -m gpu 2>&1 | tail -3
for r in 1 2 3; do
  for f in 0 1; do
    WHMR_TRAIN_COMPOSE_TZ=$f python bench.py --workload whmr_train --no-cpu --no-ceilings --steps 30 --warmup 30 2>/dev/null | python -c "import sys,json; [print('compose_tz=$f', json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]"
  done
done
