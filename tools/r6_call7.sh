#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for i in 1 2; do
for mode in ddp reducer plain; do
  if [ $mode = ddp ]; then EXTRA="--wrap ddp"; elif [ $mode = reducer ]; then EXTRA="--batchnorm sync --always-bucket"; else EXTRA=""; fi
  python bench.py --workload whmr_train $EXTRA --no-cpu --steps 15 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$mode single-process rccl1 ms', round(d['ms_per_step'],3))"
done; done
unset MASTER_ADDR MASTER_PORT RANK WORLD_SIZE LOCAL_RANK
timeout 900 python -m pytest tests/test_train_gpu.py -m gpu -q -x -s -k "ddp_wrap" 2>&1 | grep -v "^\[W\|^$" | tail -4
