#!/bin/bash
# torch DDP wrap (one-rank RCCL group): which of today's switches costs it time
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29561 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 HSA_ENABLE_IPC_MODE_LEGACY=0
run() { env "$@" python bench.py --workload whmr_train --wrap ddp --no-cpu --no-ceilings --steps 20 --warmup 10 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; print('$*', round(json.loads(sys.stdin.read())['ms_per_step'],3))"; }
for r in 1 2; do
run A=0
run WHMR_TRAIN_HEAVY_FIRST=0
run WHMR_TRAIN_TZ_TAIL=0
run WHMR_TRAIN_HEAVY_FIRST=0 WHMR_TRAIN_TZ_TAIL=0 WHMR_TRAIN_FORK3=0 WHMR_TRAIN_GROUP_DX=0
done
