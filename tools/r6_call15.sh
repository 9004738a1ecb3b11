#!/bin/bash
# where do the 2.9 ms of the one-rank data-parallel step go?  plain | buckets without a process group (hooks + pack only) | + RCCL | SyncBatchNorm only
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() { python bench.py --workload whmr_train --no-cpu --no-ceilings --steps 20 --warmup 10 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$TAG', round(d['ms_per_step'],3))"; }
for i in 1 2; do
TAG="plain                         " run
TAG="always-bucket, no group       " run --always-bucket
( export MASTER_ADDR=127.0.0.1 MASTER_PORT=29551 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
  TAG="always-bucket + RCCL (1 rank) " run --always-bucket --batchnorm local
  TAG="sync BN only + RCCL           " run --batchnorm sync
  TAG="always-bucket + sync BN + RCCL" run --always-bucket --batchnorm sync )
done
