#!/bin/bash
# One-rank RCCL run of the training step WITH SyncBatchNorm (all a 1-GPU box allows): the packed fp64 all-reduces of the four BatchNorm layers
# (16 per step) and the gradient buckets go through RCCL on the device.   usage: bash tools/r5_rccl1_syncbn.sh -> gpurun_out/r05_bench_whmr_train_rccl1_syncbn.json
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --workload whmr_train --batchnorm sync --always-bucket --no-cpu --steps 20 --warmup 10 > $OUT/r05_bench_whmr_train_rccl1_syncbn.json 2> $OUT/r05_bench_whmr_train_rccl1_syncbn.err
echo "rc=$?"; tail -3 $OUT/r05_bench_whmr_train_rccl1_syncbn.err; python -c "
import json; d=json.loads([l for l in open('$OUT/r05_bench_whmr_train_rccl1_syncbn.json') if l.startswith('{')][-1]); print(d['n_gpus'], round(d['ms_per_step'],3), d['config']['parallelism']); print(json.dumps(d.get('multi_gpu'))[:1800])"
