#!/bin/bash
# One-rank RCCL smoke on a 1-GPU box (all a gpurun box allows): bench.py under torch.distributed.run with --nproc-per-node 1 -- init_process_group('nccl'),
# the tensor / object all_gathers of the round-4 `multi_gpu` block (rank -> device map, per-rank times, gradient-exchange bytes / buckets / exposed
# wait), barrier, all_reduce, and with --always-bucket the GradReducer's bucket exchange on its side stream + broadcast_buffers.
#   usage: bash tools/r4_rccl1.sh  -> gpurun_out/r04_bench_whmr_train_rccl1.json
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --workload whmr_train --always-bucket --no-cpu --steps 5 --warmup 2 > $OUT/r04_bench_whmr_train_rccl1.json 2> $OUT/r04_bench_whmr_train_rccl1.err
echo "rc=$?"; tail -3 $OUT/r04_bench_whmr_train_rccl1.err; python -c "
import json; d=json.loads([l for l in open('$OUT/r04_bench_whmr_train_rccl1.json') if l.startswith('{')][-1]); print(d['n_gpus'], round(d['ms_per_step'],3), json.dumps(d.get('multi_gpu'))[:1500])"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --always-bucket --no-cpu --no-secondary --steps 10 --warmup 3 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('vit224 under a 1-rank RCCL group: n_gpus', d['n_gpus'], 'ms', round(d['ms_per_step'],3), json.dumps(d.get('multi_gpu'))[:600])"
for a in "" "--graph"; do python bench.py --workload whmr_train --no-cpu --steps 10 --warmup 3 $a 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('whmr_train $a: ms', round(d['ms_per_step'],3))"; done
