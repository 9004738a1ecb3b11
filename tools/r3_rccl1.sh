#!/bin/bash
# One-rank RCCL smoke on a 1-GPU box: bench.py under torch.distributed.run with --nproc-per-node 1 -- init_process_group('nccl'), barrier,
# all_reduce (rank count / max time), and with --always-bucket the GradReducer's bucket exchange on its side stream + broadcast_buffers.
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --workload whmr_train --always-bucket --no-cpu --steps 5 --warmup 2 > $OUT/r03_bench_whmr_train_rccl1.json 2> $OUT/r03_bench_whmr_train_rccl1.err
echo "rc=$?"; tail -3 $OUT/r03_bench_whmr_train_rccl1.err; python -c "
import json; d=json.loads([l for l in open('$OUT/r03_bench_whmr_train_rccl1.json') if l.startswith('{')][-1]); print(d['n_gpus'], d['ms_per_step'], d['config']['parallelism'][:60])"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --always-bucket --no-cpu --no-secondary --steps 10 --warmup 3 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('vit224 under a 1-rank RCCL group: n_gpus', d['n_gpus'], 'ms', d['ms_per_step'])"
python bench.py --workload whmr_train --no-cpu --steps 5 --warmup 2 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('same box, no process group: ms', d['ms_per_step'])"
