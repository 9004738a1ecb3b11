#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python tools/r6_chain_overlap_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/r6_chain_overlap_probe.txt
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 1 --workload whmr_train --wrap ddp --no-cpu --steps 20 --warmup 10 > $OUT/r6_bench_whmr_train_ddp_rccl1.json 2> $OUT/r6_bench_whmr_train_ddp_rccl1.err
echo "ddp bench rc=$?"; tail -2 $OUT/r6_bench_whmr_train_ddp_rccl1.err
python -c "
import json; d=json.loads([l for l in open('$OUT/r6_bench_whmr_train_ddp_rccl1.json') if l.startswith('{')][-1]); print(d['n_gpus'], round(d['ms_per_step'],3), d['config']['parallelism'][:120]); print(json.dumps(d.get('multi_gpu'))[:1200])"
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29522 bench.py --gpus 1 --workload whmr_train --batchnorm sync --always-bucket --no-cpu --steps 20 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('reducer + sync bn rccl1 ms', round(d['ms_per_step'],3))"
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
