#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 180 python -m pytest tests/test_hotpath_gpu.py -m gpu -q -x -s -k "one_launch" > $OUT/r3e_smpl.log 2>&1
echo "smpl rc=$?"; tail -15 $OUT/r3e_smpl.log
timeout 900 python -m pytest tests/test_hotpath_gpu.py tests/test_blocked_gpu.py -m gpu -q -x -s > $OUT/r3e_hot.log 2>&1
echo "hot rc=$?"; grep -E "passed|failed|Error|error|assert|stress|offset stream" $OUT/r3e_hot.log | tail -20
timeout 1200 python -m pytest tests/test_train_gpu.py tests/test_raster_gpu.py -m gpu -q -x -s > $OUT/r3e_train.log 2>&1
echo "train rc=$?"; grep -E "passed|failed|Error|error|assert|B=64 train" $OUT/r3e_train.log | tail -30
timeout 300 python bench.py --workload whmr --no-cpu --no-parity --steps 10 --warmup 3 > $OUT/r3e_bench_whmr.json 2> $OUT/r3e_bench_whmr.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r3e_bench_whmr.json'))
print(d['ms_per_step'], d['hbm_rows'])
PY
