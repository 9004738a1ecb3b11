#!/bin/bash
# SMPL call: bit-identity test of the launch forms, whmr bench (hbm_rows), per-batch timing + phase stamps (tools/smpl_timing.py)
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 180 python -m pytest tests/test_hotpath_gpu.py -m gpu -q -x -s -k "one_launch or smpl" > $OUT/r3f_smpl.log 2>&1
echo "smpl rc=$?"; tail -5 $OUT/r3f_smpl.log
timeout 300 python bench.py --workload whmr --no-cpu --no-parity --steps 10 --warmup 3 > $OUT/r3f_bench_whmr.json 2> $OUT/r3f_bench_whmr.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r3f_bench_whmr.json'))
print(d['ms_per_step'], d['hbm_rows'])
PY
timeout 300 python tools/smpl_timing.py
