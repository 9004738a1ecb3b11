#!/bin/bash
# Round-3 GPU call: the whole GPU suite, then the default bench line (headline + parity + secondary legs) and the bf16x3 full forward.
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out
TAG=${1:-r3b}
mkdir -p $OUT
cd $R
timeout 2400 python -m pytest tests -m gpu -q -x -s > $OUT/${TAG}_tests.log 2>&1
echo "tests rc=$?"; grep -E "passed|failed|Error|error|assert" $OUT/${TAG}_tests.log | tail -15
python bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err; echo "bench rc=$?"; cat $OUT/${TAG}_bench_default.json; tail -5 $OUT/${TAG}_bench_default.err
python bench.py --workload whmr --numerics bf16x3 --eager --no-cpu --steps 5 --warmup 2 > $OUT/${TAG}_bench_whmr_x3.json 2> $OUT/${TAG}_bench_whmr_x3.err; cat $OUT/${TAG}_bench_whmr_x3.json; tail -3 $OUT/${TAG}_bench_whmr_x3.err
