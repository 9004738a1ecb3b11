#!/bin/bash
# Round-3 first GPU call: the new bf16x3 kernels + the parity-gap tests, then the three numerics of the headline workload.
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_x3_gpu.py tests/test_blocked_gpu.py tests/test_vit_gpu.py -m gpu -q -x -s > $OUT/r3a_tests.log 2>&1
echo "tests rc=$?"; grep -E "max-rel|passed|failed|Error|error" $OUT/r3a_tests.log | tail -40
python bench.py --no-cpu > $OUT/r3a_bench_vit224.json 2> $OUT/r3a_bench_vit224.err; cat $OUT/r3a_bench_vit224.json
python bench.py --no-cpu --numerics bf16x3 --steps 10 --warmup 3 > $OUT/r3a_bench_vit224_x3.json 2> $OUT/r3a_bench_vit224_x3.err; cat $OUT/r3a_bench_vit224_x3.json; tail -3 $OUT/r3a_bench_vit224_x3.err
