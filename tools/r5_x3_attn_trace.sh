#!/bin/bash
# Round 5: why is the bf16x3 ViT at 224^2 not faster with the (isolated: 17 us faster) persistent attention kernel?  Kernel traces of the same bench command
# with the old and the new attention on ONE box -> per-kernel averages side by side.   usage: bash tools/r5_x3_attn_trace.sh
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for v in 1 0 1 0; do
  export WHMR_ATTN_OLD=$v
  rocprofv3 --kernel-trace --stats -d $OUT/r05_x3t_$v -o t -- python3 $R/bench.py --no-cpu --no-secondary --numerics bf16x3 --steps 10 --warmup 3 > $OUT/r05_x3t_$v.log 2>&1
  db=$(find $OUT/r05_x3t_$v -name '*.db' | head -1)
  echo "== WHMR_ATTN_OLD=$v"; tail -1 $OUT/r05_x3t_$v.log | python3 -c "import json,sys; print('ms_per_step', json.loads(sys.stdin.read())['ms_per_step'])"
  python3 $R/tools/rocprof_summary.py $db | head -7 | cut -c1-150
  rm -rf $OUT/r05_x3t_$v
done
