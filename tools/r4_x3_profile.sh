#!/bin/bash
# Round 4: the parity-grade (bf16x3, now the DEFAULT numerics) full forward under rocprofv3 -> profiles/r04_whmr_b64_bf16x3_kernel_stats.txt
#   usage on the GPU box: bash tools/r4_x3_profile.sh <commit>
set -uo pipefail
COMMIT=${1:-unknown}; R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for n in whmr vit224; do
  if [ $n = whmr ]; then A="--workload whmr --numerics bf16x3 --no-cpu --no-parity --steps 10 --warmup 3"; else A="--numerics bf16x3 --no-cpu --no-secondary --steps 10 --warmup 3"; fi
  rocprofv3 --kernel-trace --stats -d $OUT/x3prof_$n -o $n -- python3 $R/bench.py $A > $OUT/x3prof_$n.log 2>&1
  db=$(find $OUT/x3prof_$n -name '*.db' | head -1)
  { echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py $A"; echo "# tree: commit $COMMIT; one MI355X gpurun box, $(date -u +%Y-%m-%d); produced by tools/r4_x3_profile.sh"; python3 $R/tools/rocprof_summary.py $db | tail -n +2; } > $OUT/r04_${n}_b64_bf16x3_kernel_stats.txt
  rm -rf $OUT/x3prof_$n
  grep '^{' $OUT/x3prof_$n.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n bf16x3 ms', round(d['ms_per_step'],3))"
done
