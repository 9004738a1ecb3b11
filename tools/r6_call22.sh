#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
COMMIT=$1
rocprofv3 --kernel-trace --stats -d $OUT/r6_prof_x3 -o x3 -- python3 $R/bench.py --no-ceilings --workload whmr --numerics bf16x3 --no-cpu --no-parity --steps 10 --warmup 3 > $OUT/r6_prof_x3.log 2>&1
DB=$(find $OUT/r6_prof_x3 -name '*.db' | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-ceilings --workload whmr --numerics bf16x3 --no-cpu --no-parity --steps 10 --warmup 3   (full W-HMR forward in the DEFAULT numerics, batch 64 + one 600x800 frame)"; echo "# tree: commit $COMMIT; one MI355X gpurun box, $(date -u +%Y-%m-%d)"; python3 $R/tools/rocprof_summary.py $DB | tail -n +2; } > $OUT/r06_whmr_b64_bf16x3_kernel_stats.txt
rm -rf $OUT/r6_prof_x3
head -24 $OUT/r06_whmr_b64_bf16x3_kernel_stats.txt | cut -c1-170
