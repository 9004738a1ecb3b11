#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/r6_prof_tlx3 -o tl -- python3 $R/bench.py --no-ceilings --workload whmr --numerics bf16x3 --no-cpu --no-parity --steps 8 --warmup 3 > $OUT/r6_prof_tlx3.log 2>&1
DB=$(find $OUT/r6_prof_tlx3 -name '*.db' | head -1)
python3 $R/tools/whmr_timeline.py $DB 25 > $OUT/r6_whmr_timeline_x3.txt 2>&1
rm -rf $OUT/r6_prof_tlx3
head -3 $OUT/r6_whmr_timeline_x3.txt; grep -n "128, 64, 64, 2, 1, 2, 2, 0, 0, 0, true" $OUT/r6_whmr_timeline_x3.txt | head; tail -5 $OUT/r6_whmr_timeline_x3.txt
