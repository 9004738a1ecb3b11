#!/bin/bash
# round 6, call 31: timeline of one training step (every launch, queue = stream)
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/r6_prof_tltrain -o tl -- python3 $R/bench.py --no-ceilings --workload whmr_train --no-cpu --steps 6 --warmup 3 > $OUT/r6_prof_tltrain.log 2>&1
DB=$(find $OUT/r6_prof_tltrain -name '*.db' | head -1)
python3 $R/tools/whmr_timeline.py $DB 0 > $OUT/r6_train_timeline.txt 2>&1
rm -rf $OUT/r6_prof_tltrain
head -3 $OUT/r6_train_timeline.txt; tail -5 $OUT/r6_train_timeline.txt
