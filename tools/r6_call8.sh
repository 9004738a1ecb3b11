#!/bin/bash
# where do the 2.8 ms of the one-rank reducer + SyncBatchNorm step go (22.9 vs 20.1 ms)?  kernel traces of both, per-kernel difference per step
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29543 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
rocprofv3 --kernel-trace -d $OUT/r6_prof_plain -o p -- python3 $R/bench.py --workload whmr_train --no-cpu --steps 6 --warmup 3 > $OUT/r6_prof_plain.log 2>&1
rocprofv3 --kernel-trace -d $OUT/r6_prof_red -o p -- python3 $R/bench.py --workload whmr_train --batchnorm sync --always-bucket --no-cpu --steps 6 --warmup 3 > $OUT/r6_prof_red.log 2>&1
rocprofv3 --kernel-trace -d $OUT/r6_prof_bucket -o p -- python3 $R/bench.py --workload whmr_train --always-bucket --batchnorm local --no-cpu --steps 6 --warmup 3 > $OUT/r6_prof_bucket.log 2>&1
A=$(find $OUT/r6_prof_plain -name '*.db' | head -1); B=$(find $OUT/r6_prof_red -name '*.db' | head -1); C=$(find $OUT/r6_prof_bucket -name '*.db' | head -1)
echo "== plain (A) vs reducer + sync BN (B)"; python3 $R/tools/rocprof_diff.py $A $B 13 13 | cut -c1-150
echo "== plain (A) vs reducer only (B)"; python3 $R/tools/rocprof_diff.py $A $C 13 13 | cut -c1-150 | head -16
for f in plain red bucket; do grep -h '"ms_per_step"' $OUT/r6_prof_$f.log | python3 -c "import sys,json; [print('$f ms under the profiler', round(json.loads(l)['ms_per_step'],2)) for l in sys.stdin if l.startswith('{')]"; done
rm -rf $OUT/r6_prof_plain $OUT/r6_prof_red $OUT/r6_prof_bucket
