#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
COMMIT=$1
rocprofv3 --kernel-trace --stats -d $OUT/r6_prof_ser -o ser -- python3 $R/bench.py --no-ceilings --workload whmr --serial --no-cpu --no-parity --steps 10 --warmup 3 > $OUT/r6_prof_ser.log 2>&1
DB=$(find $OUT/r6_prof_ser -name '*.db' | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-ceilings --workload whmr --serial --no-cpu --no-parity --steps 10 --warmup 3   (the full forward with the side streams folded into the main one, eager: every kernel's duration WITHOUT concurrency -- deconv3 = the 256x256 gather kernel's max)"; echo "# tree: commit $COMMIT; one MI355X gpurun box, $(date -u +%Y-%m-%d); produced by tools/r6_call23.sh (tools/gpu_round.sh now does the same)"; python3 $R/tools/rocprof_summary.py $DB | tail -n +2; } > $OUT/r06_whmr_b64_serial_kernel_stats.txt
rm -rf $OUT/r6_prof_ser
grep '"ms_per_step"' $OUT/r6_prof_ser.log | python3 -c "import sys,json; [print('serial eager ms under the profiler', round(json.loads(l)['ms_per_step'],3)) for l in sys.stdin if l.startswith('{')]"
grep -n "256, 256, 64, 2, 4, 2, 2, 0, 1, 2, true\|128, 128, 64, 2, 2, 2, 2, 0, 0, 0, true\|tz_fold\|192, 256, 64, 2, 4, 2, 2, 0, 1, 2, true" $OUT/r06_whmr_b64_serial_kernel_stats.txt | cut -c1-175
