# rocprofv3 kernel stats of the training step, selected kernels (quick per-kernel check of a training-kernel change): bash tools/lab/tnprof.sh [grep pattern]
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/tnprof -o t -- python3 $R/bench.py --workload whmr_train --no-cpu --steps 4 --warmup 2 > $OUT/tnprof.log 2>&1
db=$(find $OUT/tnprof -name '*.db' | head -1)
python3 $R/tools/rocprof_summary.py $db > $OUT/tnprof_stats.txt
rm -rf $OUT/tnprof
grep -i "${1:-gemm_tn\|tn_reduce}" $OUT/tnprof_stats.txt | cut -c1-200
