#!/bin/bash
# kernel sequence (start order, with gaps) of one HIP-graph replay of the full W-HMR forward: bash tools/seq_round.sh <batch> <n_kernels>
B=${1:-64}; N=${2:-150}
OUT=$PWD/gpurun_out; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/seq_$B -o s -- python3 $R/bench.py --workload whmr --full-x none --batch $B --no-cpu --no-parity --steps 3 --warmup 2 > /tmp/seq_$B.log 2>&1
cd $R
# bench runs one instrumented EAGER step last; the graph replays come before it: print a window that ends before the eager step
python tools/rocprof_seq.py $(find /tmp/seq_$B -name '*.db' | head -1) $((3 * N)) > $OUT/seq_b$B.txt
tail -3 /tmp/seq_$B.log
wc -l $OUT/seq_b$B.txt
