#!/bin/bash
# Extra round-2 evidence beside tools/gpu_round.sh: kernel stats of the ViT-L share, of the fp32 parity mode, MFMA-busy counters of the training
# step's MFMA kernels (TN weight-gradient kernel included).  bash tools/gpu_round_extra.sh <tag>
TAG=${1:-r02x}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof_vitl -o p -- python3 $R/bench.py --workload vitl256x192 --batch 32 --no-cpu --steps 10 --warmup 3 > $OUT/${TAG}_prof_vitl.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof_fp32 -o p -- python3 $R/bench.py --numerics fp32 --no-cpu --steps 4 --warmup 2 > $OUT/${TAG}_prof_fp32.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY -d $OUT/${TAG}_pmc_train -o p -- python3 $R/bench.py --workload whmr_train --no-cpu --steps 2 --warmup 1 > $OUT/${TAG}_pmc_train.log 2>&1
cd $R
python tools/rocprof_summary.py $(find $OUT/${TAG}_prof_vitl -name '*.db' | head -1) > $OUT/${TAG}_vitl_kernel_stats.txt 2>&1
python tools/rocprof_summary.py $(find $OUT/${TAG}_prof_fp32 -name '*.db' | head -1) > $OUT/${TAG}_fp32_kernel_stats.txt 2>&1
python tools/pmc_summary.py $(find $OUT/${TAG}_pmc_train -name '*.db' | head -1) > $OUT/${TAG}_train_pmc.txt 2>&1
rm -rf $OUT/${TAG}_prof_vitl $OUT/${TAG}_prof_fp32 $OUT/${TAG}_pmc_train
ls -la $OUT | grep ${TAG} | head
