#!/bin/bash
# One gpurun call that refreshes the round's evidence: GPU tests, the bench lines of every workload, rocprofv3 kernel stats and the
# PMC traffic passes of the headline command.  Usage (from the repo root on the GPU box):  bash tools/gpu_round.sh <tag> [tests]
# Outputs land under gpurun_out/<tag>_*; copy the summaries into profiles/ afterwards (tools/rocprof_summary.py / make_traffic.py).
TAG=${1:-r02}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
if [ "$2" = "tests" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_gpu_tests.log 2>&1
  tail -3 $OUT/${TAG}_gpu_tests.log
fi
python bench.py > $OUT/${TAG}_bench_vit224.json 2> $OUT/${TAG}_bench_vit224.err
python bench.py --workload whmr > $OUT/${TAG}_bench_whmr.json 2> $OUT/${TAG}_bench_whmr.err
python bench.py --workload whmr_train --steps 10 --warmup 3 > $OUT/${TAG}_bench_whmr_train.json 2> $OUT/${TAG}_bench_whmr_train.err
python bench.py --workload vit256x192 --no-cpu > $OUT/${TAG}_bench_vit256x192.json 2>/dev/null
python bench.py --workload vitl256x192 --batch 32 --no-cpu > $OUT/${TAG}_bench_vitl256x192_b32.json 2>/dev/null
cat $OUT/${TAG}_bench_vit224.json $OUT/${TAG}_bench_whmr.json $OUT/${TAG}_bench_whmr_train.json
cd /tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof_vit224 -o vit224 -- python3 $R/bench.py --no-cpu --steps 10 --warmup 3 > $OUT/${TAG}_prof_vit224.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof_whmr -o whmr -- python3 $R/bench.py --workload whmr --no-cpu --no-parity --steps 10 --warmup 3 > $OUT/${TAG}_prof_whmr.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof_whmr_train -o whmr_train -- python3 $R/bench.py --workload whmr_train --no-cpu --steps 4 --warmup 2 > $OUT/${TAG}_prof_whmr_train.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $OUT/${TAG}_pmc_vit224_$C -o pmc -- python3 $R/bench.py --no-cpu --steps 3 --warmup 2 > $OUT/${TAG}_pmc_vit224_$C.log 2>&1
  rocprofv3 --kernel-trace --pmc $C -d $OUT/${TAG}_pmc_whmr_$C -o pmc -- python3 $R/bench.py --workload whmr --eager --no-cpu --no-parity --steps 3 --warmup 2 > $OUT/${TAG}_pmc_whmr_$C.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/${TAG}_pmc_vit224_SQ -o pmc -- python3 $R/bench.py --no-cpu --steps 3 --warmup 2 > $OUT/${TAG}_pmc_vit224_SQ.log 2>&1
cd $R
# summarise on the box and drop the sqlite databases (gpurun merges back at most 64 MiB)
for W in vit224 whmr whmr_train; do
  python tools/rocprof_summary.py $(find $OUT/${TAG}_prof_$W -name '*.db' | head -1) > $OUT/${TAG}_${W}_kernel_stats.txt 2>&1
done
for W in vit224 whmr; do
  for C in FETCH_SIZE WRITE_SIZE; do
    python tools/pmc_summary.py $(find $OUT/${TAG}_pmc_${W}_$C -name '*.db' | head -1) > $OUT/${TAG}_${W}_pmc_$C.txt 2>&1
  done
done
python tools/pmc_summary.py $(find $OUT/${TAG}_pmc_vit224_SQ -name '*.db' | head -1) > $OUT/${TAG}_vit224_pmc_SQ.txt 2>&1
python tools/make_traffic.py $(find $OUT/${TAG}_pmc_vit224_FETCH_SIZE -name '*.db' | head -1) $(find $OUT/${TAG}_pmc_vit224_WRITE_SIZE -name '*.db' | head -1) $OUT/${TAG}_vit224_gemm_traffic.json > /dev/null 2>&1
rm -rf $OUT/${TAG}_prof_vit224 $OUT/${TAG}_prof_whmr $OUT/${TAG}_prof_whmr_train $OUT/${TAG}_pmc_*_FETCH_SIZE $OUT/${TAG}_pmc_*_WRITE_SIZE $OUT/${TAG}_pmc_vit224_SQ
du -sh $OUT | tail -1
