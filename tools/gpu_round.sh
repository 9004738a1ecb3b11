#!/bin/bash
# One gpurun call that refreshes the round's evidence under gpurun_out/<tag>_* and writes the judged summaries straight into profiles/<tag>_*
# (each file starts with the command that produced it and the commit it ran on -- recorded HERE, at run time; nothing is stamped afterwards).
#   usage (repo root, on the GPU box):  bash tools/gpu_round.sh <tag> <commit> [tests]
set -uo pipefail
TAG=${1:?tag}
COMMIT=${2:?commit}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out
PROF=$OUT/profiles_$TAG
mkdir -p $OUT $PROF
export TMPDIR=/tmp
cd $R
fail() { echo "gpu_round: $*" >&2; exit 1; }
hdr() { echo "# $1"; echo "# tree: commit $COMMIT; one MI355X gpurun box, $(date -u +%Y-%m-%d); produced by tools/gpu_round.sh $TAG"; }
if [ "${3:-}" = "tests" ]; then
  timeout 2400 python -m pytest tests -m gpu -q > $OUT/${TAG}_gpu_tests.log 2>&1
  echo "gpu tests rc=$?"; tail -3 $OUT/${TAG}_gpu_tests.log
fi
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.log 2>&1 || fail "smoke failed: $(tail -3 $OUT/${TAG}_smoke.log)"
grep "smoke ok" $OUT/${TAG}_smoke.log
bench() {  # name, args...
  local n=$1; shift
  python bench.py "$@" > $PROF/${TAG}_bench_$n.json 2> $OUT/${TAG}_bench_$n.err || fail "bench $n failed: $(tail -2 $OUT/${TAG}_bench_$n.err)"
  [ -s $PROF/${TAG}_bench_$n.json ] || fail "bench $n wrote nothing"
  grep '^{' $PROF/${TAG}_bench_$n.json | tail -1 > $PROF/${TAG}_bench_$n.json.tmp && mv $PROF/${TAG}_bench_$n.json.tmp $PROF/${TAG}_bench_$n.json    # (RCCL prints a banner to stdout)
  [ -s $PROF/${TAG}_bench_$n.json ] || fail "bench $n printed no JSON line"
}
bench vit224
bench vit224_bf16x3 --numerics bf16x3 --steps 20 --warmup 5
bench vit224_fp32 --numerics fp32 --steps 5 --warmup 2 --no-cpu
bench whmr --workload whmr
bench whmr_bf16x3 --workload whmr --numerics bf16x3 --no-cpu --steps 10 --warmup 3
bench whmr_train --workload whmr_train --steps 30 --warmup 30
bench whmr_bf16x3_b1 --workload whmr --numerics bf16x3 --batch 1 --no-cpu --no-parity --steps 50 --warmup 10
bench vit256x192 --workload vit256x192 --no-cpu
bench vitl256x192_b32 --workload vitl256x192 --batch 32 --no-cpu
bench whmr_b1 --workload whmr --batch 1 --no-cpu --no-parity --steps 50 --warmup 10
# one-rank RCCL group in THIS process (env:// rendezvous, no torchrun): the reference's own SyncBatchNorm + DDP wrap, and GradReducer + convert_sync_batchnorm
( export MASTER_ADDR=127.0.0.1 MASTER_PORT=29561 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 HSA_ENABLE_IPC_MODE_LEGACY=0
  bench whmr_train_ddp_rccl1 --workload whmr_train --wrap ddp --no-cpu --steps 20 --warmup 10
  bench whmr_train_rccl1_syncbn --workload whmr_train --batchnorm sync --always-bucket --no-cpu --steps 20 --warmup 10 ) || fail "one-rank RCCL benches failed"
cd /tmp
prof() {  # name, description, bench args...
  local n=$1 d=$2; shift 2
  rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof_$n -o $n -- python3 $R/bench.py --no-ceilings "$@" > $OUT/${TAG}_prof_$n.log 2>&1
  local db=$(find $OUT/${TAG}_prof_$n -name '*.db' | head -1)
  [ -n "$db" ] || fail "no rocprof database for $n"
  { hdr "rocprofv3 --kernel-trace --stats -- python3 bench.py $*   ($d)"; python3 $R/tools/rocprof_summary.py $db | tail -n +2; } > $PROF/${TAG}_${n}_kernel_stats.txt
  rm -rf $OUT/${TAG}_prof_$n
}
prof vit224_b64 "ViT-B/16 224^2 batch 64 bf16: timed steps + warm-ups + 1 instrumented step in the trace" --no-cpu --no-secondary --steps 10 --warmup 3
prof vit224_b64_bf16x3 "the same workload in the bf16x3 numerics" --no-cpu --no-secondary --numerics bf16x3 --steps 10 --warmup 3
prof whmr_b64 "full W-HMR forward, batch 64 + one 600x800 frame, bf16, HIP-graph replays + one eager instrumented step" --workload whmr --no-cpu --no-parity --steps 10 --warmup 3
prof whmr_b64_serial "the same forward with the side streams folded into the main one (eager): every kernel's duration WITHOUT concurrency" --workload whmr --serial --no-cpu --no-parity --steps 10 --warmup 3
prof whmr_train_b64 "W-HMR training step, batch 64, bf16, Adam inside the step" --workload whmr_train --no-cpu --steps 4 --warmup 2
# timeline of one replayed full forward (every launch with start offset / duration / HSA queue)
rocprofv3 --kernel-trace -d $OUT/${TAG}_prof_tl -o tl -- python3 $R/bench.py --no-ceilings --workload whmr --no-cpu --no-parity --steps 10 --warmup 3 > $OUT/${TAG}_prof_tl.log 2>&1
TDB=$(find $OUT/${TAG}_prof_tl -name '*.db' | head -1)
[ -n "$TDB" ] || fail "no rocprof database for the timeline"
{ hdr "rocprofv3 --kernel-trace -- python3 bench.py --workload whmr --no-cpu --no-parity --steps 10 --warmup 3 ; python3 tools/whmr_timeline.py <db>   (the profiler serialises queues and stretches small launches)"; python3 $R/tools/whmr_timeline.py $TDB 0; } > $PROF/${TAG}_whmr_timeline.txt
rm -rf $OUT/${TAG}_prof_tl
pmc() {  # name, counters (quoted), bench args...
  local n=$1 c=$2; shift 2
  rocprofv3 --kernel-trace --pmc $c -d $OUT/${TAG}_pmc_$n -o pmc -- python3 $R/bench.py --no-ceilings "$@" > $OUT/${TAG}_pmc_$n.log 2>&1
  local db=$(find $OUT/${TAG}_pmc_$n -name '*.db' | head -1)
  [ -n "$db" ] || fail "no PMC database for $n"
  echo $db
}
FDB=$(pmc vit224_FETCH FETCH_SIZE --no-cpu --no-secondary --steps 3 --warmup 2)
WDB=$(pmc vit224_WRITE WRITE_SIZE --no-cpu --no-secondary --steps 3 --warmup 2)
SDB=$(pmc vit224_SQ "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" --no-cpu --no-secondary --steps 3 --warmup 2)
XDB=$(pmc vit224_x3_SQ "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" --no-cpu --no-secondary --numerics bf16x3 --steps 3 --warmup 2)
WF=$(pmc whmr_FETCH FETCH_SIZE --workload whmr --eager --no-cpu --no-parity --steps 3 --warmup 2)
WW=$(pmc whmr_WRITE WRITE_SIZE --workload whmr --eager --no-cpu --no-parity --steps 3 --warmup 2)
cd $R
K="gemm_blk|attention|layernorm|patch"
{ hdr "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -- python3 bench.py --no-cpu --no-secondary --steps 3 --warmup 2  (mean per dispatch, summed over the device; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES = 32 x MFMA count)";
  python tools/pmc_summary.py $SDB | grep -E "$K"; echo "# the same with --numerics bf16x3"; python tools/pmc_summary.py $XDB | grep -E "$K";
  echo "# --pmc FETCH_SIZE (own pass, KiB, raw)"; python tools/pmc_summary.py $FDB | grep -E "$K"; echo "# --pmc WRITE_SIZE (own pass, KiB)"; python tools/pmc_summary.py $WDB | grep -E "$K"; } > $PROF/${TAG}_vit224_gemm_pmc.txt
python tools/make_traffic.py $FDB $WDB $PROF/${TAG}_vit224_gemm_traffic.json > /dev/null 2>&1 || fail "make_traffic failed"
K2="maf_sample|smpl_|regressor_|tz_|attention|layernorm_blk|split3"
{ hdr "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --workload whmr --eager --no-cpu --no-parity --steps 3 --warmup 2; mean per dispatch, summed over the device, KiB; HBM-side bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 on gfx950 (MI355X_MICROARCH.md)";
  echo "# FETCH_SIZE"; python tools/pmc_summary.py $WF | grep -E "$K2"; echo "# WRITE_SIZE"; python tools/pmc_summary.py $WW | grep -E "$K2"; } > $PROF/${TAG}_whmr_pmc.txt
rm -rf $OUT/${TAG}_pmc_*
for f in $PROF/*; do [ -s $f ] || fail "empty evidence file $f"; done
python - <<PY
import json, glob, os
for f in sorted(glob.glob('$PROF/${TAG}_bench_*.json')):
    d = json.load(open(f)); print(os.path.basename(f), round(d['ms_per_step'], 3), round(d['value']), round(d['roofline']['frac'], 3), d['roofline'].get('traffic'))
PY
du -sh $OUT | tail -1
