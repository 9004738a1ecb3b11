#!/bin/bash
# Round 5, VERDICT r4 item 1: (a) GEMM-pattern LDS-DMA probe with same-XCD / cross-XCD / unshared operand panels (tools/lab/dma_rate x),
# (b) L2 hit / miss counters of the real bf16 GEMM launches, (c) the stamped time budget of those launches (tools/gemm_stamps.py on the lab build
# tools/lab/libwhmr_hip_stamps.so = the tree's objects + gemm_blk.hip compiled with -DWHMR_BLK_STAMPS).
#   usage (repo root, on the GPU box):  bash tools/r5_gemm.sh <commit>
set -uo pipefail
COMMIT=${1:?commit}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out
PROF=$OUT/profiles_r05
mkdir -p $OUT $PROF
export TMPDIR=/tmp
cd $R
hdr() { echo "# $1"; echo "# tree: commit $COMMIT; one MI355X gpurun box, $(date -u +%Y-%m-%d); produced by tools/r5_gemm.sh"; }
{ hdr "tools/lab/dma_rate 256 4096 x   (GEMM-pattern operand streams: 16 KiB of an A panel + 16 KiB of a W panel per half tile and workgroup, 96 KiB in flight, one barrier per half tile)";
  timeout 300 tools/lab/dma_rate 256 4096 x; } > $PROF/r05_dma_rate_gemm_pattern.txt 2>&1
tail -5 $PROF/r05_dma_rate_gemm_pattern.txt
{ hdr "python tools/gemm_stamps.py   (lab build with WHMR_BLK_STAMPS)"; timeout 600 python tools/gemm_stamps.py; } > $PROF/r05_gemm_stamps.txt 2> $OUT/r05_gemm_stamps.err
tail -30 $PROF/r05_gemm_stamps.txt; tail -3 $OUT/r05_gemm_stamps.err
cd /tmp
rocprofv3 -L > $OUT/r05_counters_list.txt 2>&1
grep -o -E "TCC_(HIT|MISS|REQ|READ|EA0_RDREQ|EA0_RDREQ_32B|TAG_STALL|BUBBLE)[A-Za-z0-9_]*|TCP_TCC_READ_REQ[A-Za-z0-9_]*|TCP_TCC_[A-Z_]*REQ[A-Za-z0-9_]*" $OUT/r05_counters_list.txt | sort -u | head -40
pmc() {  # name, counters
  local n=$1 c=$2
  rocprofv3 --kernel-trace --pmc $c -d $OUT/r05_pmc_$n -o pmc -- python3 $R/bench.py --no-cpu --no-secondary --steps 3 --warmup 2 > $OUT/r05_pmc_$n.log 2>&1
  find $OUT/r05_pmc_$n -name '*.db' | head -1
}
L2DB=$(pmc l2 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum")
RDDB=$(pmc rd "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum")
cd $R
{ hdr "rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum / --pmc TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum (separate passes) -- python3 bench.py --no-cpu --no-secondary --steps 3 --warmup 2; mean per dispatch";
  [ -n "$L2DB" ] && python tools/pmc_summary.py $L2DB | grep -E "gemm_blk|attention"; [ -n "$RDDB" ] && python tools/pmc_summary.py $RDDB | grep -E "gemm_blk|attention"; } > $PROF/r05_vit224_gemm_l2_pmc.txt
cat $PROF/r05_vit224_gemm_l2_pmc.txt | head -40
rm -rf $OUT/r05_pmc_*
