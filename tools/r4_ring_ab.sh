#!/bin/bash
# Round 4: residual ring of the blocked GEMM's fp32 epilogue (gemm_blk16_impl.h) against the round-3 kernel, interleaved on one box.
#   tools/lab/libwhmr_hip_r3gemm.so = this tree's objects with gemm_blk.o built from the round-3 sources (git archive 7a2c1ef w-hmr_amd/csrc)
# Usage on the GPU box: bash tools/r4_ring_ab.sh > gpurun_out/r4_ring_ab.txt
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
fmt() { grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 ms', round(d['ms_per_step'],4), 'gemm frac', round(d['roofline']['frac'],4))"; }
new() { WHMR_BLK_RES_LEAD=$1 python bench.py --no-cpu --no-secondary --steps 30 --warmup 8 ${@:2} 2>/dev/null | fmt "ring lead=$1 ${*:2}"; }
old() { python tools/lab/run_with_lib.py tools/lab/libwhmr_hip_r3gemm.so --no-cpu --no-secondary --steps 30 --warmup 8 $* 2>/dev/null | fmt "round-3 kernel $*"; }
echo "# $(git rev-parse --short HEAD 2>/dev/null) $(date -u +%FT%TZ)"
old; new 6; old; new 6; new 2; new 10; new 14; new 0; old; new 6
old --workload vit256x192; new 6 --workload vit256x192
old --workload vitl256x192 --batch 32; new 6 --workload vitl256x192 --batch 32
old --workload whmr --no-parity; new 6 --workload whmr --no-parity
