#!/bin/bash
# A/B of the blocked bf16 GEMM's MFMA shape on one box, interleaved: v_mfma_f32_16x16x32_bf16 (default, gemm_blk16_impl.h) against the
# 32x32x16 kernel of gemm_blk_impl.h (WHMR_BLK_MFMA=32); same packed operands, results equal to rounding (tests/test_blocked_gpu.py).
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
fmt() { grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 ms', round(d['ms_per_step'],4), 'gemm frac', round(d['roofline']['frac'],4))"; }
new() { python bench.py --no-cpu --no-secondary --steps 30 --warmup 8 $* 2>/dev/null | fmt "16x16x32 $*"; }
old() { WHMR_BLK_MFMA=32 python bench.py --no-cpu --no-secondary --steps 30 --warmup 8 $* 2>/dev/null | fmt "32x32x16 $*"; }
old; new; old; new; old; new
old --workload vitl256x192 --batch 32; new --workload vitl256x192 --batch 32
old --workload whmr; new --workload whmr
