#!/bin/bash
# Round 4: the blocked forward GEMM's 256 x 256 tile on four waves, one per SIMD (tile id 0x144) against the eight-wave ping-pong kernel (0x44) for the
# two launches that use it (qkv N = 2304, fc1 N = 3072); headline bench, interleaved on one box; the ViT parity tests run with the new tile forced.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
fmt() { grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 ms', round(d['ms_per_step'],4), 'gemm frac', round(d['roofline']['frac'],4), 'avg launch us', round(d['roofline'].get('avg_launch_us', 0),2))"; }
run() { env $1 python bench.py --no-cpu --no-secondary --steps 30 --warmup 8 2>/dev/null | fmt "$1"; }
echo "# $(date -u +%FT%TZ)"
WHMR_BLK_TILE_QKV=0x144 WHMR_BLK_TILE_FC1=0x144 python -m pytest tests/test_vit_gpu.py -x -q 2>&1 | tail -3
for i in 1 2 3; do run X=0; run "WHMR_BLK_TILE_QKV=0x144"; run "WHMR_BLK_TILE_FC1=0x144"; run "WHMR_BLK_TILE_QKV=0x144 WHMR_BLK_TILE_FC1=0x144"; done
