#!/bin/bash
# Round 4: tile height of the N = 768 launches (proj K = 768, fc2 K = 3072) with the residual ring -- two rounds of low tiles (the second round's main loop
# over the first round's store drain) against the chooser's one round of 160-row tiles; interleaved on one box.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
fmt() { grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 ms', round(d['ms_per_step'],4), 'gemm frac', round(d['roofline']['frac'],4))"; }
run() { env $1 python bench.py --no-cpu --no-secondary --steps 30 --warmup 8 2>/dev/null | fmt "$1"; }
old() { python tools/lab/run_with_lib.py tools/lab/libwhmr_hip_r3gemm.so --no-cpu --no-secondary --steps 30 --warmup 8 2>/dev/null | fmt "round-3 kernel"; }
echo "# $(date -u +%FT%TZ)"
old; run X=0
run WHMR_BLK_TILE_PROJ=0x22; run WHMR_BLK_TILE_PROJ=0x21; run WHMR_BLK_TILE_PROJ=0x33
run WHMR_BLK_TILE_FC2=0x22; run WHMR_BLK_TILE_FC2=0x21; run WHMR_BLK_TILE_FC2=0x33
old; run X=0
run "WHMR_BLK_TILE_PROJ=0x22 WHMR_BLK_TILE_FC2=0x22"; run "WHMR_BLK_TILE_PROJ=0x21 WHMR_BLK_TILE_FC2=0x21"
