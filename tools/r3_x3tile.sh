#!/bin/bash
# A/B of forced tile heights for the bf16x3 GEMMs (WHMR_BLK_TILE_*): the chooser picks hold
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
run() { python bench.py --no-cpu --no-secondary --numerics bf16x3 --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 ms', round(d['ms_per_step'],4), 'issue', round(d['roofline']['mfma_issue_frac'],4))"; }
run base
WHMR_BLK_TILE_QKV=0x43 run qkv224
WHMR_BLK_TILE_QKV=0x33 run qkv192
WHMR_BLK_TILE_FC1=0x44 run fc1_256
WHMR_BLK_TILE_FC1=0x43 run fc1_224
WHMR_BLK_TILE_FC2=0x33 WHMR_BLK_TILE_PROJ=0x33 run n768_192
WHMR_BLK_TILE_FC2=0x22 WHMR_BLK_TILE_PROJ=0x22 run n768_128
run base
