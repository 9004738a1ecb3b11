#!/bin/bash
# A/B of forced tile heights for the bf16 GEMMs on the 16x16x32 kernel (WHMR_BLK_TILE_*): do the chooser's picks still hold?
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
run() { python bench.py --no-cpu --no-secondary --steps 30 --warmup 8 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 ms', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],4))"; }
run base
WHMR_BLK_TILE_QKV=0x43 run qkv224
WHMR_BLK_TILE_QKV=0x55 run qkv320
WHMR_BLK_TILE_QKV=0x33 run qkv192
WHMR_BLK_TILE_FC1=0x44 run fc1_256
WHMR_BLK_TILE_FC1=0x54 run fc1_288
run base
WHMR_BLK_TILE_FC2=0x33 WHMR_BLK_TILE_PROJ=0x33 run n768_192
WHMR_BLK_TILE_FC2=0x22 WHMR_BLK_TILE_PROJ=0x22 run n768_128
WHMR_BLK_TILE_FC2=0x44 WHMR_BLK_TILE_PROJ=0x44 run n768_256
WHMR_BLK_SCHED=0 run sched0
run base
