#!/bin/bash
# Round 4: the four weight gradients of a ViT layer in ONE launch of the TN kernel (whmr_gemm_tn_bf16_group, two K slices) against the four single
# launches (7-28 slices each), and the four-wave body (one wave per SIMD; WHMR_TN_W4 bit 0 plain products, bit 1 gathering convolution products) against
# the eight-wave ping-pong body; the W-HMR training step, interleaved on one box -> profiles/r04_tn_group_ab.txt
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
fmt() { grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 ms/step', round(d['ms_per_step'],3))"; }
run() { env $1 python bench.py --workload whmr_train --no-cpu --steps 20 --warmup 5 2>/dev/null | fmt "$1"; }
echo "# $(date -u +%FT%TZ)"
python -m pytest tests/test_kernels_gpu.py -q -k "gemm_tn or conv_dw" 2>&1 | tail -2
for i in 1 2 3; do run "WHMR_TN_GROUP=0 WHMR_TN_W4=0"; run "WHMR_TN_GROUP=1 WHMR_TN_W4=0"; run "WHMR_TN_GROUP=1 WHMR_TN_W4=1"; run "WHMR_TN_GROUP=1 WHMR_TN_W4=3"; done
