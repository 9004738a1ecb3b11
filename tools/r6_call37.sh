#!/bin/bash
# round 6, call 37: which switch makes regressor.2.fc1.weight's gradient differ between two runs of the batch-64 bf16 step
run() { echo "== $*"; for i in 1 2 3; do env "$@" python -m pytest tests/test_train_gpu.py -x -q -m gpu -k "batch64_bf16_finite" 2>&1 | grep -E "AssertionError|passed|failed" | head -2 | tr '\n' ' '; echo; done; }
run WHMR_TRAIN_GROUP_DX=0
run WHMR_TRAIN_GROUP_DX=1 WHMR_TRAIN_TZ_TAIL=0
run WHMR_TRAIN_GROUP_DX=1 WHMR_TRAIN_HEAVY_FIRST=0
run WHMR_TRAIN_GROUP_DX=1 WHMR_TRAIN_FORK3=0
run WHMR_TRAIN_GROUP_DX=1 WHMR_TRAIN_TZ_TAIL=0 WHMR_TRAIN_HEAVY_FIRST=0 WHMR_TRAIN_FORK3=0
