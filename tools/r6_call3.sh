#!/bin/bash
# Round 6, third GPU call: the reference's SyncBatchNorm + DDP wrap at world 1, camera-branch launch position A/B (graph replay), MFMA ceiling (inline asm).
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests/test_train_gpu.py -m gpu -q -x -s -k "ddp_wrap or sync_batchnorm" 2>&1 | grep -v "^\[W\|^$" | tail -12
python - <<PY
import torch
from whmr_amd import _lib as L
dev = torch.device('cuda:0')
for _ in range(2):
    print('mfma ceiling', L.mfma_ceiling(dev), 'hbm copy GB/s', L.hbm_copy_ceiling(dev))
PY
for i in 1 2; do
for pos in early vit loop; do
  WHMR_CAM_LAUNCH=$pos python bench.py --workload whmr --no-cpu --no-parity --steps 30 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cam launch $pos: whmr bf16 ms', round(d['ms_per_step'],3), round(d['roofline'].get('sclk_mhz_observed') or 0))"
done; done
for pos in early vit loop; do
  WHMR_CAM_LAUNCH=$pos python bench.py --workload whmr --numerics bf16x3 --no-cpu --no-parity --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cam launch $pos: whmr x3 ms', round(d['ms_per_step'],3))"
  WHMR_CAM_LAUNCH=$pos python bench.py --workload whmr --eager --no-cpu --no-parity --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cam launch $pos: whmr bf16 EAGER ms', round(d['ms_per_step'],3))"
done
python bench.py --workload whmr --full-x none --no-cpu --no-parity --steps 30 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no cam_model at all: whmr bf16 ms', round(d['ms_per_step'],3))"
timeout 900 python -m pytest tests/test_hotpath_gpu.py -m gpu -q -x 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/r6_prof_whmr -o whmr -- python3 $R/bench.py --workload whmr --no-cpu --no-parity --steps 10 --warmup 3 > $OUT/r6_prof_whmr.log 2>&1
DB=$(find $OUT/r6_prof_whmr -name '*.db' | head -1)
python3 $R/tools/whmr_timeline.py $DB 0 > $OUT/r6_whmr_timeline_vit.txt 2>&1
rm -rf $OUT/r6_prof_whmr
head -3 $OUT/r6_whmr_timeline_vit.txt; tail -4 $OUT/r6_whmr_timeline_vit.txt
