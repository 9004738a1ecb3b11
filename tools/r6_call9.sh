#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_hotpath_gpu.py -m gpu -q -x -s -k "smpl" 2>&1 | grep -v "^$" | tail -12
python - <<PY
import torch, time
from oracle import synth, geometry as OG
from whmr_amd.models.smpl import SMPL
dev = torch.device('cuda:0')
assets = synth.make_assets(0)
m = SMPL(arrays=assets['smpl'], marker_ids=assets['ssm']).to(dev)
for B in (1, 64):
    g = torch.Generator().manual_seed(B)
    betas = torch.randn(B, 10, generator=g).to(dev)
    rot = OG.batch_rodrigues(torch.randn(B * 24, 3, generator=g) * 0.5).view(B, 24, 3, 3).to(dev)
    for x3 in (False, True, False, True):
        m.offsets_x3 = x3
        fn = lambda: m.run(betas, rot, gram_schmidt=True, want_aa=True, want_smpl_joints=True, want_markers=True)
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3): fn()
        torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(20): fn()
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); gr.replay(); e1.record(); torch.cuda.synchronize()
        print('B = %d SMPL call, offsets %s: %.2f us per call' % (B, 'split-bf16' if x3 else 'exact f32 ', e0.elapsed_time(e1) * 1e3 / 40), flush=True)
PY
timeout 900 python -m pytest tests/test_hotpath_gpu.py tests/test_kernels_gpu.py -m gpu -q -x 2>&1 | tail -3
for i in 1 2; do
WHMR_SMPL_X3=0 python bench.py --workload whmr --no-cpu --no-parity --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('exact offsets: whmr ms', round(d['ms_per_step'],3), {k: round(v['avg_us'],2) for k,v in d['hbm_rows'].items()})"
WHMR_SMPL_X3=1 python bench.py --workload whmr --no-cpu --no-parity --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('x3 offsets:    whmr ms', round(d['ms_per_step'],3), {k: round(v['avg_us'],2) for k,v in d['hbm_rows'].items()})"
done
