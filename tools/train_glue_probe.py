"""Which torch (aten) ops are still inside the W-HMR training step, and where do they come from?  One profiled eager step of bench.py's
whmr_train workload (torch.profiler, stacks + shapes); device time per (op, input shapes, first frame inside the repo).  The HIP kernels are
launched through ctypes and carry no aten op, so everything listed here is glue.  `python tools/train_glue_probe.py [top]`."""
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity

import bench

top = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 70
args = bench.parse(['--workload', 'whmr_train', '--no-cpu'])
dev = torch.device('cuda:0')
step = bench.build_workload(args, dev)[0]
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
agg = defaultdict(lambda: [0.0, 0])
total = 0.0
for e in prof.events():
    t = getattr(e, 'self_device_time_total', 0.0)
    if not t or not e.name.startswith(('aten::', 'Optimizer', 'autograd::')) and 'Backward' not in e.name:
        continue
    frames = [s for s in (e.stack or []) if ('w-hmr_amd' in s or 'whmr_amd' in s or 'bench.py' in s) and 'tools/' not in s]
    where = frames[0].replace(root + '/', '') if frames else (e.stack[0] if e.stack else '?')
    shapes = str([s for s in (e.input_shapes or []) if s])[:70]
    k = (e.name, shapes, where[:110])
    agg[k][0] += t
    agg[k][1] += 1
    total += t
print('torch ops with device time in one training step: %.2f ms' % (total / 1e3))
if '--by-count' in sys.argv:
    byname = defaultdict(lambda: [0.0, 0])
    for (name, shapes, where), (t, n) in agg.items():
        byname[(name, shapes)][0] += t
        byname[(name, shapes)][1] += n
    for (name, shapes), (t, n) in sorted(byname.items(), key=lambda kv: -kv[1][1])[:top]:
        print('x%-4d %8.1f us  %-34s %s' % (n, t, name[:34], shapes))
    raise SystemExit(0)
for (name, shapes, where), (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
    print('%8.1f us x%-3d %-28s %-70s %s' % (t, n, name[:28], shapes, where))
