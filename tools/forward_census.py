"""Kernel launches of ONE eager W-HMR forward (after warm-up), by kernel name: ours (libwhmr_hip.so) vs the framework's.
    python tools/forward_census.py [batch] [numerics]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
num = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
args = bench.parse(['--workload', 'whmr', '--no-cpu', '--eager', '--batch', str(B), '--numerics', num])
step = bench.build_workload(args, torch.device('cuda:0'))[0]
with torch.no_grad():
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        step()
        torch.cuda.synchronize()
ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
rows = {}
for e in ev:
    r = rows.setdefault(e.name, [0, 0.0])
    r[0] += 1
    r[1] += e.device_time if hasattr(e, 'device_time') else e.cuda_time
fw = lambda n: n.startswith('void at::') or 'rocprim' in n or n.startswith('Cijk') or 'Memcpy' in n or 'Memset' in n or 'elementwise' in n
n_fw = sum(c for n, (c, _) in rows.items() if fw(n))
n_all = sum(c for c, _ in rows.values())
print('batch %d %s: %d device launches in one eager forward, %d of them framework / copy kernels (%.1f us of %.1f us)' % (
    B, num, n_all, n_fw, sum(t for n, (c, t) in rows.items() if fw(n)), sum(t for c, t in rows.values())))
for n, (c, t) in sorted(rows.items(), key=lambda kv: -kv[1][0])[:40]:
    print('%5d %9.1f us  %s%s' % (c, t, 'FW ' if fw(n) else '   ', n[:150]))
