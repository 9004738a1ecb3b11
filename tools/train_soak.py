"""Soak of bench.py's whmr_train step: N steps, loss / gradient finiteness and allocator footprint every 20 steps (a leak or a NaN shows here, not in a 30-step timing run)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
args = bench.parse(['--workload', 'whmr_train', '--no-cpu'] + sys.argv[2:])
step = bench.build_workload(args, torch.device('cuda:0'))[0]
for i in range(n):
    loss = step()
    if i % 20 == 0 or i == n - 1:
        torch.cuda.synchronize()
        print('step %3d  loss %.6f  finite %s  allocated %.2f GB  reserved %.2f GB' % (i, float(loss), bool(torch.isfinite(loss)),
              torch.cuda.memory_allocated() / 2 ** 30, torch.cuda.memory_reserved() / 2 ** 30), flush=True)
