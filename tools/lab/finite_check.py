"""Are the activations of the garbage-numerics experiment library benign (finite, unit-scale)?  usage: finite_check.py <lib.so | ->"""
import os
import sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch
import bench
from whmr_amd import _lib
if sys.argv[1] != '-':
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
args = bench.parse(['--no-cpu'])
dev = torch.device('cuda:0')
step = bench.build_workload(args, dev)[0]
y = step()
torch.cuda.synchronize()
y = y.float()
print(sys.argv[1], 'finite', bool(torch.isfinite(y).all()), 'rms', float(y.pow(2).mean().sqrt()), 'absmax', float(y.abs().max()))
