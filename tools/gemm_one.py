"""Run one GEMM variant a few times (for rocprofv3 --pmc passes).  usage: gemm_one.py N K tile [probe]"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whmr_amd import _lib as L
dev = torch.device('cuda:0')
N, K, tile = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
probe = int(sys.argv[4]) if len(sys.argv) > 4 else 0
M = 12544
a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for _ in range(5):
    L.gemm(a, w, out, tile=(tile if tile else None), res_row_mod=probe)
torch.cuda.synchronize()
