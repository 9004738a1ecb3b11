import sys, os, math
sys.path.insert(0, '/root/repo')
import torch
from whmr_amd import _lib as L
dev = torch.device('cuda:0')
M = 12544
def timeit(fn, n=30, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, N, K, tile, act in (('qkv', 2304, 768, 257, 0), ('fc1', 3072, 768, 320, 1)):
    a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
    bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    alias = torch.empty(1, N, device=dev, dtype=torch.bfloat16).expand(M, N)
    for rnd in range(2):
        print(name, 'normal %.1f' % timeit(lambda: L.gemm(a, w, out, bias=bias, act=act, tile=tile)),
              'aliased rows (stores stay in L2) %.1f' % timeit(lambda: L.gemm(a, w, alias, bias=bias, act=act, tile=tile)),
              'no global stores %.1f' % timeit(lambda: L.gemm(a, w, out, bias=bias, act=act, tile=tile, res_row_mod=-2003)),
              'main only %.1f' % timeit(lambda: L.gemm(a, w, out, tile=tile, res_row_mod=-12345)), flush=True)

for name, N, K, tile in (('proj', 768, 768, 192), ('fc2', 768, 3072, 192)):
    a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
    bias = torch.randn(N, device=dev); t = torch.randn(M, N, device=dev)
    for rnd in range(2):
        print(name, 'normal (fp32 out + residual, in place) %.1f' % timeit(lambda: L.gemm(a, w, t, bias=bias, residual=t, tile=tile)),
              'no global stores %.1f' % timeit(lambda: L.gemm(a, w, t, bias=bias, residual=t, tile=tile, res_row_mod=-2003)),
              'main only %.1f' % timeit(lambda: L.gemm(a, w, t, tile=tile, res_row_mod=-12345)), flush=True)
