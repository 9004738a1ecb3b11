import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
N, K = 2304, 768
a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
alias = torch.empty(1, N, device=dev, dtype=torch.bfloat16).expand(M, N)
print('fill 57.8MB bf16: %.1f us' % timeit(lambda: out.fill_(1.0)))
big = torch.empty(M, N * 4, device=dev, dtype=torch.bfloat16)
print('fill 231MB: %.1f us' % timeit(lambda: big.fill_(1.0)))
print('copy 57.8MB->57.8MB: %.1f us' % timeit(lambda: out.copy_(big[:, :N])))
for tile in (128, 256):
    print('tile', tile, 'normal %.1f us' % timeit(lambda: L.gemm(a, w, out, tile=tile)),
          'aliased-rows (no HBM writes) %.1f us' % timeit(lambda: L.gemm(a, w, alias, tile=tile)),
          'main-loop-only %.1f us' % timeit(lambda: L.gemm(a, w, out, tile=tile, res_row_mod=-12345)))
print('128x128 kernel normal %.1f us' % timeit(lambda: L.gemm(a, w, out)), 'aliased %.1f us' % timeit(lambda: L.gemm(a, w, alias)))
