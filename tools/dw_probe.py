"""Weight-gradient GEMMs of the ViT backward (dW = dY^T . X: small M x N, K = tokens = 12544): tile / split-K sweep."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whmr_amd import _lib as L
dev = torch.device('cuda:0')
K = 12544
def timeit(fn, n=10, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, M, N in (('dWqkv', 2304, 768), ('dWproj', 768, 768), ('dW1', 3072, 768), ('dW2', 768, 3072)):
    a = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev)
    res = [('auto', timeit(lambda: L.gemm(a, w, out)))]
    for tile in (64, 128, 257, 192):
        for sp in (2, 4, 6, 8, 12, 16, 24):
            if sp * M * N * 4 > (128 << 20): continue
            try:
                res.append(('%d/%d' % (tile, sp), timeit(lambda: L.gemm(a, w, out, tile=tile, splits=sp))))
            except Exception as e:
                pass
    best = min(res, key=lambda r: r[1])
    print('%-6s %4dx%4d  %5.1f GF | auto %.1f us | best %s %.1f us (%.0f TF) | %s' % (name, M, N, 2.0 * M * N * K / 1e9, res[0][1], best[0], best[1],
          2.0 * M * N * K / best[1] / 1e6, ' '.join('%s:%.0f' % r for r in res[1:])), flush=True)
