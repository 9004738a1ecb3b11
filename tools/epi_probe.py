"""Epilogue decomposition of the four ViT-B GEMMs (M = 12544): full kernel vs main loop only vs cheaper epilogues."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whmr_amd import _lib as L
dev = torch.device('cuda:0')
M = 12544
if len(sys.argv) > 1: M = int(sys.argv[1])
def timeit(fn, n=30, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, N, K, tiles in (('qkv', 2304, 768, (257, 224, 257, 224)), ('proj', 768, 768, (192, 160, 192, 160)), ('fc1', 3072, 768, (320, 224)), ('fc2', 768, 3072, (192, 160, 192, 160))):
    a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
    bias = torch.randn(N, device=dev)
    ob = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    of = torch.empty(M, N, device=dev)
    res = torch.randn(M, N, device=dev)
    for tile in tiles:
        r = {}
        r['main'] = timeit(lambda: L.gemm(a, w, ob, tile=tile or tiles[1], res_row_mod=-12345))
        r['bf16'] = timeit(lambda: L.gemm(a, w, ob, bias=bias, tile=tile))
        r['bf16+gelu'] = timeit(lambda: L.gemm(a, w, ob, bias=bias, act=L.ACT_GELU, tile=tile))
        r['f32'] = timeit(lambda: L.gemm(a, w, of, bias=bias, tile=tile))
        r['f32+res'] = timeit(lambda: L.gemm(a, w, of, bias=bias, residual=res, tile=tile))
        r['f32+res inplace'] = timeit(lambda: L.gemm(a, w, res, bias=bias, residual=res, tile=tile))
        print('%-4s tile %-4s ' % (name, tile) + '  '.join('%s %.1f' % kv for kv in r.items()) + '   (2MNK at 1 PF: %.1f us)' % (2.0 * M * N * K / 1e9), flush=True)
