"""GEMM micro-benchmark on the box: the 4 ViT-B shapes x epilogue kinds, vs hipBLASLt (torch.matmul) on the same data."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whmr_amd import _lib as L

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
M = B * 196


def timeit(fn, n=30, w=5):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


tot = {}
for name, (N, K, kind) in {'qkv': (2304, 768, 'bf16'), 'proj': (768, 768, 'res'), 'fc1': (3072, 768, 'gelu'), 'fc2': (768, 3072, 'res')}.items():
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
    bias = torch.randn(N, device=dev)
    if kind == 'res':
        out = torch.randn(M, N, device=dev)
        fn = lambda: L.gemm(a, w, out, bias=bias, residual=out)
    elif kind == 'gelu':
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        fn = lambda: L.gemm(a, w, out, bias=bias, act=L.ACT_GELU)
    else:
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        fn = lambda: L.gemm(a, w, out, bias=bias)
    ms = timeit(fn)
    o2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ms0 = timeit(lambda: L.gemm(a, w, o2))
    msb = timeit(lambda: torch.matmul(a, w.t()))
    for tile in (64, 192, 257, 320):
        if kind == 'res':
            f2 = lambda: L.gemm(a, w, out, bias=bias, residual=out, tile=tile)
        elif kind == 'gelu':
            f2 = lambda: L.gemm(a, w, out, bias=bias, act=L.ACT_GELU, tile=tile)
        else:
            f2 = lambda: L.gemm(a, w, out, bias=bias, tile=tile)
        mst = timeit(f2)
        mst0 = timeit(lambda: L.gemm(a, w, o2, tile=tile))
        msm = timeit(lambda: L.gemm(a, w, o2, tile=tile, res_row_mod=-12345))
        print('      tile %dx256: main-loop-only %.1f us %.0f TF' % (tile, msm * 1e3, 2.0 * M * N * K / msm / 1e9))
        print('      tile %dx256: fused %.1f us %.0f TF | plain %.1f us %.0f TF' % (tile, mst * 1e3, 2.0 * M * N * K / mst / 1e9, mst0 * 1e3, 2.0 * M * N * K / mst0 / 1e9))
    tf = lambda t: 2.0 * M * N * K / t / 1e9
    print('%-5s M=%d N=%d K=%d  fused-epilogue %.1f us %.0f TF | plain bf16-out %.1f us %.0f TF | hipBLASLt %.1f us %.0f TF'
          % (name, M, N, K, ms * 1e3, tf(ms), ms0 * 1e3, tf(ms0), msb * 1e3, tf(msb)))
    tot[name] = ms
print('sum per layer: %.1f us' % (sum(tot.values()) * 1e3))
