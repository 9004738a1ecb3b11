"""fp32 (exact-f32 MFMA) GEMM at the ViT-B shapes, batch 64: 128x128 double-buffered kernel vs the 64x64 kernel; fp32-mode ViT forward."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whmr_amd import _lib as L
dev = torch.device('cuda:0')
def timeit(fn, n=10, w=2):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
M = 12544
for N, K in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.03; out = torch.empty(M, N, device=dev)
    t = {}
    for big in (1, 0):
        L.gemm_f32_set_big(big)
        t[big] = timeit(lambda: L.gemm(a, w, out))
    L.gemm_f32_set_big(1)
    fl = 2.0 * M * N * K
    print('%5d x %4d x %4d: 128x128 kernel %.3f ms (%.1f TFLOP/s)   64x64 kernel %.3f ms (%.1f TFLOP/s)' % (M, N, K, t[1], fl / t[1] / 1e9, t[0], fl / t[0] / 1e9), flush=True)
from whmr_amd.models.pose_vit import ViT
m = ViT(img_size=224, qkv_bias=True, numerics='fp32').to(dev).eval()
x = torch.randn(64, 3, 224, 224, device=dev)
with torch.no_grad():
    for big in (1, 0):
        L.gemm_f32_set_big(big)
        print('ViT-B 224 batch 64 fp32 parity mode, big kernel %d: %.2f ms' % (big, timeit(lambda: m(x), n=3, w=1)), flush=True)
L.gemm_f32_set_big(1)
