"""In-model A/B of the trimmed tiles (224 / 160) against the untrimmed 256 / 192 kernels: ViT-B/16 224^2 batch 64, interleaved on one box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whmr_amd import _lib as L
from whmr_amd.models.pose_vit import ViT
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
res = int(sys.argv[2]) if len(sys.argv) > 2 else 224
def timeit(fn, n=10, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
m = ViT(img_size=res, qkv_bias=True, numerics='bf16').to(dev).eval()
x = torch.randn(B, 3, res, res, device=dev)
def fwd_ms(): return min(timeit(lambda: m(x)) for _ in range(3))
def setall(q, pr, f2):
    L.set_option(100, q); L.set_option(101, pr); L.set_option(103, f2)
for rnd in range(4):
    setall(257, 192, 192); old = fwd_ms()
    setall(0, 0, 0); new = fwd_ms()
    setall(224, 192, 192); q = fwd_ms()
    setall(257, 160, 192); pr = fwd_ms()
    setall(257, 192, 160); f2 = fwd_ms()
    print('round %d: untrimmed %.3f ms  chooser %.3f ms   only qkv->224 %+.0f us  only proj->160 %+.0f us  only fc2->160 %+.0f us' %
          (rnd, old, new, (q - old) * 1e3, (pr - old) * 1e3, (f2 - old) * 1e3), flush=True)
setall(0, 0, 0)
