"""Fused IUV losses (csrc/iuv_loss.hip) at the training shape: batch 64, 128 x 96 map, bf16 logits in the padded [P, 128] layout; us per launch and
achieved HBM rate (forward reads 256 B rows + 12 B of targets per pixel; backward also writes 256 B)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whmr_amd import _lib as L
dev = torch.device('cuda:0')
B, H, W = 64, 128, 96
y = (torch.randn(B * H * W, 128, device=dev) * 1.5).bfloat16().view(B, H, W, 128)[..., :90]
part = torch.randint(0, 25, (B, H, W + 32), device=dev).float()
img = torch.stack([part / 24, torch.rand(B, H, W + 32, device=dev), torch.rand(B, H, W + 32, device=dev)], 1)[:, :, :, 16:-16]
g = torch.ones(4, device=dev)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
P = B * H * W
tf = timeit(lambda: L.iuv_losses(y, img, 0.125))
tb = timeit(lambda: L.iuv_losses_bwd(y, img, 0.125, g, 128))
print('forward  %.1f us  (%.0f GB/s of %d MB)' % (tf, P * (180 + 12) / tf / 1e3, P * (180 + 12) / 1e6))
print('backward %.1f us  (%.0f GB/s of %d MB)' % (tb, P * (180 + 12 + 256) / tb / 1e3, P * (180 + 12 + 256) / 1e6))
