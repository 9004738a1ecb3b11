import sys, os, time
sys.path.insert(0, '/root/repo')
import torch
from whmr_amd import _lib as L
from whmr_amd.models.pose_vit import ViT
from whmr_amd.graph import GraphedForward
dev = torch.device('cuda:0')
m = ViT(img_size=(256, 192), qkv_bias=True, numerics='bf16').to(dev).eval()
for B in (1, 2, 4, 8):
    x = torch.randn(B, 3, 256, 192, device=dev)
    for rnd in range(2):
        for opt in (0, 1):
            L.set_option(3, opt) if False else None   # option 3 (small-M split-K) was removed after this measurement
            g = GraphedForward(m, x)
            for _ in range(5): g(x)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(30): g(x)
            torch.cuda.synchronize()
            print('B=%d small-M split=%d: ViT forward (HIP graph) %.3f ms' % (B, opt, (time.perf_counter() - t0) / 30 * 1e3), flush=True)
