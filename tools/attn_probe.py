"""Attention core timing at the two ViT-B shapes (batch 64, 12 heads): chunked kernel vs the single-pass variant."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whmr_amd import _lib as L
dev = torch.device('cuda:0')
B = 64
def timeit(fn, n=50, w=10):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for N in (196, 192):
    qkv = torch.randn(B, N, 2304, device=dev).bfloat16()
    att = torch.empty(B, N, 768, device=dev, dtype=torch.bfloat16)
    for var in (1, 0):
        L.attention_set_variant(var)
        print('N=%d %s: %.1f us' % (N, 'chunked' if var else 'single-pass', timeit(lambda: L.attention(qkv, att, B, N, 12, 64, 0.125))))
L.attention_set_variant(1)
