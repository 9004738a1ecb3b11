"""Split-bf16 (bf16x3) attention core timing at the two ViT-B shapes, batch 64: persistent 16-row-tile kernel (round 5) vs the round-3 kernel."""
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
    qkv = torch.randn(B * N, 2304, device=dev)
    hi, lo = L.split_bf16(qkv)
    qh, ql = L.to_blocked(hi), L.to_blocked(lo)
    oh = torch.empty((B * N + 31) // 32, 96, 32, 8, device=dev, dtype=torch.bfloat16)
    ol = torch.empty_like(oh)
    for var, name in ((0, 'persistent 16-row-tile kernel (round 5)'), (1, 'round-3 kernel')):
        L.attention_x3_set_variant(var)
        print('N=%d split-bf16 attention, %s: %.1f us' % (N, name, timeit(lambda: L.attention_blk(qh, oh, B, N, 12, 0.125, qkv_lo=ql, out_lo=ol))))
L.attention_x3_set_variant(0)
