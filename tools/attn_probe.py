"""Attention core timing at the two ViT-B shapes (batch 64, 12 heads): row-major chunked / single-pass kernels, the blocked-layout kernel and
its staging-only / key-loop-only ablations (wrong results, timing only)."""
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
        print('N=%d row-major %s: %.1f us' % (N, 'chunked' if var else 'single-pass', timeit(lambda: L.attention(qkv, att, B, N, 12, 64, 0.125))))
    qb = L.to_blocked(qkv.view(B * N, 2304))
    ob = torch.empty((B * N + 31) // 32, 96, 32, 8, device=dev, dtype=torch.bfloat16)
    L.attention_set_variant(1)
    print('N=%d blocked, persistent 16-row-tile kernel (round 5): %.1f us' % (N, timeit(lambda: L.attention_blk(qb, ob, B, N, 12, 0.125))))
    for var, name in ((1 | 2, 'no operand traffic after the first item'), (1 | 4, 'no arithmetic after the first item'), (1 | 2 | 4, 'neither')):
        L.attention_set_variant(var)
        print('N=%d blocked, persistent kernel, %s: %.1f us' % (N, name, timeit(lambda: L.attention_blk(qb, ob, B, N, 12, 0.125))))
    for var, name in ((1 | 16, 'round-2 kernel, full'), (1 | 16 | 4, 'round-2 kernel, staging only (no key loop)'), (1 | 16 | 2, 'round-2 kernel, key loop only (no staging)'), (1 | 16 | 2 | 4, 'round-2 kernel, neither (launch + Q load + store)')):
        L.attention_set_variant(var)
        print('N=%d blocked %s: %.1f us' % (N, name, timeit(lambda: L.attention_blk(qb, ob, B, N, 12, 0.125))))
L.attention_set_variant(1)
for N in (196, 192):
    q32 = torch.randn(B, N, 2304, device=dev)
    a32 = torch.empty(B, N, 768, device=dev)
    for var, name in ((1, 'matrix-pipe kernel'), (1 | 8, 'VALU kernel')):
        L.attention_set_variant(var)
        print('N=%d fp32 attention, %s: %.1f us' % (N, name, timeit(lambda: L.attention(q32, a32, B, N, 12, 64, 0.125), n=10, w=2)))
L.attention_set_variant(1)
