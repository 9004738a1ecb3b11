import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whmr_amd import _lib as L
dev = torch.device('cuda:0')
B = 64
def timeit(fn, n=30, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
qkv = torch.randn(B, 196, 2304, device=dev).bfloat16()
att = torch.empty(B, 196, 768, device=dev, dtype=torch.bfloat16)
for name, sc in (('full', 0.125), ('no-staging', -1.0), ('staging-only', -2.0)):
    print(name, '%.1f us' % timeit(lambda: L.attention(qkv, att, B, 196, 12, 64, sc)))
