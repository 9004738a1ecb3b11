"""In-model A/B of GEMM tile variants (ViT-B/16 224^2 batch 64, interleaved on one box): per-slot tile override vs the chooser.
usage: python tools/spread_ab.py "<slot>:<tile>[,<slot>:<tile>...]" ...   (slot 0 qkv, 1 proj, 2 fc1, 3 fc2)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whmr_amd import _lib as L
from whmr_amd.models.pose_vit import ViT
dev = torch.device('cuda:0')
B, res = 64, 224
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
def setcfg(cfg):
    for s in range(4): L.set_option(100 + s, 0)
    for s, t in cfg: L.set_option(100 + s, t)
cfgs = [[tuple(int(v) for v in kv.split(':')) for kv in a.split(',')] for a in sys.argv[1:]]
# correctness of every override against the chooser's result first
setcfg([]); ref = m(x).float().clone()
for c in cfgs:
    setcfg(c); out = m(x).float()
    print('cfg %s: max-rel diff vs chooser %.2e' % (c, ((out - ref).abs().max() / ref.abs().max()).item()))
for rnd in range(3):
    setcfg([]); base = fwd_ms()
    line = 'round %d: chooser %.3f ms' % (rnd, base)
    for c in cfgs:
        setcfg(c); t = fwd_ms()
        setcfg([]); b2 = fwd_ms()
        line += '  %s %+.0f us' % (','.join('%d:%d' % kv for kv in c), (t - 0.5 * (base + b2)) * 1e3)
        base = b2
    print(line, flush=True)
setcfg([])
