"""In-model A/B: blocked-layout ViT path vs the row-major path (ViT-B/16 224^2 batch 64, interleaved), + per-slot blocked tile overrides.
usage: python tools/blk_ab.py ["slot:tile[,slot:tile]" ...]    (tile in hex, e.g. 0:43 for qkv on the 224-row tile)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whmr_amd import _lib as L
from whmr_amd.models.pose_vit import ViT
dev = torch.device('cuda:0')
B, res = 64, ((256, 192) if os.environ.get('BLK_AB_RES') == '256x192' else 224)
def timeit(fn, n=10, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
m = ViT(img_size=res, qkv_bias=True, numerics='bf16').to(dev).eval()
x = torch.randn(B, 3, *(res if isinstance(res, tuple) else (res, res)), device=dev)
def fwd_ms(): return min(timeit(lambda: m(x)) for _ in range(3))
def setcfg(cfg):
    for s in range(4): L.lib().whmr_gemm_blk_set_tile(s, 0)
    for s, t in cfg: L.lib().whmr_gemm_blk_set_tile(s, t)
cfgs = [[(int(kv.split(':')[0]), int(kv.split(':')[1], 16)) for kv in a.split(',')] for a in sys.argv[1:]]
for rnd in range(3):
    m.blocked = False; rm = fwd_ms()
    m.blocked = True; setcfg([]); base = fwd_ms()
    line = 'round %d: row-major %.3f ms  blocked %.3f ms' % (rnd, rm, base)
    for c in cfgs:
        setcfg(c); t = fwd_ms()
        setcfg([]); b2 = fwd_ms()
        line += '  %s %+.0f us' % (','.join('%d:%x' % kv for kv in c), (t - 0.5 * (base + b2)) * 1e3)
        base = b2
    print(line, flush=True)
setcfg([])
# per-kernel timing of one blocked forward
L.PROFILE = []
m(x); torch.cuda.synchronize()
prof, L.PROFILE = L.PROFILE, None
per = {}
for name, f, e0, e1 in prof:
    per.setdefault(round(f / 1e9, 1), []).append(e0.elapsed_time(e1) * 1e3)
print('GEMM launches by GFLOP: ' + '  '.join('%.1f GF: %.1f us (%.0f TF) x%d' % (k, sum(v) / len(v), k * 1e3 / (sum(v) / len(v)), len(v)) for k, v in sorted(per.items())))
tot_f = sum(f for _, f, _, _ in prof); tot_t = sum(e0.elapsed_time(e1) for _, _, e0, e1 in prof) * 1e-3
print('all GEMM launches: %.0f TF = %.3f of 2500' % (tot_f / tot_t / 1e12, tot_f / tot_t / 2.5e15))
# LayerNorm folding on / off
for rnd in range(3):
    m.ln_fold = True; a = fwd_ms()
    m.ln_fold = False; b = fwd_ms()
    print('round %d: LayerNorm folded %.3f ms   explicit LayerNorm passes %.3f ms' % (rnd, a, b), flush=True)
m.ln_fold = True
