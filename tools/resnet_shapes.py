"""Tile / split-K sweep over the GEMM shapes of cam_model's ResNet-50 at n x 600x800 (tunes the chooser in gemm_bf16.hip).

    python tools/resnet_shapes.py [n_images]
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whmr_amd import _lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda:0')
H1, W1 = 150, 200
def hw(l): return ((H1 - 1) // (2 ** l) + 1 if l else H1, (W1 - 1) // (2 ** l) + 1 if l else W1)
dims = {0: (150, 200), 1: (75, 100), 2: (38, 50), 3: (19, 25)}
def bench(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

shapes = []   # (name, kind, in_level, out_level, Cin, N, skip)
for li, planes in enumerate([64, 128, 256, 512]):
    cin_first = 64 if li == 0 else planes * 2
    lin = max(li - 1, 0)
    shapes += [('l%d.0.c1' % (li + 1), '1x1', lin, lin, cin_first, planes, False),
               ('l%d.0.c2' % (li + 1), '3x3', lin, li, planes, planes, False),
               ('l%d.0.down' % (li + 1), 'down', lin, li, cin_first, planes * 4, False),
               ('l%d.x.c3' % (li + 1), '1x1', li, li, planes, planes * 4, True),
               ('l%d.x.c1' % (li + 1), '1x1', li, li, planes * 4, planes, False),
               ('l%d.x.c2' % (li + 1), '3x3', li, li, planes, planes, False)]

def bench(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

if len(sys.argv) > 2 and sys.argv[2] == 'tz':        # Tz head conv0 of W-HMR (whmr.py:419): 7x7 s3, 256 -> 64 on the 128x96 map, batch n
    for bs in (n,):
        x = torch.randn(bs, 128, 96, 256, device=dev).bfloat16()
        w = (torch.randn(64, 49 * 256, device=dev) / 112.).bfloat16()
        out = torch.empty(bs, 41, 31, 64, device=dev, dtype=torch.bfloat16)
        conv = dict(IH=128, IW=96, Cin=256, OH=41, OW=31, KW=7, SH=3, SW=3, PH=0, PW=0)
        for cm in (False, True):
            conv['chunk_major'] = cm
            for tile in (None, 65, 64):
                print('tz conv0 B=%d chunk_major=%d tile %s: %.1f us' % (bs, cm, tile, bench(lambda: L.gemm(x, w, out.view(-1, 64), conv=conv, tile=tile))), flush=True)
    sys.exit(0)
for name, kind, lin, lout, Cin, N, skip in shapes:
    IH, IW = dims[lin]; OH, OW = dims[lout]; s = 1 if lin == lout else 2
    x = torch.randn(n, IH, IW, Cin, device=dev).bfloat16()
    KW = 3 if kind == '3x3' else 1
    K = Cin * KW * KW
    w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    bias = torch.randn(N, device=dev)
    out = torch.empty(n, OH, OW, N, device=dev, dtype=torch.bfloat16)
    sk = torch.randn(n * OH * OW, N, device=dev).bfloat16() if skip else None
    conv = None
    if kind == '3x3' or s == 2:
        conv = dict(IH=IH, IW=IW, Cin=Cin, OH=OH, OW=OW, KW=KW, SH=s, SW=s, PH=KW // 2, PW=KW // 2)
    M = n * OH * OW
    fl = 2.0 * M * N * K
    byt = (x.numel() * (1 if KW == 1 and s == 1 else 1) + w.numel() + out.numel() * (2 if skip else 1)) * 2
    res = []
    def run(tile=None, splits=None):
        return bench(lambda: L.gemm(x, w, out, bias=bias, act=L.ACT_RELU, conv=conv, residual=sk, res_first=skip, tile=tile, splits=splits))
    res.append(('auto', run()))
    for tile in (65, 64, 128, 256, 192, 257, 320):
        if tile in (128, 256) and conv is not None and Cin % 32: continue
        res.append((str(tile), run(tile)))
        if K >= 512 and tile in (65, 64, 128):
            for sp in (2, 3, 4, 8):
                if K // sp >= 128 and sp * M * N * 4 <= (64 << 20):
                    res.append(('%d/%d' % (tile, sp), run(tile, sp)))
    best = min(res[1:], key=lambda r: r[1])
    print('%-10s M=%7d N=%5d K=%5d  %6.1f GF  hbm-floor %5.1f us | auto %6.1f us  best %s %6.1f us (%4.0f TF) | %s' % (
        name, M, N, K, fl / 1e9, byt / 8e12 * 1e6, res[0][1], best[0], best[1], fl / best[1] / 1e6,
        ' '.join('%s:%.0f' % r for r in res[1:])), flush=True)
