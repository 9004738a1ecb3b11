"""Round 6: the Tz head's two convolutions (whmr.py:418-421) composed into ONE Conv2d(256, 5, k25, s6) and evaluated as a
space-to-depth implicit GEMM (M = B * 22 * 16, N = 128, K = 36 * 256) + a 25-term fold, against the two-convolution form; batch 64, one box.
   python tools/r6_tz_probe.py [bf16|bf16x3|fp32]"""
import math
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from whmr_amd import _lib as L
from whmr_amd.models.whmr import compose_tz_weights

dev = torch.device('cuda:0')
mode = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
B, H, W, C = 64, 128, 96, 256
g = torch.Generator().manual_seed(0)
x = torch.relu(torch.randn(B, H, W, C, generator=g)).to(dev)
w0 = (torch.randn(64, C, 7, 7, generator=g) / math.sqrt(49 * C)).to(dev)
w1 = (torch.randn(5, 64, 7, 7, generator=g) / math.sqrt(49 * 64)).to(dev)
ref = F.conv2d(F.conv2d(x[:8].double().permute(0, 3, 1, 2), w0.double(), stride=3), w1.double(), stride=2).reshape(8, 5, -1)


def timeit(fn, n=20, w=5):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


G = compose_tz_weights(w0, w1)
OHp, OWp = 22, 16
if mode == 'bf16':
    a, gw = x.bfloat16().contiguous(), G.bfloat16().contiguous()
    Cp, halves = C, 1
elif mode == 'bf16x3':
    hi, lo = L.split_bf16(x)
    a = torch.cat([hi, lo], -1).contiguous()
    ghi, glo = L.split_bf16(G)
    ghi, glo = ghi.view(128, 36, C), glo.view(128, 36, C)
    gw = torch.cat([torch.cat([ghi, ghi], -1), torch.cat([glo, torch.zeros_like(glo)], -1)], 0).reshape(256, -1).contiguous()
    Cp, halves = 2 * C, 2
else:
    a, gw, Cp, halves = x, G, C, 1
P = torch.empty(B * OHp * OWp, gw.shape[0], dtype=torch.float32, device=dev)
tok = torch.empty(B * 5, 216, dtype=torch.float32, device=dev)
conv = dict(IH=H, IW=OWp, Cin=6 * Cp, OH=OHp, OW=OWp, KW=1, SH=6, SW=1, PH=0, PW=0)
av = a.view(B, H, OWp, 6 * Cp)
flops = 2.0 * B * OHp * OWp * gw.shape[0] * gw.shape[1]


def composed(**kw):
    L.gemm(av, gw, P, conv=conv, **kw)
    L.tz_fold(P, tok, B, OHp, OWp, 18, 12, halves=halves)


combos = [dict()] if mode == 'fp32' else [dict(), dict(tile=64), dict(tile=64, splits=2), dict(tile=64, splits=3), dict(tile=65), dict(tile=65, splits=2),
                                          dict(tile=128), dict(tile=128, splits=2), dict(tile=128, splits=3), dict(tile=257, splits=3), dict(tile=257, splits=6),
                                          dict(tile=192, splits=2), dict(tile=192, splits=4)]
for kw in combos:
    try:
        tok.fill_(float('nan'))
        composed(**kw)
        err = ((tok.view(B, 5, -1)[:8].double() - ref).abs().max() / ref.abs().max()).item()
        t = timeit(lambda: composed(**kw))
        tg = timeit(lambda: L.gemm(av, gw, P, conv=conv, **kw))
        print('%-7s composed %-28s total %7.1f us (gemm %7.1f us = %4.0f TF/s, map read %5.2f TB/s)  max-rel vs fp64 two-conv %.2e'
              % (mode, kw or 'chooser', t, tg, flops / tg / 1e6, a.numel() * a.element_size() / tg / 1e6, err), flush=True)
    except Exception as e:          # a tile / split combination the launcher rejects
        print('%-7s composed %-28s rejected: %s' % (mode, kw, str(e)[:80]), flush=True)

# the two-convolution form (what _tz_head ran until round 5)
H1, W1 = 41, 30
w1r = w1.permute(0, 2, 3, 1).reshape(5, 49, 64).contiguous()
if mode == 'bf16':
    w0r = w0.permute(0, 2, 3, 1).reshape(64, 49, C // 64, 64).permute(0, 2, 1, 3).reshape(64, -1).bfloat16().contiguous()
    y0 = torch.empty(B, H1, W1, 64, dtype=torch.bfloat16, device=dev)
    old = lambda: (L.gemm(a, w0r, y0.view(-1, 64), conv=dict(IH=H, IW=W, Cin=C, OH=H1, OW=W1, KW=7, SH=3, SW=3, PH=0, PW=0, chunk_major=True)),
                   L.tz_conv1(y0, w1r, tok.view(B, 5, -1)))
elif mode == 'bf16x3':
    w0p = w0.permute(0, 2, 3, 1)
    hi, lo = L.split_bf16(w0p.contiguous())
    w2 = torch.cat([torch.cat([hi, hi], -1), torch.cat([lo, torch.zeros_like(lo)], -1)], 0)
    n, kh, kw_, ci = w2.shape
    w0r = w2.reshape(n, kh * kw_, ci // 64, 64).permute(0, 2, 1, 3).reshape(n, -1).contiguous()
    y0 = torch.empty(B, H1, W1, 128, dtype=torch.float32, device=dev)
    old = lambda: (L.gemm(a, w0r, y0.view(-1, 128), conv=dict(IH=H, IW=W, Cin=2 * C, OH=H1, OW=W1, KW=7, SH=3, SW=3, PH=0, PW=0, chunk_major=True)),
                   L.tz_conv1(y0, w1r, tok.view(B, 5, -1)))
else:
    w0r = w0.permute(0, 2, 3, 1).reshape(64, -1).contiguous()
    y0 = torch.empty(B, H1, W1, 64, dtype=torch.float32, device=dev)
    old = lambda: (L.gemm(a, w0r, y0.view(-1, 64), conv=dict(IH=H, IW=W, Cin=C, OH=H1, OW=W1, KW=7, SH=3, SW=3, PH=0, PW=0)),
                   L.tz_conv1(y0, w1r, tok.view(B, 5, -1)))
tok.fill_(float('nan'))
old()
err = ((tok.view(B, 5, -1)[:8].double() - ref).abs().max() / ref.abs().max()).item()
print('%-7s two convolutions (round 5)            total %7.1f us   max-rel vs fp64 two-conv %.2e' % (mode, timeit(old), err), flush=True)
