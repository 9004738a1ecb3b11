"""Can the Tz head's first convolution (7x7 s3, N = 64: LDS-DMA-bound, matrix pipes ~13 % busy) run UNDER deconv 3 (MFMA-bound) when the batch
is split in two?  serial: deconv3(64) -> conv(64);  split: deconv3(A) -> [conv(A) on a side stream || deconv3(B)] -> conv(B).  Shapes of
whmr.py:488-498 / 419-420 at batch 64; deconv tiles that leave LDS for a 49-KB conv workgroup beside them (128: 72 KB) against the default."""
import math
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whmr_amd import _lib as L
dev = torch.device('cuda:0')
B, H, W, C = 64, 64, 48, 256
x = torch.randn(B, H, W, C, device=dev).bfloat16()
ph = (torch.randn(4, C, 4 * C, device=dev) / math.sqrt(4 * C)).bfloat16()
shift = torch.randn(C, device=dev)
fmap = torch.empty(B, 2 * H, 2 * W, C, device=dev, dtype=torch.bfloat16)
w0 = (torch.randn(64, 49 * C, device=dev) / math.sqrt(49 * C)).bfloat16()
H1, W1 = (2 * H - 7) // 3 + 1, (2 * W - 7) // 3 + 1
y0 = torch.empty(B, H1, W1, 64, device=dev, dtype=torch.bfloat16)


def deconv(b0, b1, tile):
    n = b1 - b0
    L.gemm(x[b0:b1], ph, fmap[b0:b1], bias=shift, act=L.ACT_RELU, tile=tile,
           conv=dict(IH=H, IW=W, Cin=C, OH=H, OW=W, KW=2, SH=1, SW=1, PH=1, PW=1),
           scatter=dict(c_off=0, osb=4 * H * W * C, osy=4 * W * C, osx=2 * C), phases=dict(cy=2 * W * C, cx=C))


def conv(b0, b1):
    L.gemm(fmap[b0:b1], w0, y0[b0:b1].view(-1, 64), conv=dict(IH=2 * H, IW=2 * W, Cin=C, OH=H1, OW=W1, KW=7, SH=3, SW=3, PH=0, PW=0, chunk_major=True))


side = torch.cuda.Stream()


def serial(tile):
    deconv(0, B, tile)
    conv(0, B)


def split(tile, parts=2):
    main = torch.cuda.current_stream()
    step = B // parts
    deconv(0, step, tile)
    for i in range(parts):
        ev = torch.cuda.Event(); ev.record(main)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            conv(i * step, (i + 1) * step)
        if i + 1 < parts:
            deconv((i + 1) * step, (i + 2) * step, tile)
    main.wait_stream(side)


def timeit(fn, n=10, w=3):
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


print('deconv3 alone (default tile) %.1f us, conv alone %.1f us' % (timeit(lambda: deconv(0, B, None)), timeit(lambda: conv(0, B))))
for tile in (None, 128, 192, 64):
    print('tile %-4s  deconv alone %.1f | serial %.1f us | split x2 %.1f us | split x4 %.1f us' % (tile, timeit(lambda: deconv(0, B, tile)), timeit(lambda: serial(tile)),
                                                                                      timeit(lambda: split(tile, 2)), timeit(lambda: split(tile, 4))), flush=True)
