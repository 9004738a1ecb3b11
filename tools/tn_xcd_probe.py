"""TN weight-gradient kernel, grid mapping A/B: run once as is (XCD-aware (slice, tile) order) and once with WHMR_TN_RR=1 (dispatch order).
ViT-B shapes at 12544 tokens + the convolution weight gradients of the training step (IUV head 3x3 on the 128x96 map, Tz head 7x7 s3,
deconv 2 / 3 as k4 s2 p1 transposed convolutions)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whmr_amd import _lib as L
dev = torch.device('cuda:0')


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


print('mapping:', 'round robin' if os.environ.get('WHMR_TN_RR') == '1' else 'XCD-aware')
M = 12544
for n_out, k_in in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
    dy = torch.randn(M, n_out, device=dev).bfloat16()
    x = torch.randn(M, k_in, device=dev).bfloat16()
    dw = torch.empty(n_out, k_in, device=dev)
    t = timeit(lambda: L.gemm_tn(dy, x, dw))
    print('dW %4d x %4d: %.1f us (%.0f TF)' % (n_out, k_in, t, 2.0 * M * n_out * k_in / t / 1e6), flush=True)
B = 64
for name, Mo, IH, IW, C, OH, OW, KH, KW, S, P in (('iuv 3x3', 128, 128, 96, 256, 128, 96, 3, 3, 1, 1), ('tz 7x7 s3', 64, 128, 96, 256, 41, 30, 7, 7, 3, 0),
                                                   ('deconv3 k4 s2', 256, 128, 96, 256, 64, 48, 4, 4, 2, 1), ('deconv2 k4 s2', 256, 64, 48, 256, 32, 24, 4, 4, 2, 1)):
    a = torch.randn(B * OH * OW, Mo, device=dev).bfloat16()
    img = torch.randn(B, IH, IW, C, device=dev).bfloat16()
    out = torch.empty(Mo, KH * KW * C, device=dev)
    t = timeit(lambda: L.conv_dw_tn(a, img, out, OH, OW, KH, KW, S, P))
    print('%-14s dW %3d x %5d, K %6d: %.1f us (%.0f TF)' % (name, Mo, KH * KW * C, B * OH * OW, t, 2.0 * B * OH * OW * Mo * KH * KW * C / t / 1e6), flush=True)
