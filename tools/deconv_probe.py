"""Deconv pyramid GEMMs (whmr.py:488-498 as 4 sub-pixel phases in one launch), B=64: tile sweep, main loop vs full kernel."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whmr_amd import _lib as L
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
def timeit(fn, n=10, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, H, W, Cin in (('deconv1', 16, 12, 768), ('deconv2', 32, 24, 256), ('deconv3', 64, 48, 256)):
    Cout = 256
    x = torch.randn(B, H, W, Cin, device=dev).bfloat16()
    ph = (torch.randn(4, Cout, 4 * Cin, device=dev) / math.sqrt(4 * Cin)).bfloat16()
    shift = torch.randn(Cout, device=dev)
    out = torch.empty(B, 2 * H, 2 * W, Cout, device=dev, dtype=torch.bfloat16)
    kw = dict(conv=dict(IH=H, IW=W, Cin=Cin, OH=H, OW=W, KW=2, SH=1, SW=1, PH=1, PW=1),
              scatter=dict(c_off=0, osb=4 * H * W * Cout, osy=4 * W * Cout, osx=2 * Cout), phases=dict(cy=2 * W * Cout, cx=Cout))
    fl = 2.0 * 4 * B * H * W * Cout * 4 * Cin
    for tile in (None, 257, 259, 192, 320, 128, 64):
        full = timeit(lambda: L.gemm(x, ph, out, bias=shift, act=L.ACT_RELU, tile=tile, **kw))
        main = timeit(lambda: L.gemm(x, ph, out, bias=shift, act=L.ACT_RELU, tile=tile or 257, res_row_mod=-12345, **kw))
        print('%s tile %-4s full %.1f us (%.0f TF)  main-only %.1f us   out %.0f MB' % (name, tile, full, fl / full / 1e6, main, out.numel() * 2 / 1e6), flush=True)
