"""dW = dY^T . X at the ViT-B shapes (12544 tokens): the TN kernel (no operand copies) vs the transposed-copy path (2 transposes + NT GEMM)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whmr_amd import _lib as L
dev = torch.device('cuda:0')
def timeit(fn, n=10, w=2):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = 12544
for n_out, k_in in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
    dy = torch.randn(M, n_out, device=dev).bfloat16(); x = torch.randn(M, k_in, device=dev).bfloat16()
    dw = torch.empty(n_out, k_in, device=dev)
    def old():
        dyt = L.transpose_cast(dy, torch.bfloat16, pad_to=64); xt = L.transpose_cast(x, torch.bfloat16, pad_to=64)
        L.gemm(dyt, xt, dw)
    def old_gemm_only(dyt=L.transpose_cast(dy, torch.bfloat16, pad_to=64), xt=L.transpose_cast(x, torch.bfloat16, pad_to=64)):
        L.gemm(dyt, xt, dw)
    t_old, t_g = timeit(old), timeit(old_gemm_only)
    ref = dw.clone()
    res = []
    for sp in (0, 4, 8, 12):
        t = timeit(lambda: L.gemm_tn(dy, x, dw, splits=sp))
        res.append('splits %d: %.1f us' % (sp, t))
    err = ((dw - ref).abs().max() / ref.abs().max()).item()
    fl = 2.0 * M * n_out * k_in
    print('dW %4d x %4d: transposes + NT GEMM %.1f us (GEMM alone %.1f us = %.0f TF)   TN %s   max-rel diff %.1e'
          % (n_out, k_in, t_old, t_g, fl / t_g / 1e6, '  '.join(res), err), flush=True)
