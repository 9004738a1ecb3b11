"""Weight-gradient product dW = dY^T . X (TN kernel + its split-K reduce) at the training shapes (256x192 crops, batch 64: 12288 tokens): time of the
whole call against the number of K slices.  Fewer slices = less partial-sum traffic but fewer workgroups."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whmr_amd import _lib as L
dev = torch.device('cuda:0')
def timeit(fn, n=20, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
for n_out, k_in in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
    dy = torch.randn(M, n_out, device=dev).bfloat16(); x = torch.randn(M, k_in, device=dev).bfloat16()
    dw = torch.empty(n_out, k_in, device=dev); db = torch.empty(n_out, device=dev)
    res = []
    for sp in (0, 2, 3, 4, 5, 6, 7, 8, 9, 12, 16, 20, 28):
        t = min(timeit(lambda: L.gemm_tn(dy, x, dw, splits=sp, db=db)) for _ in range(2))
        res.append('%d: %.1f' % (sp, t))
    print('dW %4d x %4d (us by slices; 0 = chooser): %s' % (n_out, k_in, '  '.join(res)), flush=True)
