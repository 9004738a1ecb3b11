"""bf16x3 blocked GEMM (three MFMAs per product): isolated launches at M = 12544, N = 2304 / 768, K = 768 / 1536 / 3072 -- the slope over K is the
main loop (issued flops = 3 x 2MNK), the intercept prologue + epilogue.  HIP-graph timed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from whmr_amd import _lib as L
dev = torch.device('cuda:0')
M = 12544


def timeit(fn, reps=20):
    g = torch.cuda.CUDAGraph()
    fn(); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


for N, epi in ((2304, 0), (768, 2)):
    ts = {}
    for K in (768, 1536, 3072):
        mk = lambda r, s: (torch.randn(r // 32, K // 8, 32, 8, device=dev) * s).bfloat16()
        a, al, w, wl = mk(M, 0.5), mk(M, 0.002), mk(N, 0.05), mk(N, 0.0002)
        bias = torch.randn(N, device=dev)
        if epi >= 2:
            out = torch.zeros(M // 32, N // 4, 32, 4, device=dev); res = torch.randn_like(out); ol = None
        else:
            out = torch.zeros(M // 32, N // 8, 32, 8, device=dev, dtype=torch.bfloat16); ol = torch.zeros_like(out); res = None
        ts[K] = timeit(lambda: L.gemm_blk(a, w, out, M, bias=bias, epi=epi, res=res, a_lo=al, w_lo=wl, out_lo=ol))
        tb = timeit(lambda: L.gemm_blk(a, w, out, M, bias=bias, epi=epi, res=res)) if epi >= 2 else float('nan')
        print('N %4d epi %d K %4d: bf16x3 %6.1f us (issued %4.0f TF/s)   bf16 %6.1f us' % (N, epi, K, ts[K], 3 * 2.0 * M * N * K / ts[K] / 1e6, tb), flush=True)
    sl = (ts[3072] - ts[768]) / (3072 - 768)
    print('   slope %.4f us per K -> main loop issues %.0f TF/s; intercept %.1f us' % (sl, 3 * 2.0 * M * N / sl / 1e6, ts[768] - sl * 768), flush=True)
