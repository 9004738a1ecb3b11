"""Forward blocked GEMM, 256 x 256 tile: four waves (tile 0x144) vs eight waves (0x44), isolated launches at M = 12544, per N and epilogue, at
K = 768 / 1536 / 3072 -- the slope over K is the main loop, the intercept prologue + epilogue.  HIP-graph timed (20 launches per replay)."""
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


for N, epi in ((2304, L.EPI_BF16), (3072, 1), (768, 2)):
    for K in (768, 1536, 3072):
        a = (torch.randn(M // 32, K // 8, 32, 8, device=dev) * 0.5).bfloat16()
        w = (torch.randn(N // 32, K // 8, 32, 8, device=dev) * 0.05).bfloat16()
        bias = torch.randn(N, device=dev)
        if epi >= 2:
            out = torch.zeros(M // 32, N // 4, 32, 4, device=dev)
            res = torch.randn_like(out)
        else:
            out = torch.zeros(M // 32, N // 8, 32, 8, device=dev, dtype=torch.bfloat16)
            res = None
        t = {}
        for tile in (0x44, 0x144, 0x44, 0x144):
            t.setdefault(tile, []).append(timeit(lambda: L.gemm_blk(a, w, out, M, bias=bias, epi=epi, res=res, tile=tile)))
        o8 = out.clone(); L.gemm_blk(a, w, o8, M, bias=bias, epi=epi, res=res, tile=0x44)
        o4 = out.clone(); L.gemm_blk(a, w, o4, M, bias=bias, epi=epi, res=res, tile=0x144)
        print('N %4d epi %d K %4d: 8 waves %6.1f us  4 waves %6.1f us   same bits: %s' % (N, epi, K, min(t[0x44]), min(t[0x144]), torch.equal(o8, o4)), flush=True)
