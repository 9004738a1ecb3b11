"""fp32 skinny GEMMs of the regressor loop (M = batch): time + achieved weight-streaming bandwidth."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whmr_amd import _lib as L
dev = torch.device('cuda:0')
M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
def timeit(fn, n=50, w=5):
    """GPU time per call: n calls captured in one HIP graph (no host launch cost in the number)"""
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(w): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, N, K in (('reg.fc1', 1024, 2149 + 24 * 6 + 13), ('reg.fc2', 1024, 1024), ('decpose', 144, 1024), ('go.fc1', 2048, 2164),
                   ('go.fc2', 2048, 2048), ('posedirs', 20670, 207)):
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / math.sqrt(K)
    bias = torch.randn(N, device=dev); out = torch.empty(M, N, device=dev)
    us = timeit(lambda: L.gemm(a, w, out, bias=bias))
    ref = timeit(lambda: torch.addmm(bias, a, w.t(), out=out))
    print('%-9s M=%d N=%5d K=%5d: %6.1f us  %5.2f TB/s of weights | torch addmm %6.1f us' % (name, M, N, K, us, N * K * 4 / us / 1e6, ref), flush=True)
