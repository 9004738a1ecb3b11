"""Achievable HBM write / read / copy bandwidth on this box for the buffer sizes of the GEMM epilogues (HIP-graph timed)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device('cuda:0')
def timeit(fn, n=20, w=3):
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
for mb in (19.3, 38.5, 57.8, 77.1, 403.0):
    n = int(mb * 1e6 / 4)
    a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
    big = [torch.empty(n, device=dev) for _ in range(8)]          # rotate destinations so nothing stays cache-resident
    i = [0]
    def fill():
        big[i[0] % 8].fill_(1.0); i[0] += 1
    def copy():
        big[i[0] % 8].copy_(big[(i[0] + 4) % 8]); i[0] += 1
    def read():
        big[i[0] % 8].sum(); i[0] += 1
    tf, tc, tr = timeit(fill), timeit(copy), timeit(read)
    print('%6.1f MB: fill %6.1f us (%.2f TB/s)  copy %6.1f us (%.2f TB/s r+w)  sum-read %6.1f us (%.2f TB/s)' % (
        mb, tf, mb / tf, tc, 2 * mb / tc, tr, mb / tr), flush=True)
