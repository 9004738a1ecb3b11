import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whmr_amd import _lib as L
dev = torch.device('cuda:0')
M = 12544
def timeit(fn, n=30, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for N, K in ((2304, 768), (768, 3072), (4096, 4096)):
    a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    f = 2.0 * M * N * K / 1e6
    r = {}
    for name, probe in (('full', 0), ('mainloop', -12345), ('epi-nostore', -2001), ('epi-nolds', -2002)):
        t = timeit(lambda: L.gemm(a, w, out, tile=257, res_row_mod=probe)); r[name] = t
    print('N=%d K=%d tile 256x256x64: ' % (N, K) + ' | '.join('%s %.1f us %.0f TF' % (k, v, f / v) for k, v in r.items()), '| hipBLASLt %.0f TF' % (f / timeit(lambda: torch.matmul(a, w.t()))))
