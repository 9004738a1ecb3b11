"""Round 6 pilot A/B: ViT-B/16 224^2 batch 64 bf16 forward with the fc1 -> fc2 pair of every layer as ONE persistent launch (WHMR_BLK_CHAIN) against
the two launches, interleaved on one box (eager and HIP-graph replay) + the pair isolated.   python tools/r6_chain_ab.py"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whmr_amd import _lib as L
from whmr_amd.models.pose_vit import ViT

dev = torch.device('cuda:0')
m = ViT(img_size=224, qkv_bias=True, numerics='bf16').to(dev).eval()
x = torch.randn(64, 3, 224, 224, device=dev)


def timeit(fn, n=30, w=10):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    m.chain_mlp = False
    ref = m(x).clone()
    m.chain_mlp = True
    out = m(x)
    torch.cuda.synchronize()
    print('chained forward equals the two-launch forward bit for bit:', torch.equal(out, ref), '| device error flag', L.chain_error(dev))
    graphs = {}
    for ch in (False, True):
        m.chain_mlp = ch
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                m(x)
        torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            m(x)
        graphs[ch] = g
    for rnd in range(4):
        res = []
        for ch in (False, True):
            m.chain_mlp = ch
            res.append((timeit(lambda: m(x)), timeit(graphs[ch].replay)))
        print('round %d: two launches %.3f ms eager / %.3f graph | chained %.3f ms eager / %.3f graph  (%+.2f %% / %+.2f %%)'
              % (rnd, res[0][0], res[0][1], res[1][0], res[1][1], 100 * (res[1][0] / res[0][0] - 1), 100 * (res[1][1] / res[0][1] - 1)), flush=True)
    print('device error flag after the runs:', L.chain_error(dev))
    # the pair isolated (ViT-B shapes)
    M, C_, Hd = 12544, 768, 3072
    h = L.to_blocked(torch.randn(M, C_, device=dev).bfloat16())
    w1 = L.to_blocked((torch.randn(Hd, C_, device=dev) / C_ ** 0.5).bfloat16()); w2 = L.to_blocked((torch.randn(C_, Hd, device=dev) / Hd ** 0.5).bfloat16())
    b1, b2 = torch.randn(Hd, device=dev), torch.randn(C_, device=dev)
    hid = torch.empty(M // 32, Hd // 8, 32, 8, dtype=torch.bfloat16, device=dev)
    t = L.to_blocked(torch.randn(M, C_, device=dev))
    xh = torch.empty_like(h); st = torch.empty(M * 3 * 2, device=dev)
    fc1 = dict(a=h, w=w1, out=hid, M=M, bias=b1, epi=L.EPI_BF16_GELU)
    fc2 = dict(a=hid, w=w2, out=t, M=M, bias=b2, epi=L.EPI_F32_RES, res=t, xhat=xh, stats_out=st)

    def two():
        L.gemm_blk(**fc1); L.gemm_blk(**fc2)

    def one():
        assert L.gemm_blk_chain(fc1, fc2)
    gs = {}
    for name, fn in (('two', two), ('one', one)):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(10):
                fn()
        gs[name] = g
    for rnd in range(3):
        print('isolated pair: two launches %.1f us | one chained launch %.1f us' % (timeit(gs['two'].replay) * 100, timeit(gs['one'].replay) * 100), flush=True)
    # lab: the same chained launch with WORKGROUP-scope fences (not correct across the XCD-private L2s) -- what the agent-scope fences cost
    L.lib().whmr_gemm_blk_set_tile(4, 1)
    one(); torch.cuda.synchronize()
    glab = torch.cuda.CUDAGraph()
    with torch.cuda.graph(glab):
        for _ in range(10):
            one()
    ref_t = t.clone()
    for rnd in range(3):
        print('isolated pair, LAB (workgroup-scope fences, results not guaranteed): one chained launch %.1f us' % (timeit(glab.replay) * 100), flush=True)
    L.lib().whmr_gemm_blk_set_tile(4, 0)
    print('device error flag at the end:', L.chain_error(dev))
