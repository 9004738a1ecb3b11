"""Round 6: an UPPER BOUND for a cross-launch (persistent, arrive-counter) form of the ViT's fc1 -> fc2 pair, measured without
building it: what does the hardware's own dynamic scheduling gain when the tiles of one launch may fill the ragged last round and the launch boundary of
another?  Two INDEPENDENT fc1 -> fc2 chains (own buffers, ViT-B shapes, batch 64: M = 12544) are replayed from HIP graphs
  serial    : chain A then chain B on one stream (what the model does with consecutive layers),
  shifted   : chain A on stream 1, chain B on stream 2 started half a chain later (fc2 of A runs beside fc1 of B: every boundary and every ragged
              round of one chain has ready tiles of the other to fill it -- no dependency hand-over, no fence, no poll: the best case of the
              persistent form),
  parallel  : both chains from t = 0 on two streams.
A persistent kernel pays a hand-over (release fence + arrive + poll + acquire: 5-6 us measured on the SMPL pilot, profiles/r05_experiments_that_did_not_pay.txt)
on top of whatever 'shifted' gains.   python tools/r6_chain_overlap_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whmr_amd import _lib as L

dev = torch.device('cuda:0')
M, C, Hd = 12544, 768, 3072
g = torch.Generator().manual_seed(0)


def chain_buffers(seed):
    g.manual_seed(seed)
    x = L.to_blocked(torch.randn(M, C, generator=g).bfloat16().to(dev))
    w1 = L.to_blocked((torch.randn(Hd, C, generator=g) / C ** 0.5).bfloat16().to(dev))
    w2 = L.to_blocked((torch.randn(C, Hd, generator=g) / Hd ** 0.5).bfloat16().to(dev))
    b1, b2 = torch.randn(Hd, generator=g).to(dev), torch.randn(C, generator=g).to(dev)
    hid = torch.empty(M // 32, Hd // 8, 32, 8, dtype=torch.bfloat16, device=dev)
    stream = L.to_blocked(torch.randn(M, C, generator=g).to(dev))
    xhat = torch.empty(M // 32, C // 8, 32, 8, dtype=torch.bfloat16, device=dev)
    stats = torch.empty(M * (C // 256) * 2, dtype=torch.float32, device=dev)
    return x, w1, w2, b1, b2, hid, stream, xhat, stats


def run_chain(bufs, layers=1):
    x, w1, w2, b1, b2, hid, stream, xhat, stats = bufs
    for _ in range(layers):
        L.gemm_blk(x, w1, hid, M, bias=b1, epi=L.EPI_BF16_GELU)
        L.gemm_blk(hid, w2, stream, M, bias=b2, epi=L.EPI_F32_RES, res=stream, xhat=xhat, stats_out=stats)


A, B = chain_buffers(1), chain_buffers(2)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
LAYERS = 6


def serial():
    run_chain(A, LAYERS)
    run_chain(B, LAYERS)


def parallel():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        run_chain(A, LAYERS)
    with torch.cuda.stream(s2):
        run_chain(B, LAYERS)
    cur.wait_stream(s1); cur.wait_stream(s2)


def shifted():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    x, w1, w2, b1, b2, hid, stream, xhat, stats = A
    with torch.cuda.stream(s1):
        L.gemm_blk(x, w1, hid, M, bias=b1, epi=L.EPI_BF16_GELU)          # A is half a chain ahead
        ev = torch.cuda.Event(); ev.record(s1)
        L.gemm_blk(hid, w2, stream, M, bias=b2, epi=L.EPI_F32_RES, res=stream, xhat=xhat, stats_out=stats)
        run_chain(A, LAYERS - 1)
    s2.wait_event(ev)
    with torch.cuda.stream(s2):
        run_chain(B, LAYERS)
    cur.wait_stream(s1); cur.wait_stream(s2)


def graphed(fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        fn()
    return gr


def timeit(gr, n=20, w=5):
    for _ in range(w):
        gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


gs, gp, gh = graphed(serial), graphed(parallel), graphed(shifted)
pairs = 2 * LAYERS
for rnd in range(4):
    ts, tp, th = timeit(gs), timeit(gp), timeit(gh)
    print('round %d: per fc1 + fc2 pair: serial %.1f us | two chains in parallel %.1f us (%+.1f %%) | second chain half a chain behind %.1f us (%+.1f %%)'
          % (rnd, ts / pairs, tp / pairs, 100 * (tp / ts - 1), th / pairs, 100 * (th / ts - 1)), flush=True)
