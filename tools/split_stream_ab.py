"""ViT-B/16 224^2 batch 64: one stream vs the batch split in two halves on two streams (two module copies: the workspaces are per module)."""
import sys, os, time, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whmr_amd.models.pose_vit import ViT
dev = torch.device('cuda:0')
m1 = ViT(img_size=224, qkv_bias=True, numerics='bf16').to(dev).eval()
m2 = copy.deepcopy(m1)
x = torch.randn(64, 3, 224, 224, device=dev)
xa, xb = x[:32].contiguous(), x[32:].contiguous()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def single():
    return m1(x)
def split():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): a = m1(xa)
    with torch.cuda.stream(s2): b = m2(xb)
    cur.wait_stream(s1); cur.wait_stream(s2)
    return a, b
def bench(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    for r in range(3):
        print('one stream %.3f ms   two half-batches on two streams %.3f ms' % (bench(single), bench(split)), flush=True)
    # graph-captured versions (no host launch effects)
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1): o = single()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2): o2 = split()
    for r in range(3):
        print('graph: one stream %.3f ms   two streams %.3f ms' % (bench(g1.replay), bench(g2.replay)), flush=True)
