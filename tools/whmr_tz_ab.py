import sys, os, time
sys.path.insert(0, '/root/repo')
import torch
from whmr_amd.utils import synth
from whmr_amd.models import whmr_net
from whmr_amd.graph import GraphedForward
dev = torch.device('cuda:0')
assets = synth.make_assets(0); sd = synth.make_state_dict(0, assets)
m = whmr_net(None, assets=assets, numerics='bf16'); m.load_state_dict(sd, strict=True); m = m.to(dev).eval()
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for B in (64, 8, 1):
    inp = {k: v.to(dev) for k, v in synth.make_inputs(B, 7).items()}
    a = (inp['x'], None, inp['center'], inp['scale'], inp['bbox_height'], inp['orig_shape'], inp['bbox_info'])
    for rnd in range(2):
        for otz in (False, True):
            m.overlap_tz = otz
            g = GraphedForward(m, *a)
            print('B=%2d no frame, overlap_tz=%-5s: HIP graph %.3f ms' % (B, otz, min(bench(g.graph.replay) for _ in range(2))), flush=True)
