"""A/B of the cam_model side stream (WHMR.overlap_camera): full W-HMR forward, batch 64 + one hoisted 600x800 frame, HIP-graph replay and eager."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whmr_amd.utils import synth
from whmr_amd.models import whmr_net
from whmr_amd.graph import GraphedForward
dev = torch.device('cuda:0')
assets = synth.make_assets(0)
sd = synth.make_state_dict(0, assets)
m = whmr_net(None, assets=assets, numerics='bf16')
m.load_state_dict(sd, strict=True)
m = m.to(dev).eval()
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for B in (64, 8, 1):
    inp = {k: v.to(dev) for k, v in synth.make_inputs(B, 7).items()}
    a = (inp['x'], None, inp['center'], inp['scale'], inp['bbox_height'], inp['orig_shape'], inp['bbox_info'])
    full = torch.randn(1, 3, 600, 800, generator=torch.Generator().manual_seed(11)).to(dev)
    outs = {}
    for ov, otz in ((False, False), (True, False), (True, True), (False, False), (True, True)):
        m.overlap_camera, m.overlap_tz = ov, otz
        g = GraphedForward(m, *a, full_x=full)
        tg = min(bench(g.graph.replay) for _ in range(2))
        te = bench(lambda: m(*a, full_x=full), n=10)
        outs[(ov, otz)] = {k: v.clone() for k, v in g.out.items()}
        print('B=%2d overlap_camera=%-5s overlap_tz=%-5s  HIP graph %.3f ms   eager %.3f ms' % (B, ov, otz, tg, te), flush=True)
    same = all(torch.equal(outs[(False, False)][k], outs[(True, False)][k]) for k in outs[(False, False)])
    worst = max(((outs[(False, False)][k] - outs[(True, True)][k]).abs().max() / outs[(False, False)][k].abs().max().clamp_min(1e-30)).item() for k in outs[(False, False)])
    print('B=%2d camera side stream bit-identical: %s; with the Tz side stream (finalize kernel instead of the tail) max-rel difference %.1e' % (B, same, worst))
