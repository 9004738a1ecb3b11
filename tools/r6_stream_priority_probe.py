"""Round 6 probe: stream PRIORITIES for the side branches of the full forward (camera chain low, heavy deconv / Tz chain high or normal), eager and
replayed from a HIP graph; batch 64.   python tools/r6_stream_priority_probe.py [bf16|bf16x3]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whmr_amd.graph import GraphedForward
from whmr_amd.models import whmr as W
from whmr_amd.models import whmr_net
from whmr_amd.utils import synth

dev = torch.device('cuda:0')
print('stream priority range (least, greatest):', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else 'n/a')
assets = synth.make_assets(0)
sd = synth.make_state_dict(0, assets)
m = whmr_net(None, assets=assets, numerics=sys.argv[1] if len(sys.argv) > 1 else 'bf16')
m.load_state_dict(sd, strict=True)
m = m.to(dev).eval()
inp = {k: v.to(dev) for k, v in synth.make_inputs(64, 7).items()}
a = (inp['x'], None, inp['center'], inp['scale'], inp['bbox_height'], inp['orig_shape'], inp['bbox_info'])
full = torch.randn(1, 3, 600, 800, generator=torch.Generator().manual_seed(11)).to(dev)


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
    ref = {k: v.clone() for k, v in m(*a, full_x=full).items()}
    for rnd in range(2):
        for cam_p, tz_p, main_p in ((0, 0, 0), (1, 0, 0), (2, 0, 0), (1, -1, 0), (0, -1, 0), (1, 0, -1), (1, -1, -1)):
            saved = dict(W._CAM_STREAMS)
            W._CAM_STREAMS[(dev, None)] = torch.cuda.Stream(device=dev, priority=cam_p)
            W._CAM_STREAMS[(dev, 'tz')] = torch.cuda.Stream(device=dev, priority=tz_p)
            main = torch.cuda.Stream(device=dev, priority=main_p)

            def run():
                cur = torch.cuda.current_stream()
                main.wait_stream(cur)
                with torch.cuda.stream(main):
                    out = m(*a, full_x=full)
                cur.wait_stream(main)
                return out
            out = run()
            torch.cuda.synchronize()
            same = all(torch.equal(out[k], ref[k]) for k in ref)
            te = timeit(run)
            with torch.cuda.stream(main):
                g = GraphedForward(m, *a, full_x=full)
            tg = timeit(lambda: g.graph.replay())
            print('round %d: priorities camera %2d, heavy chain %2d, main %2d: eager %.3f ms | graph replay %.3f ms  (same bits %s)'
                  % (rnd, cam_p, tz_p, main_p, te, tg, same), flush=True)
            W._CAM_STREAMS.clear()
            W._CAM_STREAMS.update(saved)
