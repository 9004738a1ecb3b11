"""Round 6 probe: does the camera-calibration branch overlap the backbone when it is replayed from its OWN HIP graph on a side stream (two graph launches
per step) instead of being one branch of the single captured graph?  Timing only: the main graph gets a constant camera rotation (its dependency on the
side graph is left out on purpose -- an upper bound of what a proper three-graph split could reach).   python tools/r6_cam_graphs.py [bf16|bf16x3]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whmr_amd.graph import GraphedForward
from whmr_amd.models import whmr_net
from whmr_amd.utils import synth

dev = torch.device('cuda:0')
numerics = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
B = 64
assets = synth.make_assets(0)
sd = synth.make_state_dict(0, assets)
m = whmr_net(None, assets=assets, numerics=numerics)
m.load_state_dict(sd, strict=True)
m = m.to(dev).eval()
inp = {k: v.to(dev) for k, v in synth.make_inputs(B, 7).items()}
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
    one = GraphedForward(m, *a, full_x=full)
    none = GraphedForward(m, *a)
    eye = torch.eye(3, device=dev).unsqueeze(0).expand(B, -1, -1).contiguous()
    main_only = GraphedForward(m, *a, cam_rotmat=eye)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            m._camera(full, None, B, dev)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gcam = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gcam):
        cam_out = m._camera(full, None, B, dev)

    def two_graphs():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            gcam.replay()
        main_only.graph.replay()
        main.wait_stream(side)

    def cam_then_main():
        gcam.replay()
        main_only.graph.replay()

    for rnd in range(3):
        print('%s round %d: one graph (camera = a branch) %.3f ms | no camera at all %.3f | camera graph alone %.3f | camera graph on a side stream + main graph %.3f | '
              'camera graph THEN main graph on one stream %.3f' % (numerics, rnd, timeit(lambda: one.graph.replay()), timeit(lambda: none.graph.replay()),
                                                                  timeit(lambda: gcam.replay()), timeit(two_graphs), timeit(cam_then_main)), flush=True)
