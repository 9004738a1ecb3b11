"""End-to-end per-frame latency of the demo pipeline on the device: upload -> person crops -> cam_model (once) -> W-HMR forward."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from whmr_amd.utils import synth
from whmr_amd.models import whmr_net
from whmr_amd.demo import infer_frame, prepare_frame, FramePipeline
dev = torch.device('cuda:0')
assets = synth.make_assets(0); sd = synth.make_state_dict(0, assets)
m = whmr_net(None, assets=assets, numerics='bf16'); m.load_state_dict(sd, strict=False); m = m.to(dev).eval()
rng = np.random.default_rng(0)
frame = rng.integers(0, 256, size=(720, 1280, 3), dtype=np.uint8)
for n in (1, 4, 16, 64):
    dets = [(float(rng.uniform(200, 1080)), float(rng.uniform(150, 570)), s, s) for s in rng.uniform(120, 400, size=n)]
    def run():
        f = torch.from_numpy(frame).to(dev, non_blocking=True)          # host -> device upload of the 2.8 MB frame
        return infer_frame(m, f, dets)
    for _ in range(3): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    f = torch.from_numpy(frame).to(dev)
    for _ in range(2): prepare_frame(f, dets)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): prepare_frame(f, dets)
    torch.cuda.synchronize(); dp = (time.perf_counter() - t0) / 10
    pipe = FramePipeline(m)
    def run_g():
        return pipe(torch.from_numpy(frame).to(dev, non_blocking=True), dets)
    for _ in range(3): run_g()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): run_g()
    torch.cuda.synchronize(); dg = (time.perf_counter() - t0) / 10
    print('720p frame, %2d persons: eager %.2f ms / frame | HIP graph %.2f ms (%.0f frames/s, %.0f persons/s); input preparation alone %.2f ms'
          % (n, dt * 1e3, dg * 1e3, 1 / dg, n / dg, dp * 1e3))
