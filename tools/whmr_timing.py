"""Time the full W-HMR forward (BASELINE config #3) on the box: bf16 mode, B=64, with and without the cam_model image."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whmr_amd.utils import synth
from whmr_amd.models import whmr_net
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
assets = synth.make_assets(0)
sd = synth.make_state_dict(0, assets)
m = whmr_net(None, assets=assets, numerics='bf16')
m.load_state_dict(sd, strict=False)
m = m.to(dev).eval()
inp = {k: v.to(dev) for k, v in synth.make_inputs(B, 0).items()}
args = (inp['x'], None, inp['center'], inp['scale'], inp['bbox_height'], inp['orig_shape'], inp['bbox_info'])
def run(**kw):
    return m(*args, **kw)
for name, kw in (('no full_x (cam_rotmat = I)', {}),):
    for _ in range(3): run(**kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): run(**kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print('WHMR forward B=%d %s: %.2f ms  %.0f img/s' % (B, name, dt * 1e3, B / dt))
from whmr_amd.graph import GraphedForward
for b in (1, 8, B):
    a2 = tuple(t[:b].contiguous() if torch.is_tensor(t) else t for t in args)
    for _ in range(3): m(*a2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): m(*a2)
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / steps
    fast = GraphedForward(m, *a2)
    for _ in range(3): fast(*a2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fast(*a2)
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / steps
    print('B=%d: eager %.2f ms (%.0f img/s) | HIP graph %.2f ms (%.0f img/s)' % (b, te * 1e3, b / te, tg * 1e3, b / tg))
