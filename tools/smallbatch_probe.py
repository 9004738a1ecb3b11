"""Small-batch latency: ViT-B 256x192 forward under a HIP graph, blocked-layout path vs row-major path, and the full W-HMR forward (no full frame)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whmr_amd.models.pose_vit import ViT
from whmr_amd.graph import GraphedForward
dev = torch.device('cuda:0')
def bench(g, *a, n=40):
    for _ in range(5): g(*a)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): g(*a)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
m = ViT(img_size=(256, 192), qkv_bias=True, numerics='bf16').to(dev).eval()
for B in (1, 2, 4, 8, 16):
    x = torch.randn(B, 3, 256, 192, device=dev)
    res = []
    for blocked in (True, False):
        m.blocked = blocked
        g = GraphedForward(m, x)
        res.append(min(bench(g, x) for _ in range(2)))
    print('B=%2d ViT forward (HIP graph): blocked %.3f ms   row-major %.3f ms' % (B, res[0], res[1]), flush=True)
m.blocked = True
if len(sys.argv) > 1 and sys.argv[1] == 'full':
    from whmr_amd.utils import synth
    from whmr_amd.models import whmr_net
    assets = synth.make_assets(0)
    sd = synth.make_state_dict(0, assets)
    w = whmr_net(None, assets=assets, numerics='bf16')
    w.load_state_dict(sd, strict=False)
    w = w.to(dev).eval()
    for B in (1, 8, 64):
        inp = {k: v.to(dev) for k, v in synth.make_inputs(B, 0).items()}
        a = (inp['x'], None, inp['center'], inp['scale'], inp['bbox_height'], inp['orig_shape'], inp['bbox_info'])
        r = []
        for blocked in (True, False):
            w.feature_extractor.backbone.blocked = blocked
            g = GraphedForward(w, *a)
            r.append(min(bench(g, *a, n=20) for _ in range(2)))
        print('B=%2d full W-HMR forward (HIP graph, no full frame): blocked %.3f ms   row-major %.3f ms' % (B, r[0], r[1]), flush=True)
