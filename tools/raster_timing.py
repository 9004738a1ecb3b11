"""IUV rasteriser timing (csrc/rasterize.hip) at the training step's size: 64 meshes, 13774 DensePose faces, 128 x 128.
   python tools/raster_timing.py      -> synthetic posed mesh (random skinning: large triangles) and the rigid rest-pose mesh (small triangles)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whmr_amd.utils import synth
from whmr_amd.utils.renderer import IUV_Renderer
dev = torch.device('cuda:0')
assets = synth.make_assets(0)
dp = synth.make_densepose_tables(0, assets)
mk = IUV_Renderer(orig_size=(256, 256), output_size=(128, 128), dp=dp)
B = 64
cam = torch.tensor([[0.9, 0.0, 0.0]], device=dev).expand(B, -1).contiguous()
vt = assets['smpl']['v_template'].to(dev)
g = torch.Generator().manual_seed(0)
rest = (vt[None] + 0.002 * torch.randn(B, 6890, 3, generator=g).to(dev)).contiguous()
scattered = (0.3 * torch.randn(B, 6890, 3, generator=g)).to(dev).contiguous()
def timeit(fn, n=20, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, v in (('rest pose + 2 mm noise (small faces)', rest), ('scattered vertices (every face large)', scattered)):
    img = mk.verts2iuvimg(v, cam)
    cov = (img[:, 0] > 0).float().mean().item()
    print('%-40s %8.1f us / call   coverage %.2f' % (name, timeit(lambda: mk.verts2iuvimg(v, cam)), cov))
