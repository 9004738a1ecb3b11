"""cam_model timings: HIP NHWC ResNet-50 vs the same module on torch/MIOpen, per-GEMM breakdown."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whmr_amd.utils import synth
from whmr_amd import _lib as L
from whmr_amd.models import whmr_net
from whmr_amd.graph import GraphedForward
dev = torch.device('cuda:0')
assets = synth.make_assets(0); sd = synth.make_state_dict(0, assets)
m = whmr_net(None, assets=assets, numerics='bf16'); m.load_state_dict(sd, strict=False); m = m.to(dev).eval()
B = 64
inp = {k: v.to(dev) for k, v in synth.make_inputs(B, 0).items()}
args = (inp['x'], None, inp['center'], inp['scale'], inp['bbox_height'], inp['orig_shape'], inp['bbox_info'])
full1 = torch.randn(1, 3, 600, 800, device=dev)
def torch_forward(cm, x):
    """the same ResNet-50 on torch / MIOpen (timing reference only)"""
    import torch.nn.functional as F
    bb = cm.backbone
    bn = lambda y, b: F.batch_norm(y, b.running_mean, b.running_var, b.weight, b.bias, False, 0.0, b.eps)
    y = F.max_pool2d(F.relu(bn(bb.conv1(x), bb.bn1)), 3, 2, 1)
    for li in range(1, 5):
        for blk in getattr(bb, 'layer%d' % li):
            z = F.relu(bn(blk.conv1(y), blk.bn1)); z = F.relu(bn(blk.conv2(z), blk.bn2)); z = bn(blk.conv3(z), blk.bn3)
            y = F.relu(z + (y if blk.downsample is None else bn(blk.downsample[0](y), blk.downsample[1])))
    f = y.mean((2, 3))
    return cm.fc_vfov(f), cm.fc_pitch(f), cm.fc_roll(f)


def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print('no cam_model: %.2f ms' % t(lambda: m(*args)))
print('cam_model hoisted (1 full image 600x800 per batch): %.2f ms' % t(lambda: m(*args, full_x=full1)))
for nimg in (1, 8, 64):
    full = full1.expand(nimg, -1, -1, -1).contiguous()
    for mode in ('bf16', 'fp32'):
        if mode == 'fp32' and nimg == 64: continue
        m.cam_model.numerics = mode
        print('cam_model %2d x 600x800  HIP %s: %.2f ms' % (nimg, mode, t(lambda: m.cam_model(full))))
    m.cam_model.numerics = 'bf16'
    if nimg < 64:
        with torch.no_grad():
            print('cam_model %2d x 600x800  torch/MIOpen fp32: %.2f ms' % (nimg, t(lambda: torch_forward(m.cam_model, full))))
    g = GraphedForward(m.cam_model, full)
    print('cam_model %2d x 600x800  HIP bf16, HIP graph: %.2f ms' % (nimg, t(lambda: g(full))))
full = full1.expand(8, -1, -1, -1).contiguous()
m.cam_model(full)
L.PROFILE = []
m.cam_model(full)
torch.cuda.synchronize()
rows, L.PROFILE = L.PROFILE, None
tot = 0.
for i, (name, fl, e0, e1) in enumerate(rows):
    ms = e0.elapsed_time(e1); tot += ms
    print('%2d %-10s %8.1f us %7.1f TF' % (i, name, ms * 1e3, fl / ms / 1e9))
print('sum of GEMM launches: %.2f ms (8 images)' % tot)
