"""rocprofv3 target: cam_model (HIP NHWC ResNet-50) on N 600x800 frames.  usage: cam_profile.py [n_images] [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whmr_amd.utils import synth
from whmr_amd.models.cam_model import CameraRegressorNetwork
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device('cuda:0')
sd = synth.make_state_dict(0, synth.make_assets(0))
m = CameraRegressorNetwork()
m.load_state_dict({k[len('cam_model.'):]: v for k, v in sd.items() if k.startswith('cam_model.')}, strict=True)
m = m.to(dev).eval()
x = torch.randn(n, 3, 600, 800, device=dev)
for _ in range(iters):
    m(x)
torch.cuda.synchronize()
