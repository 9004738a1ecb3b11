"""BASELINE config #1 (plumbing case): SPIN/HMR regressor on a ResNet-50 trunk -- models/hmr.py:164-278 of the reference.

``hmr(smpl_mean_params, pretrained)`` -> module whose ``forward(x, init_pose, init_shape, init_cam, n_iter=3)`` returns
``(pred_rotmat [B,24,3,3], pred_shape [B,10], pred_cam [B,3])``; state_dict keys as in the reference (conv1/bn1/layer1-4,
fc1, fc2, decpose, decshape, deccam, init_*).  The convolutional trunk runs on the same HIP NHWC implicit-GEMM kernels as the camera model
(``cam_model.fold_resnet50`` / ``run_resnet50``; ``numerics`` 'fp32' by default: the 1e-4 parity mode), the iterative FC loop on the fp32
split-K GEMM kernel and the final rot6d_to_rotmat on the geometry kernel.  Device tensors only.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib as L
from ..utils.geometry import rot6d_to_rotmat
from .cam_model import ResNet50, fold_resnet50, run_resnet50


class HMR(ResNet50):
    def __init__(self, smpl_mean_params, assets=None):
        super().__init__()
        npose = 24 * 6
        self.avgpool = nn.AvgPool2d(7, stride=1)
        self.fc1 = nn.Linear(2048 + npose + 13, 1024)
        self.drop1 = nn.Dropout()
        self.fc2 = nn.Linear(1024, 1024)
        self.drop2 = nn.Dropout()
        self.decpose = nn.Linear(1024, npose)
        self.decshape = nn.Linear(1024, 10)
        self.deccam = nn.Linear(1024, 3)
        for m in (self.decpose, self.decshape, self.deccam):
            nn.init.xavier_uniform_(m.weight, gain=0.01)
        mp = assets['mean_params'] if assets is not None else np.load(smpl_mean_params)
        self.register_buffer('init_pose', torch.from_numpy(np.asarray(mp['pose'], dtype=np.float32)).unsqueeze(0))
        self.register_buffer('init_shape', torch.from_numpy(np.asarray(mp['shape'], dtype=np.float32)).unsqueeze(0))
        self.register_buffer('init_cam', torch.from_numpy(np.asarray(mp['cam'], dtype=np.float32)).unsqueeze(0))
        self.numerics = 'fp32'
        self._prep = None
        self.eval()

    @torch.no_grad()
    def features(self, x):
        """pose_resnet.py:200-217 global_mode view of the same trunk: (feature map [B,2048,H/32,W/32] as an NCHW view, pooled feature [B,2048])."""
        dt = torch.bfloat16 if self.numerics == 'bf16' else torch.float32
        ver = tuple(t._version for t in list(self.parameters()) + list(self.buffers())) + (self.numerics, str(x.device))
        if self._prep is None or self._prep[0] != ver:
            self._prep = (ver, fold_resnet50(self, dt))
        f = run_resnet50(self._prep[1], x, dt)                                     # NHWC
        assert f.shape[1] == 7 and f.shape[2] == 7, 'AvgPool2d(7) of hmr.py:177 expects a 224 x 224 crop'
        return f.permute(0, 3, 1, 2), L.avgpool_nhwc(f)

    @torch.no_grad()
    def forward(self, x, init_pose=None, init_shape=None, init_cam=None, n_iter=3):
        if not x.is_cuda:
            raise RuntimeError('whmr_amd.HMR runs on a HIP device only (no CPU fallback)')
        B, dev = x.shape[0], x.device
        _, xf = self.features(x)
        state = torch.empty(B, 2048 + 157, dtype=torch.float32, device=dev)       # [xf | pose 144 | shape 10 | cam 3]
        state[:, :2048] = xf
        state[:, 2048:2192] = self.init_pose.expand(B, -1) if init_pose is None else init_pose
        state[:, 2192:2202] = self.init_shape.expand(B, -1) if init_shape is None else init_shape
        state[:, 2202:] = self.init_cam.expand(B, -1) if init_cam is None else init_cam
        wh = torch.cat([self.decpose.weight, self.decshape.weight, self.deccam.weight], 0).detach().contiguous()
        bh = torch.cat([self.decpose.bias, self.decshape.bias, self.deccam.bias], 0).detach().contiguous()
        h1 = torch.empty(B, 1024, dtype=torch.float32, device=dev)
        h2 = torch.empty(B, 1024, dtype=torch.float32, device=dev)
        for _ in range(n_iter):                                                     # hmr.py:255-263
            L.gemm(state, self.fc1.weight.detach(), h1, bias=self.fc1.bias.detach())
            L.gemm(h1, self.fc2.weight.detach(), h2, bias=self.fc2.bias.detach())
            L.gemm(h2, wh, state[:, 2048:], bias=bh, residual=state[:, 2048:])      # residual update of (pose, shape, cam)
        pose6 = state[:, 2048:2192].contiguous()
        return rot6d_to_rotmat(pose6).view(B, 24, 3, 3), state[:, 2192:2202].contiguous(), state[:, 2202:].contiguous()


def hmr(smpl_mean_params, pretrained=True, **kwargs):
    """models/hmr.py:269-278 (ImageNet weights are loaded by the caller when available; none ship with this repo)."""
    return HMR(smpl_mean_params, **kwargs)
