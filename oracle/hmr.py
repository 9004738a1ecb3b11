"""Oracle restatement of BASELINE config #1 (plumbing case): ResNet-50 trunk + SPIN/HMR iterative regressor.

TEST INFRASTRUCTURE ONLY.  Follows models/hmr.py:164-267 (HMR: torchvision-style R50, AvgPool2d(7), 3-iteration FC loop on
the 6-D pose state, rot6d_to_rotmat at the end); the same trunk is models/pose_resnet.py:200-217 in global_mode
([B,2048,7,7] map + [B,2048] pooled feature).
"""
import torch
import torch.nn.functional as F

from . import geometry as G
from .whmr import resnet50_features


def pose_resnet_global(sd, x, prefix=''):
    """pose_resnet.py:200-217 (global_mode=True): (s_feat [B,2048,H/32,W/32], g_feat [B,2048])."""
    f = resnet50_features(sd, x, prefix)
    g = F.avg_pool2d(f, 7, stride=1)
    return f, g.view(g.shape[0], -1)


def hmr_forward(sd, x, n_iter=3, prefix=''):
    """hmr.py:232-267 -> (pred_rotmat [B,24,3,3], pred_shape [B,10], pred_cam [B,3])."""
    p = prefix
    B = x.shape[0]
    _, xf = pose_resnet_global(sd, x, p)
    pose = sd[p + 'init_pose'].expand(B, -1)
    shape = sd[p + 'init_shape'].expand(B, -1)
    cam = sd[p + 'init_cam'].expand(B, -1)
    for _ in range(n_iter):
        xc = torch.cat([xf, pose, shape, cam], 1)
        xc = F.linear(xc, sd[p + 'fc1.weight'], sd[p + 'fc1.bias'])
        xc = F.linear(xc, sd[p + 'fc2.weight'], sd[p + 'fc2.bias'])
        pose = F.linear(xc, sd[p + 'decpose.weight'], sd[p + 'decpose.bias']) + pose
        shape = F.linear(xc, sd[p + 'decshape.weight'], sd[p + 'decshape.bias']) + shape
        cam = F.linear(xc, sd[p + 'deccam.weight'], sd[p + 'deccam.bias']) + cam
    return G.rot6d_to_rotmat(pose).view(B, 24, 3, 3), shape, cam
