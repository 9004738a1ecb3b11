"""IUV image -> DensePose target maps (utils/iuvmap.py:67-110 of the reference, uv_rois=None branch): the ground-truth side of the AUX
supervision (core/trainer.py:464).  O(B x 25 x H x W) element-wise tensor arithmetic on the device, not a path kernel."""
import torch

INDEX2MASK = ((0,), (1, 2), (3,), (4,), (5,), (6,), (7, 9), (8, 10), (11, 13), (12, 14), (15, 17), (16, 18), (19, 21), (20, 22), (23, 24))


def iuv_img2map(uvimages):
    """uvimages [B, 3, H, W] = (I / 24, U, V) -> (U [B,25,H,W], V [B,25,H,W], Index_UV [B,25,H,W], Ann_Index [B,15,H,W]).
    Channel i of Index_UV is the indicator of part i (i = 0: background); U / V are the coordinates masked by it."""
    part = torch.round(uvimages[:, 0] * 24)
    ids = torch.arange(25, device=uvimages.device, dtype=part.dtype).view(1, 25, 1, 1)
    index_uv = (part.unsqueeze(1) == ids).to(uvimages.dtype)
    u = index_uv * uvimages[:, 1:2]
    v = index_uv * uvimages[:, 2:3]
    ann = torch.stack([sum(index_uv[:, j] for j in grp) for grp in INDEX2MASK], dim=1)
    return u, v, index_uv, ann
