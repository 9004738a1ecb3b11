"""Camera-calibration head (models/cam_model.py:24-81 + utils/cam_utils.py:114-145 of the reference).

Not a hand-written-kernel component this round: SURVEY 2.1 row 7 / 8(f) N1 keep the ResNet-50 on PyTorch-ROCm (MIOpen
convolutions on the same HIP device).  Module/parameter names follow torchvision's ResNet so that
``cam_model.backbone.*`` checkpoint keys load.  The tiny post-processing (softargmax over 256 bins -> angles ->
euler -> rotation matrix) is a handful of [B,256] tensor ops.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

VFOV_RANGE = (0.2617, 2.1)        # utils/cam_utils.py:56
PITCH_RANGE = (-0.6, 0.6)         # utils/cam_utils.py:38
ROLL_RANGE = (-0.6, 0.6)          # utils/cam_utils.py:139


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return F.relu(y + (x if self.downsample is None else self.downsample(x)))


class ResNet50(nn.Module):
    """PARE's resnet50: torchvision layout, forward returns the layer4 feature map."""

    def __init__(self, pretrained=False):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        inpl = 64
        for li, (n, planes) in enumerate(zip([3, 4, 6, 3], [64, 128, 256, 512])):
            blocks = []
            for bi in range(n):
                stride = (1 if li == 0 else 2) if bi == 0 else 1
                down = None
                if bi == 0:
                    down = nn.Sequential(nn.Conv2d(inpl, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))
                blocks.append(Bottleneck(inpl, planes, stride, down))
                inpl = planes * 4
            setattr(self, 'layer%d' % (li + 1), nn.Sequential(*blocks))

    def forward(self, x):
        x = F.max_pool2d(F.relu(self.bn1(self.conv1(x))), 3, 2, 1)
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))


def resnet50(pretrained=False):
    return ResNet50()


class CameraRegressorNetwork(nn.Module):
    def __init__(self, backbone='resnet50', num_fc_layers=1, num_fc_channels=1024, num_out_channels=256):
        super().__init__()
        assert backbone == 'resnet50' and num_fc_layers == 1
        self.backbone = resnet50()
        self.num_out_channels = num_out_channels
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        for n in ('fc_vfov', 'fc_pitch', 'fc_roll'):
            fc = nn.Linear(2048, num_out_channels)
            nn.init.normal_(fc.weight, mean=0, std=0.01)
            nn.init.constant_(fc.bias, 0)
            setattr(self, n, fc)

    def forward(self, images):
        x = torch.flatten(self.avgpool(self.backbone(images)), 1)
        return [self.fc_vfov(x), self.fc_pitch(x), self.fc_roll(x)], x


def softargmax1d(logits):
    """pare softargmax1d(normalize_keypoints=True) [3P, restated]: E[bin index] under softmax, mapped to [-1,1]."""
    D = logits.shape[-1]
    w = F.softmax(logits, dim=-1)
    idx = (w * torch.arange(D, dtype=logits.dtype, device=logits.device)).sum(-1)
    return idx / (D - 1) * 2 - 1


def soft_idx_to_angle(soft_idx, lo, hi):
    return (hi - lo) * ((soft_idx + 1) / 2) + lo


@torch.no_grad()
def convert_preds_to_angles(pred_vfov, pred_pitch, pred_roll, loss_type='softargmax_l2', **kw):
    """utils/cam_utils.py:121-145, softargmax branch."""
    assert loss_type in ('softargmax_l2', 'softargmax_biased_l2')
    return (soft_idx_to_angle(softargmax1d(pred_vfov), *VFOV_RANGE), soft_idx_to_angle(softargmax1d(pred_pitch), *PITCH_RANGE),
            soft_idx_to_angle(softargmax1d(pred_roll), *ROLL_RANGE))


def batch_euler2matrix(r):
    """pare.utils.geometry.batch_euler2matrix [3P, restated]: euler (x, y, z) -> quaternion -> rotation matrix."""
    h = r * 0.5
    cx, cy, cz = torch.cos(h[:, 0]), torch.cos(h[:, 1]), torch.cos(h[:, 2])
    sx, sy, sz = torch.sin(h[:, 0]), torch.sin(h[:, 1]), torch.sin(h[:, 2])
    q = torch.stack([cx * cy * cz + sx * sy * sz, sx * cy * cz - cx * sy * sz,
                     cx * sy * cz + sx * cy * sz, cx * cy * sz - sx * sy * cz], dim=1)
    q = q / q.norm(dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    w2, x2, y2, z2 = w * w, x * x, y * y, z * z
    wx, wy, wz, xy, xz, yz = w * x, w * y, w * z, x * y, x * z, y * z
    return torch.stack([w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz, 2 * wz + 2 * xy, w2 - x2 + y2 - z2,
                        2 * yz - 2 * wx, 2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2], dim=1).view(-1, 3, 3)
