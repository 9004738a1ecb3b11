"""Camera-calibration head (models/cam_model.py:24-81 + utils/cam_utils.py:114-145 of the reference) -- SURVEY 8(f) N1.

Module/parameter names follow torchvision's ResNet so that ``cam_model.backbone.*`` checkpoint keys load.  On a HIP
device ``CameraRegressorNetwork.forward`` runs the ResNet-50 on libwhmr_hip.so: activations NHWC, eval-mode BatchNorm
folded into the conv weights, every conv an (implicit-)GEMM on the same MFMA kernels as the ViT --

    stem 7x7 s2   im2col (k = ci,ky,kx8; K 168 -> 192) + GEMM + ReLU         [bf16]   /  NHWC gather GEMM          [fp32]
    max-pool      whmr_maxpool_nhwc
    bottleneck    1x1 GEMM+ReLU -> 3x3 gather GEMM+ReLU -> 1x1 GEMM + skip (added BEFORE the ReLU, epi_flags bit 1);
                  projection skip = 1x1 (strided) gather GEMM
    head          whmr_avgpool_nhwc -> one [768,2048] fp32 GEMM for the three fc layers

``numerics``: 'fp32' (default; exact-f32 MFMA), 'bf16x3' (what WHMR's default sets: fp32 maps, every convolution behind the stem as ONE bf16 MFMA
launch on K-concatenated split operands -- fp32-grade logits at ~3x the bf16 cost instead of ~7x) or 'bf16' (bf16 operands / activations, fp32
accumulate: the throughput opt-in).
The nn.Module tree (``Bottleneck`` / ``ResNet50``) only CONTAINS the parameters under torchvision's names; it has no forward of its
own -- tests compare the HIP path with the CPU oracle (oracle/whmr.py::cam_model_forward).  The post-processing (softargmax over 256
bins -> angles -> euler -> rotation matrix) is a handful of [B,256] tensor ops.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib as L

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
        raise RuntimeError('parameter container: the ResNet-50 runs through CameraRegressorNetwork.forward on the HIP kernels')


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
        raise RuntimeError('parameter container: the ResNet-50 runs through CameraRegressorNetwork.forward on the HIP kernels')


def resnet50(pretrained=False):
    return ResNet50()


def fold_resnet50(bb, dt, x3=False):
    """Eval-mode BatchNorm folded into the conv weights of a torchvision-layout ResNet-50 trunk ``bb`` (conv1 / bn1 / layer1-4), weights laid out
    [N, (ky, kx, ci)] in the compute dtype ``dt`` -- the operand set of ``run_resnet50``.  ``x3`` (dt fp32): the bf16x3 numerics -- every conv
    behind the stem gets the K-concatenated split weight [N, (ky, kx), [W_hi | W_hi | W_lo]] (bf16, 3 Cin per tap) that pairs with ``L.split3`` of
    its fp32 input map; the stem (K = 147, not a multiple of the bf16 kernel's K step) stays on the exact-f32 kernel."""
    def fold(conv, bn, stem_cols=False, split=False):
        s = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach().float()
        w = conv.weight.detach().float() * s[:, None, None, None]
        b = (bn.bias - bn.running_mean * s).detach().float().contiguous()
        if stem_cols:                                          # whmr_conv_im2col column order (ci, ky, kx 7->8), K 168 -> 192
            w = F.pad(F.pad(w, (0, 1)).reshape(w.shape[0], -1), (0, 192 - 3 * 7 * 8))
        elif split:
            w = L.split3_weight(w.permute(0, 2, 3, 1).contiguous()).reshape(w.shape[0], -1)       # per tap [hi | hi | lo] over the channels
            return w.contiguous(), b
        else:
            w = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)
        return w.to(dt).contiguous(), b

    if x3:
        assert dt == torch.float32
        prep = {'stem': fold(bb.conv1, bb.bn1), 'blocks': [], 'x3': True}
        for li in range(1, 5):
            for blk in getattr(bb, 'layer%d' % li):
                prep['blocks'].append(dict(c1=fold(blk.conv1, blk.bn1, split=True), c2=fold(blk.conv2, blk.bn2, split=True),
                                           c3=fold(blk.conv3, blk.bn3, split=True), stride=blk.conv2.stride[0],
                                           down=fold(blk.downsample[0], blk.downsample[1], split=True) if blk.downsample is not None else None))
        return prep
    prep = {'stem': fold(bb.conv1, bb.bn1, stem_cols=(dt == torch.bfloat16)), 'blocks': []}
    for li in range(1, 5):
        for blk in getattr(bb, 'layer%d' % li):
            prep['blocks'].append(dict(c1=fold(blk.conv1, blk.bn1), c2=fold(blk.conv2, blk.bn2), c3=fold(blk.conv3, blk.bn3),
                                       stride=blk.conv2.stride[0],
                                       down=fold(blk.downsample[0], blk.downsample[1]) if blk.downsample is not None else None))
    return prep


@torch.no_grad()
def run_resnet50(P, images, dt):
    """NCHW fp32 images -> the layer4 map in NHWC [B, H/32, W/32, 2048] (dtype ``dt``) on the HIP implicit-GEMM kernels."""
    dev = images.device
    B, _, H, W = images.shape
    w, b = P['stem']
    if dt == torch.bfloat16:
        cols, OH, OW = L.conv_im2col(images.float(), 7, 7, 2, 3, 192)
        x = torch.empty(B, OH, OW, 64, dtype=dt, device=dev)
        L.gemm(cols, w, x, bias=b, act=L.ACT_RELU)
    else:
        OH, OW = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        x = torch.empty(B, OH, OW, 64, dtype=dt, device=dev)
        L.gemm(images.float().permute(0, 2, 3, 1).contiguous(), w, x, bias=b, act=L.ACT_RELU,
               conv=dict(IH=H, IW=W, Cin=3, OH=OH, OW=OW, KW=7, SH=2, SW=2, PH=3, PW=3))
    x = L.maxpool_nhwc(x, 3, 2, 1)
    if P.get('x3'):
        return _run_blocks_x3(P, x)
    for blk in P['blocks']:
        _, IH, IW, Cin = x.shape
        s = blk['stride']
        OH, OW = (IH - 1) // s + 1, (IW - 1) // s + 1
        planes = blk['c1'][0].shape[0]
        y1 = torch.empty(B, IH, IW, planes, dtype=dt, device=dev)
        L.gemm(x, blk['c1'][0], y1, bias=blk['c1'][1], act=L.ACT_RELU)
        y2 = torch.empty(B, OH, OW, planes, dtype=dt, device=dev)
        L.gemm(y1, blk['c2'][0], y2, bias=blk['c2'][1], act=L.ACT_RELU,
               conv=dict(IH=IH, IW=IW, Cin=planes, OH=OH, OW=OW, KW=3, SH=s, SW=s, PH=1, PW=1))
        if blk['down'] is None:
            skip = x
        else:
            skip = torch.empty(B, OH, OW, planes * 4, dtype=dt, device=dev)
            if s == 1:
                L.gemm(x, blk['down'][0], skip, bias=blk['down'][1])
            else:
                L.gemm(x, blk['down'][0], skip, bias=blk['down'][1],
                       conv=dict(IH=IH, IW=IW, Cin=Cin, OH=OH, OW=OW, KW=1, SH=s, SW=s, PH=0, PW=0))
        x = torch.empty(B, OH, OW, planes * 4, dtype=dt, device=dev)
        L.gemm(y2, blk['c3'][0], x, bias=blk['c3'][1], act=L.ACT_RELU, residual=skip.view(-1, planes * 4), res_first=True)
    return x


def _run_blocks_x3(P, x):
    """layer1-4 in the bf16x3 numerics: fp32 NHWC maps between the convolutions; every convolution = ONE launch of the bf16 MFMA kernel on the
    K-concatenated split operands ([x_hi | x_lo | x_hi] . [W_hi | W_hi | W_lo]^T = the three split products, fp32 accumulate; bias, skip and ReLU
    in its fp32 epilogue).  The block input is split once for conv1 and the downsample conv."""
    dev, f32, bf = x.device, torch.float32, torch.bfloat16
    B = x.shape[0]
    x3 = L.split3(x)                                                               # [B, IH, IW, 3 Cin] bf16 (the max-pool output: the only separate split pass)
    nblk = len(P['blocks'])
    for bi, blk in enumerate(P['blocks']):
        _, IH, IW, Cin = x.shape
        s = blk['stride']
        OH, OW = (IH - 1) // s + 1, (IW - 1) // s + 1
        planes = blk['c1'][0].shape[0]
        # every convolution writes the split-bf16 operand form of its fp32 output next to it (epi_flags bit 8): no split pass between the layers
        y1, y1s = torch.empty(B, IH, IW, planes, dtype=f32, device=dev), torch.empty(B, IH, IW, 3 * planes, dtype=bf, device=dev)
        L.gemm(x3, blk['c1'][0], y1, bias=blk['c1'][1], act=L.ACT_RELU, split3_out=y1s)
        y2, y2s = torch.empty(B, OH, OW, planes, dtype=f32, device=dev), torch.empty(B, OH, OW, 3 * planes, dtype=bf, device=dev)
        L.gemm(y1s, blk['c2'][0], y2, bias=blk['c2'][1], act=L.ACT_RELU, split3_out=y2s,
               conv=dict(IH=IH, IW=IW, Cin=3 * planes, OH=OH, OW=OW, KW=3, SH=s, SW=s, PH=1, PW=1))
        if blk['down'] is None:
            skip = x
        else:
            skip = torch.empty(B, OH, OW, planes * 4, dtype=f32, device=dev)
            if s == 1:
                L.gemm(x3, blk['down'][0], skip, bias=blk['down'][1])
            else:
                L.gemm(x3, blk['down'][0], skip, bias=blk['down'][1],
                       conv=dict(IH=IH, IW=IW, Cin=3 * Cin, OH=OH, OW=OW, KW=1, SH=s, SW=s, PH=0, PW=0))
        out = torch.empty(B, OH, OW, planes * 4, dtype=f32, device=dev)
        outs = torch.empty(B, OH, OW, 12 * planes, dtype=bf, device=dev) if bi + 1 < nblk else None
        L.gemm(y2s, blk['c3'][0], out, bias=blk['c3'][1], act=L.ACT_RELU, residual=skip.view(-1, planes * 4), res_first=True, split3_out=outs)
        x, x3 = out, outs
    return x


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

        self.numerics = 'fp32'              # parity-grade by default (the reference runs it in fp32); WHMR sets its own mode here ('bf16' = throughput opt-in)
        self._prep = None
        self.fused_logits = None

    # ------------------------------------------------------------------ HIP path
    def _versions(self):
        return tuple(t._version for t in list(self.parameters()) + list(self.buffers())) + (self.numerics, str(self.fc_vfov.weight.device))

    def _prepare(self):
        """Fold BN into the conv weights, lay them out [N, (ky, kx, ci)] in the compute dtype (cached until a tensor changes)."""
        key = self._versions()
        if self._prep is not None and self._prep['key'] == key:
            return self._prep
        dt = torch.bfloat16 if self.numerics == 'bf16' else torch.float32
        prep = fold_resnet50(self.backbone, dt, x3=self.numerics == 'bf16x3')
        prep['key'] = key
        prep['fc_w'] = torch.cat([self.fc_vfov.weight, self.fc_pitch.weight, self.fc_roll.weight], 0).detach().float().contiguous()
        prep['fc_b'] = torch.cat([self.fc_vfov.bias, self.fc_pitch.bias, self.fc_roll.bias], 0).detach().float().contiguous()
        self._prep = prep
        return prep

    @torch.no_grad()
    def forward(self, images):
        if not images.is_cuda:
            raise RuntimeError('whmr_amd.CameraRegressorNetwork runs on a HIP device only (no CPU fallback)')
        if self.training:
            raise RuntimeError('the HIP ResNet-50 folds eval-mode BatchNorm; call .eval() (the reference freezes cam_model, whmr.py:507)')
        P = self._prepare()
        dt = torch.bfloat16 if self.numerics == 'bf16' else torch.float32
        dev = images.device
        B = images.shape[0]
        x = run_resnet50(P, images, dt)
        feat = L.avgpool_nhwc(x)
        logits = torch.empty(B, P['fc_w'].shape[0], dtype=torch.float32, device=dev)
        L.gemm(feat, P['fc_w'], logits, bias=P['fc_b'])
        n = self.num_out_channels
        self.fused_logits = logits           # [B, 3 n] = [vfov | pitch | roll]: what WHMR._camera hands to whmr_cam_head (explicit, not via ._base)
        return [logits[:, :n], logits[:, n:2 * n], logits[:, 2 * n:]], feat


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
    """pare.utils.geometry.batch_euler2matrix [3P pare==0.1 = DECA rotation_converter, restated]: euler (x, y, z) -> quaternion
    q = qx * qy * qz -> rotation matrix, i.e. R = Rx(x) . Ry(y) . Rz(z); the reference feeds [pitch, 0, roll] (models/whmr.py:521-522),
    so cam_rotmat = Rx(pitch) . Rz(roll)."""
    h = r * 0.5
    cx, cy, cz = torch.cos(h[:, 0]), torch.cos(h[:, 1]), torch.cos(h[:, 2])
    sx, sy, sz = torch.sin(h[:, 0]), torch.sin(h[:, 1]), torch.sin(h[:, 2])
    q = torch.stack([cx * cy * cz - sx * sy * sz, cx * sy * sz + cy * cz * sx,
                     cx * cz * sy - sx * cy * sz, cx * cy * sz + sx * cz * sy], dim=1)
    q = q / q.norm(dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    w2, x2, y2, z2 = w * w, x * x, y * y, z * z
    wx, wy, wz, xy, xz, yz = w * x, w * y, w * z, x * y, x * z, y * z
    return torch.stack([w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz, 2 * wz + 2 * xy, w2 - x2 + y2 - z2,
                        2 * yz - 2 * wx, 2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2], dim=1).view(-1, 3, 3)
