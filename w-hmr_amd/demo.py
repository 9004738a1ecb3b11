"""Per-frame inference pipeline on the device: the input preparation of demo/tester.py:95-146 + the model call of :150-162.

    out = infer_frame(model, frame_u8, dets)        # frame: uint8 [H, W, 3] RGB, dets: [(c_x, c_y, w, h), ...] person boxes

What the reference does per frame on the CPU -- one cv2.warpAffine + ToTensor + Normalize per detection, bbox_info in numpy,
PIL resize of the full image, then ``full_x`` replicated once per person -- happens here in a handful of launches: the frame is
uploaded once, ``crop_persons`` cuts and normalises all person patches in one launch (only the 192 columns the model reads), the
camera-calibration ResNet-50 sees the resized frame ONCE (batch-1 ``full_x`` is broadcast), and the meta tensors
are a few dozen floats computed on the host.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .datasets.img_utils import MEAN, STD, crop_persons


def frame_meta(dets, orig_h, orig_w, device):
    """demo/tester.py:110-146: center, scale (= w / 200), bbox_height, orig_shape (H, W) and the CLIFF-style bbox_info."""
    d = np.asarray(dets, dtype=np.float64).reshape(-1, 4)
    n = d.shape[0]
    scale = d[:, 2] / 200.
    focal = math.sqrt(orig_h ** 2 + orig_w ** 2)                                       # pesudo_focal, tester.py:134-136
    cx, cy = d[:, 0] - orig_w / 2., d[:, 1] - orig_h / 2.
    info = np.stack([cx, cy, 200. * scale, np.full(n, float(orig_w)), np.full(n, float(orig_h))], 1) / np.float32(focal)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)
    return dict(center=t(d[:, :2]), scale=t(scale), bbox_height=t(200. * scale),
                orig_shape=t(np.tile(np.array([[orig_h, orig_w]], dtype=np.float32), (n, 1))), bbox_info=t(info.astype(np.float32)))


@torch.no_grad()
def full_image_input(frame_u8, short_side=600):
    """tester.py:47-53,101-103: Resize(short side 600, bilinear + antialias like PIL) -> ToTensor -> Normalize, [1, 3, h, w] fp32."""
    H, W = frame_u8.shape[:2]
    s = short_side / min(H, W)
    h, w = (short_side, int(short_side * W / H)) if H <= W else (int(short_side * H / W), short_side)
    x = frame_u8.permute(2, 0, 1).unsqueeze(0).float()
    if (h, w) != (H, W):
        x = F.interpolate(x, size=(h, w), mode='bilinear', align_corners=False, antialias=s < 1).round_().clamp_(0, 255)
    mean = torch.tensor(MEAN, device=x.device).view(1, 3, 1, 1)
    std = torch.tensor(STD, device=x.device).view(1, 3, 1, 1)
    return (x / 255. - mean) / std


@torch.no_grad()
def prepare_frame(frame_u8, dets, with_full_image=True):
    """frame_u8: uint8 [H, W, 3] RGB on the HIP device (or numpy: uploaded once).  Returns the keyword arguments of WHMR.forward."""
    if isinstance(frame_u8, np.ndarray):
        frame_u8 = torch.from_numpy(np.ascontiguousarray(frame_u8)).cuda()
    H, W = frame_u8.shape[:2]
    kw = frame_meta(dets, H, W, frame_u8.device)
    kw['x'] = crop_persons(frame_u8, dets, crop_size=256, scale=1.0, x_slice=(32, 224))        # == inp_images[:, :, :, 32:-32]
    kw['full_x'] = full_image_input(frame_u8) if with_full_image else None
    return kw


@torch.no_grad()
def infer_frame(model, frame_u8, dets, with_full_image=True):
    """One frame, all detected persons: the vis_dict of WHMR.forward (tester.py:150-165)."""
    kw = prepare_frame(frame_u8, dets, with_full_image)
    return model(x=kw['x'], meta_masks=None, center=kw['center'], scale=kw['scale'], bbox_height=kw['bbox_height'],
                 orig_shape=kw['orig_shape'], bbox_info=kw['bbox_info'], is_train=False, J_regressor=None, full_x=kw['full_x'])


class FramePipeline:
    """infer_frame with the model call (cam_model + W-HMR forward, ~200 launches) replayed from a HIP graph per (number of persons,
    full-image size): at one to a few persons per frame the eager path is bound by Python launch overhead, not by the GPU."""

    def __init__(self, model, with_full_image=True):
        self.model, self.with_full_image, self._graphs = model, with_full_image, {}

    @torch.no_grad()
    def __call__(self, frame_u8, dets):
        from .graph import GraphedForward
        kw = prepare_frame(frame_u8, dets, self.with_full_image)
        key = (kw['x'].shape[0], tuple(kw['full_x'].shape) if kw['full_x'] is not None else None)
        call = dict(x=kw['x'], meta_masks=None, center=kw['center'], scale=kw['scale'], bbox_height=kw['bbox_height'],
                    orig_shape=kw['orig_shape'], bbox_info=kw['bbox_info'], is_train=False, J_regressor=None, full_x=kw['full_x'])
        g = self._graphs.get(key)
        if g is None:
            g = self._graphs[key] = GraphedForward(self.model, **call)
        return g(**call)
