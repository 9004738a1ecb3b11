"""Input side of the demo on the GPU: person crops + normalisation for all detections of a frame in one launch.

Mirrors ``get_single_image_crop_demo`` (datasets/data_utils/img_utils.py:209-242; demo/tester.py:112-122 calls it once per
detection with scale=1.0, crop_size=256 and then feeds ``inp_images[:, :, :, 32:-32]``).  ``crop_persons`` produces the same
normalised patches for a whole frame: the uint8 frame is uploaded / read once, the affine maps (6 doubles per person) are
built on the host exactly like gen_trans_from_patch_cv + cv2.getAffineTransform, the warp + ToTensor + Normalize run in
``whmr_crop_normalize``.  With ``x_slice=(32, 224)`` only the 192 columns the model consumes are computed.
"""
import numpy as np
import torch

from .. import _lib as L

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def _affine(src, dst):
    a = np.zeros((6, 6)); b = np.zeros(6)
    for i in range(3):
        a[i, 0:3] = [src[i, 0], src[i, 1], 1.0]
        a[i + 3, 3:6] = [src[i, 0], src[i, 1], 1.0]
        b[i], b[i + 3] = dst[i, 0], dst[i, 1]
    return np.linalg.solve(a, b).reshape(2, 3)


def gen_trans_from_patch_cv(c_x, c_y, src_width, src_height, dst_width, dst_height, scale, rot, inv=False):
    """img_utils.py:53-87 (float32 control points, float64 solve)."""
    src_w, src_h = src_width * scale, src_height * scale
    rot_rad = np.pi * rot / 180
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    down = np.array([-(src_h * 0.5) * sn, (src_h * 0.5) * cs], dtype=np.float32)
    right = np.array([(src_w * 0.5) * cs, (src_w * 0.5) * sn], dtype=np.float32)
    center = np.array([c_x, c_y], dtype=np.float64)
    src = np.zeros((3, 2), dtype=np.float32)
    src[0], src[1], src[2] = center, center + down, center + right
    dc = np.array([dst_width * 0.5, dst_height * 0.5], dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    dst[0], dst[1], dst[2] = dc, dc + np.array([0, dst_height * 0.5], dtype=np.float32), dc + np.array([dst_width * 0.5, 0], dtype=np.float32)
    return _affine(dst, src) if inv else _affine(src, dst)


def _invert_affine(m):
    d = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    d = 1.0 / d if d != 0 else 0.0
    a11, a22, a12, a21 = m[1, 1] * d, m[0, 0] * d, -m[0, 1] * d, -m[1, 0] * d
    return np.array([a11, a12, -a11 * m[0, 2] - a12 * m[1, 2], a21, a22, -a21 * m[0, 2] - a22 * m[1, 2]])


@torch.no_grad()
def crop_persons(frame, bboxes, crop_size=256, scale=1.0, x_slice=None, want_raw=False):
    """frame: uint8 [H, W, 3] RGB (torch tensor on the HIP device, or numpy -> uploaded once); bboxes: iterable of
    (c_x, c_y, w, h).  Returns normalised patches [B, 3, crop_size, x1 - x0] fp32 (and the uint8 patches when want_raw)."""
    if isinstance(frame, np.ndarray):
        frame = torch.from_numpy(np.ascontiguousarray(frame)).cuda()
    if not frame.is_cuda:
        raise RuntimeError('crop_persons runs on a HIP device only (no CPU fallback)')
    assert frame.dtype == torch.uint8 and frame.dim() == 3 and frame.shape[2] == 3 and frame.stride(2) == 1 and frame.stride(1) == 3
    H, W = frame.shape[:2]
    x0, x1 = x_slice if x_slice is not None else (0, crop_size)
    inv = np.stack([_invert_affine(gen_trans_from_patch_cv(b[0], b[1], b[2], b[3], crop_size, crop_size, scale, 0)) for b in bboxes])
    B = inv.shape[0]
    inv_d = torch.from_numpy(inv).to(frame.device)                       # [B, 6] float64
    out = torch.empty(B, 3, crop_size, x1 - x0, dtype=torch.float32, device=frame.device)
    raw = torch.empty(B, crop_size, x1 - x0, 3, dtype=torch.uint8, device=frame.device) if want_raw else None
    L.crop_normalize(frame, inv_d, crop_size, crop_size, x0, x1, out, raw, MEAN, STD)
    return (out, raw) if want_raw else out


def get_single_image_crop_demo(image, bbox, kp_2d=None, scale=1.2, crop_size=224):
    """img_utils.py:209-242 call shape: (normalised patch [3, cs, cs], uint8 patch [cs, cs, 3], kp_2d)."""
    assert kp_2d is None, 'the demo passes kp_2d=None (demo/tester.py:115-121)'
    out, raw = crop_persons(image, [bbox], crop_size=crop_size, scale=scale, want_raw=True)
    return out[0], raw[0], kp_2d
