"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the demo's person crop.

Follows datasets/data_utils/img_utils.py:53-101 (gen_trans_from_patch_cv, generate_patch_image_cv), :209-242
(get_single_image_crop_demo) and :318-326 (ToTensor + Normalize) of the reference.  The pixel arithmetic lives in OpenCV
(cv2.getAffineTransform / cv2.warpAffine, 8-bit INTER_LINEAR, BORDER_CONSTANT), a third-party dependency that is NOT installed in
this image: **parity unpinned** -- the algorithm below restates OpenCV's imgwarp.cpp fixed-point scheme (AB_BITS 10, 1/32-pixel
coordinates, 15-bit weights, (sum + 2^14) >> 15) from its published source; the HIP kernel is checked bit-exactly against THIS.
Its geometry and interpolation (pixel-centre convention, inverse map, zero border blended in at the frame edge) are pinned against an independent
implementation -- scipy.ndimage.map_coordinates at the exact inverse-affine coordinates, agreement within the quantisation of the fixed-point
scheme (tests/test_oracle_cpu.py::test_crop_oracle_against_an_independent_bilinear_warp); the bit-level rounding rule remains unpinned.
"""
import numpy as np

MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def rotate_2d(pt, rot_rad):
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    return np.array([pt[0] * cs - pt[1] * sn, pt[0] * sn + pt[1] * cs], dtype=np.float32)


def get_affine_transform(src, dst):
    """cv2.getAffineTransform: the 2x3 map taking the three src points to the dst points (float64 solve)."""
    a = np.zeros((6, 6)); b = np.zeros(6)
    for i in range(3):
        a[i, 0:3] = [src[i, 0], src[i, 1], 1.0]
        a[i + 3, 3:6] = [src[i, 0], src[i, 1], 1.0]
        b[i], b[i + 3] = dst[i, 0], dst[i, 1]
    return np.linalg.solve(a, b).reshape(2, 3)


def gen_trans_from_patch_cv(c_x, c_y, src_width, src_height, dst_width, dst_height, scale, rot, inv=False):
    """img_utils.py:53-87."""
    src_w, src_h = src_width * scale, src_height * scale
    src_center = np.array([c_x, c_y], dtype=np.float64)
    rot_rad = np.pi * rot / 180
    src_down = rotate_2d(np.array([0, src_h * 0.5], dtype=np.float32), rot_rad)
    src_right = rotate_2d(np.array([src_w * 0.5, 0], dtype=np.float32), rot_rad)
    dst_center = np.array([dst_width * 0.5, dst_height * 0.5], dtype=np.float32)
    src = np.zeros((3, 2), dtype=np.float32)
    src[0], src[1], src[2] = src_center, src_center + src_down, src_center + src_right
    dst = np.zeros((3, 2), dtype=np.float32)
    dst[0] = dst_center
    dst[1] = dst_center + np.array([0, dst_height * 0.5], dtype=np.float32)
    dst[2] = dst_center + np.array([dst_width * 0.5, 0], dtype=np.float32)
    return get_affine_transform(dst, src) if inv else get_affine_transform(src, dst)


def invert_affine(m):
    """cv::invertAffineTransform (double)."""
    m = np.asarray(m, dtype=np.float64)
    d = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    d = 1.0 / d if d != 0 else 0.0
    a11, a22 = m[1, 1] * d, m[0, 0] * d
    a12, a21 = -m[0, 1] * d, -m[1, 0] * d
    b1 = -a11 * m[0, 2] - a12 * m[1, 2]
    b2 = -a21 * m[0, 2] - a22 * m[1, 2]
    return np.array([[a11, a12, b1], [a21, a22, b2]])


def warp_affine_u8(img, trans, dsize):
    """cv2.warpAffine(img, trans, dsize, flags=INTER_LINEAR, borderMode=BORDER_CONSTANT) for uint8 HWC images."""
    w, h = int(dsize[0]), int(dsize[1])
    H, W = img.shape[:2]
    M = invert_affine(trans).reshape(-1)
    x = np.arange(w, dtype=np.float64)
    adelta = np.rint(M[0] * x * 1024.0).astype(np.int64)
    bdelta = np.rint(M[3] * x * 1024.0).astype(np.int64)
    y = np.arange(h, dtype=np.float64)
    X0 = np.rint((M[1] * y + M[2]) * 1024.0).astype(np.int64) + 16
    Y0 = np.rint((M[4] * y + M[5]) * 1024.0).astype(np.int64) + 16
    X = (X0[:, None] + adelta[None, :]) >> 5
    Y = (Y0[:, None] + bdelta[None, :]) >> 5
    sx, sy, fx, fy = X >> 5, Y >> 5, X & 31, Y & 31
    acc = np.full((h, w, img.shape[2]), 1 << 14, dtype=np.int64)
    src = img.astype(np.int64)
    for dy, dx, wt in ((0, 0, (32 - fy) * (32 - fx) * 32), (0, 1, (32 - fy) * fx * 32), (1, 0, fy * (32 - fx) * 32), (1, 1, fy * fx * 32)):
        yy, xx = sy + dy, sx + dx
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        px = src[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)]
        acc += np.where(ok, wt, 0)[..., None] * px
    return (acc >> 15).astype(np.uint8)


def generate_patch_image_cv(img, c_x, c_y, bb_width, bb_height, patch_width, patch_height, do_flip, scale, rot):
    """img_utils.py:89-101."""
    if do_flip:
        img = img[:, ::-1, :]
        c_x = img.shape[1] - c_x - 1
    trans = gen_trans_from_patch_cv(c_x, c_y, bb_width, bb_height, patch_width, patch_height, scale, rot)
    return warp_affine_u8(img, trans, (int(patch_width), int(patch_height))), trans


def to_tensor_normalize(patch_u8):
    """transforms.ToTensor + Normalize (img_utils.py:318-326): HWC uint8 -> CHW float32."""
    t = patch_u8.astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)
    return (t - MEAN[:, None, None]) / STD[:, None, None]


def get_single_image_crop_demo(image, bbox, kp_2d=None, scale=1.2, crop_size=224):
    """img_utils.py:209-242 (kp_2d transform omitted when None, as in the demo)."""
    patch, trans = generate_patch_image_cv(image.copy(), bbox[0], bbox[1], bbox[2], bbox[3], crop_size, crop_size, False, scale, 0)
    return to_tensor_normalize(patch), patch, kp_2d
