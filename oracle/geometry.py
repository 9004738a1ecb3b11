"""Oracle restatement of the reference's rotation / projection helpers.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function cites the
reference lines it follows (paths relative to /root/reference).  All math is
fp32 torch on CPU, written independently of the reference source.
"""
import torch

FOCAL_LENGTH = 1000.0          # core/constants.py:4
IMG_RES = 256.0                # configs/pymaf_config.yaml:83-85 (WIDTH = HEIGHT = 256)


def _normalize(v, eps=1e-12):
    # F.normalize(p=2, dim=1): v / max(||v||, eps)
    n = v.norm(dim=-1, keepdim=True).clamp_min(eps)
    return v / n


def _cross(a, b):
    # utils/geometry.py:256,266-269 call torch.cross without dim; for [N,3]
    # operands with N != 3 that is the last dim.
    return torch.linalg.cross(a, b, dim=-1)


def rot6d_to_rotmat(x):
    """utils/geometry.py:243-257.  x[...,6] viewed (-1,3,2); columns interleaved."""
    x = x.reshape(-1, 3, 2)
    a1, a2 = x[:, :, 0], x[:, :, 1]
    b1 = _normalize(a1)
    proj = (b1 * a2).sum(-1, keepdim=True)
    b2 = _normalize(a2 - proj * b1)
    b3 = _cross(b1, b2)
    return torch.stack((b1, b2, b3), dim=-1)


def rotmat_to_rot6d(x):
    """utils/geometry.py:275-286: first two columns, row-major flattened."""
    return x[:, :, :2].reshape(x.shape[0], 6)


def unbiased_gram_schmidt(x):
    """utils/geometry.py:260-272 on [B,k,3,3]."""
    k = x.shape[1]
    m = x.reshape(-1, 3, 3)
    t1, t2, t3 = m[:, :, 0], m[:, :, 1], m[:, :, 2]
    r1 = _normalize((_cross(t2, t3) + t1) / 2.0)
    r2_ = (_cross(t3, r1) + t2) / 2.0
    r2 = _normalize(r2_ - (r2_ * r1).sum(-1, keepdim=True) * r1)
    r3 = _cross(r1, r2)
    return torch.stack((r1, r2, r3), dim=-1).reshape(-1, k, 3, 3)


def quat_to_rotmat(q):
    """utils/geometry.py:30-51 (w,x,y,z), re-normalised first."""
    q = q / q.norm(dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    w2, x2, y2, z2 = w * w, x * x, y * y, z * z
    wx, wy, wz, xy, xz, yz = w * x, w * y, w * z, x * y, x * z, y * z
    return torch.stack([
        w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
        2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
        2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2], dim=1).view(-1, 3, 3)


def batch_rodrigues(theta):
    """utils/geometry.py:14-27: norm of (theta+1e-8), division of the unshifted theta."""
    angle = (theta + 1e-8).norm(dim=1, keepdim=True)
    axis = theta / angle
    half = angle * 0.5
    return quat_to_rotmat(torch.cat([torch.cos(half), torch.sin(half) * axis], dim=1))


def rotation_matrix_to_quaternion(R, eps=1e-6):
    """utils/geometry.py:160-240 (kornia legacy), on [N,3,3].

    The reference transposes the 3x4 matrix and indexes rmat_t[:, a, b] = R[:, b, a].
    """
    def m(a, b):          # rmat_t[:, a, b]
        return R[:, b, a]
    d0, d1, d2 = m(0, 0), m(1, 1), m(2, 2)
    mask_d2 = d2 < eps
    mask_d0_d1 = d0 > d1
    mask_d0_nd1 = d0 < -d1
    t0 = 1 + d0 - d1 - d2
    q0 = torch.stack([m(1, 2) - m(2, 1), t0, m(0, 1) + m(1, 0), m(2, 0) + m(0, 2)], -1)
    t1 = 1 - d0 + d1 - d2
    q1 = torch.stack([m(2, 0) - m(0, 2), m(0, 1) + m(1, 0), t1, m(1, 2) + m(2, 1)], -1)
    t2 = 1 - d0 - d1 + d2
    q2 = torch.stack([m(0, 1) - m(1, 0), m(2, 0) + m(0, 2), m(1, 2) + m(2, 1), t2], -1)
    t3 = 1 + d0 + d1 + d2
    q3 = torch.stack([t3, m(1, 2) - m(2, 1), m(2, 0) - m(0, 2), m(0, 1) - m(1, 0)], -1)
    c0 = (mask_d2 & mask_d0_d1).to(R.dtype).unsqueeze(-1)
    c1 = (mask_d2 & ~mask_d0_d1).to(R.dtype).unsqueeze(-1)
    c2 = (~mask_d2 & mask_d0_nd1).to(R.dtype).unsqueeze(-1)
    c3 = (~mask_d2 & ~mask_d0_nd1).to(R.dtype).unsqueeze(-1)
    q = q0 * c0 + q1 * c1 + q2 * c2 + q3 * c3
    den = t0.unsqueeze(-1) * c0 + t1.unsqueeze(-1) * c1 + t2.unsqueeze(-1) * c2 + t3.unsqueeze(-1) * c3
    return 0.5 * q / torch.sqrt(den)


def quaternion_to_angle_axis(q):
    """utils/geometry.py:86-136."""
    q1, q2, q3 = q[..., 1], q[..., 2], q[..., 3]
    s2 = q1 * q1 + q2 * q2 + q3 * q3
    s = torch.sqrt(s2)
    c = q[..., 0]
    two_theta = 2.0 * torch.where(c < 0.0, torch.atan2(-s, -c), torch.atan2(s, c))
    k = torch.where(s2 > 0.0, two_theta / s, torch.full_like(s, 2.0))
    return torch.stack([q1 * k, q2 * k, q3 * k], dim=-1)


def rotation_matrix_to_angle_axis(R):
    """utils/geometry.py:54-83 on [N,3,3]; NaN -> 0 (:82)."""
    aa = quaternion_to_angle_axis(rotation_matrix_to_quaternion(R.reshape(-1, 3, 3)))
    return torch.where(torch.isnan(aa), torch.zeros_like(aa), aa)


def perspective_projection(points, rotation, translation, focal_length, camera_center):
    """utils/geometry.py:310-341.  rotation may have batch 1 (whmr.py:158-160)."""
    B = points.shape[0]
    K = torch.zeros(B, 3, 3, dtype=points.dtype)
    K[:, 0, 0] = focal_length
    K[:, 1, 1] = focal_length
    K[:, 2, 2] = 1.0
    K[:, :2, 2] = camera_center
    p = torch.einsum('bij,bkj->bki', rotation.expand(B, -1, -1), points) + translation.unsqueeze(1)
    p = p / p[:, :, 2:3]
    p = torch.einsum('bij,bkj->bki', K, p)
    return p[:, :, :2]


def projection(joints, cam):
    """utils/geometry.py:289-307: weak-perspective camera -> [-1,1] crop coords."""
    B = joints.shape[0]
    t = torch.stack([cam[:, 1], cam[:, 2], 2 * FOCAL_LENGTH / (IMG_RES * cam[:, 0] + 1e-9)], dim=-1)
    eye = torch.eye(3, dtype=joints.dtype).unsqueeze(0)
    kp = perspective_projection(joints, eye, t, FOCAL_LENGTH, torch.zeros(B, 2, dtype=joints.dtype))
    return kp / (IMG_RES / 2.0)


def convert_pare_to_full_img_cam(cam, bbox_height, center, img_w, img_h, Tz):
    """utils/geometry.py:139-157 with focal_length=None (tz = Tz)."""
    s, tx, ty = cam[:, 0], cam[:, 1], cam[:, 2]
    cx = 2 * (center[:, 0] - img_w / 2.0) / (s * bbox_height)
    cy = 2 * (center[:, 1] - img_h / 2.0) / (s * bbox_height)
    return torch.stack([tx + cx, ty + cy, Tz], dim=-1)


def batch_euler2matrix(r):
    """pare.utils.geometry.batch_euler2matrix [3P pare==0.1; source absent from /root/reference -> restated from the published code:
    PARE takes it verbatim from DECA's decalib/utils/rotation_converter.py (euler_to_quaternion -> quaternion_to_rotation_matrix)].

    euler (x, y, z) in radians -> quaternion q = qx * qy * qz (Hamilton products of the three axis quaternions, half angles):
        w = cx cy cz - sx sy sz      x = cx sy sz + cy cz sx      y = cx cz sy - sx cy sz      z = cx cy sz + sx cz sy
    -> rotation matrix (w, x, y, z), i.e. R = Rx(x) . Ry(y) . Rz(z).  With the reference's input [pitch, 0, roll]
    (models/whmr.py:521-522) that is Rx(pitch) . Rz(roll): the roll about the optical axis is applied FIRST, then the pitch.
    Pinned by the composition known answers in tests/test_oracle_cpu.py (pitch-only, roll-only, pitch + roll = Rx . Rz); the round-1
    restatement had the (+, -, +, -) sign pattern of qz * qy * qx (R = Rz . Ry . Rx), which differs whenever pitch and roll are both
    non-zero.
    """
    h = r * 0.5
    cx, cy, cz = torch.cos(h[:, 0]), torch.cos(h[:, 1]), torch.cos(h[:, 2])
    sx, sy, sz = torch.sin(h[:, 0]), torch.sin(h[:, 1]), torch.sin(h[:, 2])
    q = torch.stack([
        cx * cy * cz - sx * sy * sz,
        cx * sy * sz + cy * cz * sx,
        cx * cz * sy - sx * cy * sz,
        cx * cy * sz + sx * cz * sy], dim=1)
    return quat_to_rotmat(q)


def estimate_translation_np(S, joints_2d, joints_conf, focal_length=5000, img_size=(224., 224.)):
    """utils/geometry.py:344-385: weighted least squares for the camera translation of one sample (numpy, float64)."""
    import numpy as np
    n = S.shape[0]
    f = np.array([focal_length, focal_length])
    center = np.array(img_size) / 2.
    Z = np.reshape(np.tile(S[:, 2], (2, 1)).T, -1)
    XY = np.reshape(S[:, 0:2], -1)
    O = np.tile(center, n)
    Fv = np.tile(f, n)
    w2 = np.reshape(np.tile(np.sqrt(joints_conf), (2, 1)).T, -1)
    Q = np.array([Fv * np.tile(np.array([1, 0]), n), Fv * np.tile(np.array([0, 1]), n), O - np.reshape(joints_2d, -1)]).T
    c = (np.reshape(joints_2d, -1) - O) * Z - Fv * XY
    Q = w2[:, None] * Q                                   # == np.dot(np.diagflat(w2), Q)
    c = w2 * c
    return np.linalg.solve(Q.T @ Q, Q.T @ c)


def estimate_translation(S, joints_2d, focal_length=5000., img_size=(224., 224.)):
    """utils/geometry.py:388-408: joints 25:49 only, per-sample solve, float32 result."""
    import numpy as np
    Sn = S[:, 25:, :].cpu().numpy()
    j = joints_2d[:, 25:, :].cpu().numpy()
    out = np.zeros((Sn.shape[0], 3), dtype=np.float32)
    for i in range(Sn.shape[0]):
        out[i] = estimate_translation_np(Sn[i], j[i, :, :-1], j[i, :, -1], focal_length=focal_length, img_size=list(img_size))
    return torch.from_numpy(out).to(S.device)
