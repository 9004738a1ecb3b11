"""Geometry helpers with the reference's names and signatures (utils/geometry.py), computed by HIP kernels.

Device tensors only (no CPU fallback).  Inference path: no autograd.  estimate_translation* (geometry.py:344-408) is
trainer-side numpy code and out of scope.
"""
import torch

from .. import _lib as L
from ..core.cfgs import cfg
from ..core.constants import FOCAL_LENGTH


def batch_rodrigues(theta):
    """geometry.py:14-27: [B,3] axis-angle -> [B,3,3]."""
    return L.rot_to_mat(theta.reshape(-1, 3).float(), L.RODRIGUES)


def rot6d_to_rotmat(x):
    """geometry.py:243-257: [...,6] -> [N,3,3]."""
    return L.rot_to_mat(x.reshape(-1, 6).float(), L.ROT6D)


def unbiased_gram_schmidt(x):
    """geometry.py:260-272: [B,k,3,3] -> [B,k,3,3]."""
    k = x.shape[1]
    return L.rot_to_mat(x.reshape(-1, 9).float(), L.GRAM_SCHMIDT).reshape(-1, k, 3, 3)


def rotmat_to_rot6d(x):
    """geometry.py:275-286 (pure view/copy: first two columns)."""
    return x[:, :, :2].reshape(x.shape[0], 6)


def rotation_matrix_to_angle_axis(rotation_matrix):
    """geometry.py:54-83: [N,3,3] (or [N,3,4]) -> [N,3], NaN -> 0."""
    if rotation_matrix.shape[1:] == (3, 4):
        rotation_matrix = rotation_matrix[:, :, :3]
    return L.mat_to_aa(rotation_matrix.reshape(-1, 9).float())


def perspective_projection(points, rotation, translation, focal_length, camera_center, retain_z=False):
    """geometry.py:310-341.  rotation may have batch 1 (whmr.py:158-160)."""
    assert not retain_z, 'retain_z=True is never used on the W-HMR path'
    return L.perspective(points, rotation, translation, focal_length, camera_center)


def projection(pred_joints, pred_camera, retain_z=False):
    """geometry.py:289-307: weak-perspective projection to [-1,1] crop coordinates (cfg.IMG_RES, FOCAL_LENGTH)."""
    assert not retain_z
    return L.weak_projection(pred_joints, pred_camera, FOCAL_LENGTH, float(cfg.IMG_RES.WIDTH), float(cfg.IMG_RES.HEIGHT))


def convert_pare_to_full_img_cam(pare_cam, bbox_height, bbox_center, img_w, img_h, focal_length=None, Tz=None):
    """geometry.py:139-157.  [B]-sized elementwise arithmetic; kept as tensor expressions (3 values per image)."""
    s, tx, ty = pare_cam[:, 0], pare_cam[:, 1], pare_cam[:, 2]
    tz = Tz if focal_length is None else 2 * focal_length / (bbox_height * s)
    cx = 2 * (bbox_center[:, 0] - (img_w / 2.)) / (s * bbox_height)
    cy = 2 * (bbox_center[:, 1] - (img_h / 2.)) / (s * bbox_height)
    return torch.stack([tx + cx, ty + cy, tz], dim=-1)


def estimate_translation(S, joints_2d, focal_length=5000., img_size=(224., 224.)):
    """utils/geometry.py:388-408: camera translation [B,3] that best re-projects S [B,49,3] onto joints_2d [B,49,3] (x, y, conf),
    GT joints 25:49 only.  One HIP launch on the device instead of .cpu() + a numpy loop over the batch (core/trainer.py:435)."""
    return L.estimate_translation(S, joints_2d, 25, S.shape[1] - 25, float(focal_length), float(img_size[0]), float(img_size[1]))
