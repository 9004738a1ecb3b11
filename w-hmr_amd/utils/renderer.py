"""IUV ground-truth renderer of the training step -- the reference's ``utils/renderer.py::IUV_Renderer`` (:296-446) on the HIP rasteriser.

The reference renders DensePose IUV images of the fitted mesh with pytorch3d (MeshRasterizer faces_per_pixel=1, blur 0 + HardFlatShader over
per-vertex (I/24, U, V) textures) every training step with AUX supervision (core/trainer.py:442-464): one of the host-side stalls of SURVEY 8f
N3.  Here the same image comes from three small launches of ``whmr_iuv_rasterize`` (csrc/rasterize.hip).  Same constructor role and method:

    iuv_maker = IUV_Renderer(orig_size=(cfg.IMG_RES.HEIGHT, cfg.IMG_RES.WIDTH), output_size=cfg.MODEL.PyMAF.DP_HEATMAP_SIZE)
    iuv_image_gt = iuv_maker.verts2iuvimg(opt_vertices, cam=gt_camera)          # [B, 3, 128, 128]

The DensePose tables (data/UV_data/UV_Processed.mat via DensePoseMethods: All_vertices, FacesDensePose, FaceIndices, U_norm, V_norm) are
licensed data the reference reads at construction; pass them as ``dp=dict(vert_mapping, faces, textures_vts)`` (tests use a synthetic mesh).
"""
import os

import numpy as np
import torch

from .. import _lib as L

FOCAL_LENGTH = 1000.0        # core/constants.py:4


def load_densepose_tables(uv_mat='data/UV_data/UV_Processed.mat', vert_pid='data/dp_vert_pid.npy'):
    """renderer.py:299-327: vertex duplication map, DensePose faces and the per-vertex (I / num_part, U, V) texture."""
    from scipy.io import loadmat
    m = loadmat(uv_mat)
    vert_mapping = m['All_vertices'].reshape(-1).astype(np.int64) - 1
    faces = (m['All_Faces'] - 1).astype(np.int32)
    face_idx = m['All_FaceIndices'].reshape(-1)
    num_part = float(face_idx.max())
    if os.path.exists(vert_pid):
        pid = np.load(vert_pid)
    else:                                                            # first face that uses a vertex decides its part (renderer.py:315-320)
        pid = np.zeros(len(vert_mapping))
        seen = np.zeros(len(vert_mapping), dtype=bool)
        for i, f in enumerate(faces):
            for v in f:
                if not seen[v]:
                    pid[v], seen[v] = face_idx[i], True
    tex = np.stack([pid / num_part, m['All_U_norm'].reshape(-1), m['All_V_norm'].reshape(-1)], 1).astype(np.float32)
    return dict(vert_mapping=torch.from_numpy(vert_mapping), faces=torch.from_numpy(faces), textures_vts=torch.from_numpy(tex))


class IUV_Renderer:
    def __init__(self, focal_length=FOCAL_LENGTH, orig_size=(224, 224), output_size=(56, 56), mode='iuv', device=None, mesh_type='smpl', dp=None):
        assert mode == 'iuv' and mesh_type == 'smpl'
        self.focal_length, self.orig_size = float(focal_length), tuple(orig_size)
        self.output_size = tuple(output_size) if isinstance(output_size, (tuple, list)) else (output_size, output_size)
        dp = dp if dp is not None else load_densepose_tables()
        self.vert_mapping = dp.get('vert_mapping')
        self.faces = dp['faces'].to(torch.int32).contiguous()
        self.textures_vts = dp['textures_vts'].float().contiguous()
        # K of renderer.py:362-380: focal on the diagonal, principal point = image centre, then ALL FOUR entries scaled by orig / 224 when the
        # image is not 224 (so the principal point is not the centre of a 256 image: reproduced as written)
        fx = fy = self.focal_length
        px, py = self.orig_size[1] / 2.0, self.orig_size[0] / 2.0
        if self.orig_size[0] != 224:
            sw, sh = self.orig_size[1] / 224.0, self.orig_size[0] / 224.0
            fx, px, fy, py = fx * sw, px * sw, fy * sh, py * sh
        self.K = (fx, fy, px, py)
        self._dev = {}

    def _on(self, dev):
        t = self._dev.get(dev)
        if t is None:
            t = self._dev[dev] = (self.faces.to(dev), self.textures_vts.to(dev), self.vert_mapping.to(dev) if self.vert_mapping is not None else None)
        return t

    @torch.no_grad()
    def verts2iuvimg(self, verts, cam, iwp_mode=True, want_faces=False):
        """verts [B, 6890, 3], cam [B, 3] = (s, tx, ty) -> IUV image [B, 3, H, W] (background 0).  renderer.py:412-433."""
        if not verts.is_cuda:
            raise RuntimeError('whmr_amd.IUV_Renderer runs on a HIP device only (no CPU fallback)')
        faces, tex, vmap = self._on(verts.device)
        return L.iuv_rasterize(verts.float(), faces, tex, cam.float(), self.K, self.focal_length, self.orig_size, self.output_size, vmap=vmap,
                               want_faces=want_faces)
