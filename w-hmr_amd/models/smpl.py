"""SMPL body model on MI355X: the object W-HMR calls as ``self.smpl(betas=, body_pose=, global_orient=, pose2rot=False)``.

Stands in for ``pare.models.SMPL`` (pare==0.1 wrapping smplx==0.1.28; commented twin at models/smpl.py:61-83 of the
reference): same call signature, output with ``.vertices`` [B,6890,3] and ``.joints`` [B,49,3], ``.faces``.
Buffers use the smplx names (v_template, shapedirs, posedirs [207,20670], J_regressor, lbs_weights, parents,
faces_tensor, J_regressor_extra) so a reference checkpoint's ``regressor.N.smpl.*`` keys load.
"""
import os
import pickle
from collections import namedtuple

import numpy as np
import torch
import torch.nn as nn

from .. import _lib as L
from ..core import constants as K

SMPL_MODEL_DIR = 'data/smpl'
SMPL_MEAN_PARAMS = 'data/smpl_mean_params.npz'
JOINT_REGRESSOR_TRAIN_EXTRA = 'data/J_regressor_extra.npy'
H36M_TO_J17 = K.H36M_TO_J17
H36M_TO_J14 = K.H36M_TO_J14

ModelOutput = namedtuple('ModelOutput', ['vertices', 'joints', 'smpl_joints', 'rotmat', 'pose_aa', 'markers', 'post'], defaults=(None,))


def _np(a):
    import scipy.sparse
    return np.asarray(a.todense() if scipy.sparse.issparse(a) else a, dtype=np.float32)


def load_smpl_arrays(model_dir=SMPL_MODEL_DIR, extra=JOINT_REGRESSOR_TRAIN_EXTRA):
    """Read the licensed SMPL_NEUTRAL.pkl + J_regressor_extra.npy when present (never shipped with this repo)."""
    with open(os.path.join(model_dir, 'SMPL_NEUTRAL.pkl'), 'rb') as f:
        d = pickle.load(f, encoding='latin1')
    nv = d['v_template'].shape[0]
    return {'v_template': torch.from_numpy(_np(d['v_template'])), 'shapedirs': torch.from_numpy(_np(d['shapedirs'])[:, :, :10]),
            'posedirs': torch.from_numpy(_np(d['posedirs']).reshape(nv * 3, -1).T.copy()),
            'J_regressor': torch.from_numpy(_np(d['J_regressor'])), 'lbs_weights': torch.from_numpy(_np(d['weights'])),
            'J_regressor_extra': torch.from_numpy(np.load(extra).astype(np.float32)),
            'faces': torch.from_numpy(np.asarray(d['f'], dtype=np.int64))}


class SMPL(nn.Module):
    NUM_VERTS, NUM_JOINTS = 6890, 24

    def __init__(self, model_path=SMPL_MODEL_DIR, arrays=None, marker_ids=None, **kwargs):
        super().__init__()
        a = arrays if arrays is not None else load_smpl_arrays(model_path)
        for k in ('v_template', 'shapedirs', 'posedirs', 'J_regressor', 'lbs_weights', 'J_regressor_extra'):
            self.register_buffer(k, a[k].float().contiguous().clone())
        faces = a.get('faces', torch.zeros(1, 3, dtype=torch.int64))
        self.register_buffer('faces_tensor', faces.long())
        self.register_buffer('parents', torch.tensor(K.SMPL_PARENTS, dtype=torch.long))
        self.register_buffer('extra_joints_idxs', torch.tensor(K.EXTRA_VERTEX_IDS, dtype=torch.long))
        self.register_buffer('joint_map', torch.tensor(K.JOINT_MAP_49, dtype=torch.long))
        ids = marker_ids if marker_ids is not None else torch.zeros(0, dtype=torch.long)
        self.marker_ids = torch.as_tensor(ids, dtype=torch.long)
        self._dev_cache = None
        # launch form: pose chain -> blend shapes + skinning -> CSR joint regression + stage tail (3 launches)
        self.csr_tail = True        # False: the dense B x 33-workgroup regression + tail launch of round 2 (A/B)
        self.blend_skin = True      # pose-corrective offsets + skinning as one launch; False: fp32 GEMM into a [B, 20670] buffer + the skin kernel (A/B)
        self.offsets_x3 = False     # blend launch: pose-corrective offsets on split-bf16 MFMA operands (~3e-7 of a vertex; WHMR sets it in the bf16 / bf16x3 numerics); False: exact f32

    @property
    def faces(self):
        return self.faces_tensor.cpu().numpy()

    # ---- device-side constants (folded joint regressors, int32 index tables) -- built once per device / buffer version
    def _model(self):
        dev = self.v_template.device
        ver = (dev,) + tuple((t._version, t.data_ptr()) for t in (self.v_template, self.shapedirs, self.posedirs, self.lbs_weights,
                                                                  self.J_regressor, self.J_regressor_extra))
        if self._dev_cache is None or self._dev_cache[0] != ver:
            i32 = lambda t: t.to(device=dev, dtype=torch.int32).contiguous()
            # J = Jreg . (T + S beta) = (Jreg . T) + (Jreg . S) beta: fold the 24x6890 regressor into 24x3(+x10) constants
            Jreg64 = self.J_regressor.double()
            keep = {'regs': torch.cat([self.J_regressor_extra, self.J_regressor], 0).contiguous(),   # [9 + 24, 6890]
                    'posedirs_t': self.posedirs.t().contiguous(),                                     # [20670, 207]: GEMM weight layout
                    'J_template': (Jreg64 @ self.v_template.double()).float().contiguous(),
                    'shapedirs_t': self.shapedirs.reshape(self.NUM_VERTS, 30).t().contiguous(),         # [30, 6890]: coalesced over vertices
                    'lbs_weights_t': self.lbs_weights.t().contiguous(),                               # [24, 6890]
                    'J_shapedirs': torch.einsum('jv,vcl->jcl', Jreg64, self.shapedirs.double()).float().contiguous(),
                    'parents': i32(self.parents), 'extra': i32(self.extra_joints_idxs), 'jmap': i32(self.joint_map),
                    'markers': i32(self.marker_ids)}
            regs = keep['regs']                          # CSR of the 33 regressor rows for the one-launch call (real regressors are > 99 % zeros)
            nz = regs != 0
            ptr = torch.zeros(regs.shape[0] + 1, dtype=torch.int32, device=dev)
            ptr[1:] = nz.sum(1).cumsum(0).to(torch.int32)
            keep['csr'] = (ptr, nz.nonzero()[:, 1].to(torch.int32).contiguous(), regs[nz].contiguous())
            # posedirs per 64-vertex chunk, k-major inside the chunk, zero padded (k 207 -> 208, vertices 6890 -> 6912): one contiguous 160 KB
            # tile per work item of the one-launch call
            pt = torch.zeros(208, 108 * 192, dtype=torch.float32, device=dev)
            pt[:207, :self.NUM_VERTS * 3] = self.posedirs
            keep['posedirs_tiled'] = pt.view(208, 108, 192).permute(1, 0, 2).contiguous()
            # the same tile as split-bf16 planes in MFMA operand order (whmr_smpl_blend_skin_x3): k = 16 s + 8 h + j -> [chunk][s][h][plane][column][j]
            hi_, lo_ = L.split_bf16(pt.view(13, 2, 8, 108, 192))
            keep['posedirs_x3'] = torch.stack([hi_, lo_], 0).permute(4, 1, 2, 0, 5, 3).contiguous()
            m = L.WhmrSmplModel()
            m.v_template, m.shapedirs = self.v_template.data_ptr(), keep['shapedirs_t'].data_ptr()
            m.posedirs, m.lbs_weights = self.posedirs.data_ptr(), keep['lbs_weights_t'].data_ptr()
            m.J_template, m.J_shapedirs = keep['J_template'].data_ptr(), keep['J_shapedirs'].data_ptr()
            m.J_regressor_extra = keep['regs'].data_ptr()
            m.J_regressor = keep['regs'].data_ptr() + 9 * self.NUM_VERTS * 4
            m.parents, m.extra_vertex_ids = keep['parents'].data_ptr(), keep['extra'].data_ptr()
            m.joint_map, m.marker_ids = keep['jmap'].data_ptr(), keep['markers'].data_ptr() if keep['markers'].numel() else None
            m.n_markers = int(keep['markers'].numel())
            self._dev_cache = (ver, m, keep)
        return self._dev_cache[1]

    @torch.no_grad()
    def run(self, betas, rotmats, gram_schmidt=False, want_aa=False, want_smpl_joints=False, want_markers=False, post=None, nxt=None):
        """betas [B,10], rotmats [B,24,3,3] (raw 3x3 blocks when gram_schmidt=True) -> ModelOutput.
        ``post`` / ``nxt`` (Regressor stages, see ``_lib.smpl_stage_tail``): the joint regression, the 49-joint gather, the stage's projections and
        the next stage's input state leave in ONE launch; the projections come back as ``ModelOutput.post``."""
        if not betas.is_cuda:
            raise RuntimeError('whmr_amd.SMPL runs on a HIP device only (no CPU fallback)')
        if L.PROFILE is not None and not getattr(self, '_in_profile', False):
            # bench.py's instrumented step: HIP events around the whole call; algorithmic bytes per SURVEY 8(d): 84 172 B per image of I/O
            # (betas, rotmats, vertices, joints) + 19.6 MB of model constants per call
            self._in_profile = True
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            try:
                out = self.run(betas, rotmats, gram_schmidt, want_aa, want_smpl_joints, want_markers, post, nxt)
            finally:
                self._in_profile = False
            e1.record()
            L.PROFILE.append(('smpl_call', betas.shape[0] * 84172.0 + 19.6e6, e0, e1))
            return out
        B, dev = betas.shape[0], betas.device
        f32 = dict(dtype=torch.float32, device=dev)
        m = self._model()
        # rows may be column slices of the regressor state buffer: only the innermost dimension has to be dense
        betas = betas if (betas.dtype == torch.float32 and betas.stride(-1) == 1) else betas.float().contiguous()
        pose9 = rotmats if rotmats.dim() == 2 else rotmats.reshape(B, 216)
        pose9 = pose9 if (pose9.dtype == torch.float32 and pose9.stride(-1) == 1) else pose9.float().contiguous()
        rot = torch.empty(B, 24, 3, 3, **f32)
        aa = torch.empty(B, 72, **f32) if want_aa else None
        A = torch.empty(B, 24, 12, **f32)
        pj = torch.empty(B, 24, 3, **f32)
        pf = torch.empty(B, 207, **f32)
        L.smpl_pose_chain(m, pose9, betas, gram_schmidt, rot, aa, A, pj, pf)
        verts = torch.empty(B, self.NUM_VERTS, 3, **f32)
        if self.blend_skin:         # pose-corrective offsets (verts.py:51-53) on the matrix pipes + shape blend + skinning, one launch
            if self.offsets_x3:
                L.smpl_blend_skin_x3(m, self._dev_cache[2]['posedirs_x3'], betas, pf, A, verts)
            else:
                L.smpl_blend_skin(m, self._dev_cache[2]['posedirs_tiled'], betas, pf, A, verts)
        else:                       # one fp32 MFMA GEMM [B,207] x [207,20670], then blend + skin
            pose_off = torch.empty(B, self.NUM_VERTS * 3, **f32)
            L.gemm(pf, self._dev_cache[2]['posedirs_t'], pose_off)
            L.smpl_skin(m, betas, pf, A, verts, pose_off)
        joints = torch.empty(B, 49, 3, **f32)
        sj = torch.empty(B, 45, 3, **f32) if want_smpl_joints else None
        mk = torch.empty(B, m.n_markers, 3, **f32) if (want_markers and m.n_markers) else None
        tail = None
        if post is not None:
            tail = L.smpl_stage_tail(m, verts, pj, joints, sj, mk, post=dict(post, aa=aa), nxt=None if nxt is None else dict(nxt, rotmat=rot),
                                     csr=self._dev_cache[2]['csr'] if self.csr_tail else None)
        else:
            L.smpl_stage_tail(m, verts, pj, joints, sj, mk, csr=self._dev_cache[2]['csr'] if self.csr_tail else None)
        return ModelOutput(verts, joints, sj, rot, aa, mk, tail)

    def forward(self, betas=None, body_pose=None, global_orient=None, pose2rot=False, **kwargs):
        """pare.models.SMPL.forward call shape (whmr.py:132-137).  pose2rot=True takes axis-angle [B,69] / [B,3]."""
        if pose2rot:
            from ..utils.geometry import batch_rodrigues
            B = betas.shape[0]
            full = torch.cat([global_orient.reshape(B, -1), body_pose.reshape(B, -1)], dim=1).reshape(-1, 3)
            rot = batch_rodrigues(full).reshape(B, 24, 3, 3)
        else:
            rot = torch.cat([global_orient.reshape(-1, 1, 3, 3), body_pose.reshape(-1, 23, 3, 3)], dim=1)
        return self.run(betas, rot)


def get_smpl_faces():
    return SMPL(SMPL_MODEL_DIR).faces
