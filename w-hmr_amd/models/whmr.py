"""W-HMR on MI355X: the reference's ``WHMR`` / ``whmr_net`` module surface over libwhmr_hip.so.

Mirrors models/whmr.py of yw0208/W-HMR: ``Regressor`` (:42-269), ``Global_Orient_Regressor`` (:272-305), ``WHMR`` (:308-678),
``whmr_net`` (:681-687) -- same constructor role, same ``forward(x, meta_masks, center, scale, bbox_height, orig_shape,
bbox_info, is_train=False, J_regressor=None, full_x=None, cam_rotmat=None)`` and the same ``state_dict`` keys
(SURVEY App. B).  Differences, all deliberate and documented in DESIGN.md:
  * three return views: ``view='vis'`` (released default, the 9-tensor vis_dict), ``'train'``
    ((out_list, vis_feat_list)), ``'eval'`` (({'global_output': ...}, None)); select per call or via ``self.return_view``;
  * ``full_x=None`` / ``cam_rotmat=...`` work (the release raises NameError): render_rotmat := cam_rotmat;
  * ``is_train=True`` (module in ``.train()``) returns the train view with an autograd graph whose every stage has a hand-written HIP
    backward (``whmr_amd.train``): Dropout and the ViT's stochastic depth draw fresh masks, BatchNorm uses batch statistics and updates
    its running statistics; in eval mode they are identities / running statistics;
  * ``numerics``: 'bf16x3' (DEFAULT: split-bf16 operand pairs on the bf16 matrix pipes for ViT / deconv / Tz conv, fp32 everything else -- the
    reference computes in fp32 (whmr.py:503-678) and a module that is swapped in behind ``whmr_net`` must land within the 1e-4 contract: every
    ``vis_dict`` tensor <= 1e-4 element-wise of the reference; trains in 'fp32'), 'fp32' (exact-f32 MFMA everywhere) or 'bf16' (the explicit
    THROUGHPUT opt-in: bf16 MFMA operands for ViT / deconv / Tz conv / cam_model, 3x faster, pred_cam_t / focal_length ~2.6e-3 and cam_rotmat
    up to 2e-2 off the reference; what ``bench.py``'s headline -- BASELINE configs[1], "bf16 inference" -- runs);
  * ``load_state_dict(ckpt['model'], strict=True)`` accepts a reference checkpoint unchanged (third-party smplx / pare keys without a
    counterpart are listed in ``ignored_checkpoint_keys``).
Every tensor op on the hot path is a kernel of this repo, including the camera-calibration ResNet-50 (``cam_model.py``: NHWC implicit
GEMMs); a few O(batch)-sized scalar expressions (focal length, camera translation) are tensor arithmetic.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from .. import _lib as L
from ..core.cfgs import cfg
from ..core.constants import H36M_TO_J14
from ..utils.geometry import (convert_pare_to_full_img_cam, rot6d_to_rotmat, rotation_matrix_to_angle_axis,
                              unbiased_gram_schmidt)
from .cam_model import CameraRegressorNetwork, batch_euler2matrix, convert_preds_to_angles
from .maf_extractor import MAF_Extractor
from .pose_vit import _Holder, _linear_params, _ln_params, get_vitpose_encoder
from .smpl import SMPL, SMPL_MEAN_PARAMS, load_smpl_arrays

BN_MOMENTUM = 0.1
SMPL_Marker = 'data/smpl/smpl_ssm.npy'
MESH_DOWNSAMPLING = 'data/mesh_downsampling.npz'


def _cpu_rot6d_to_rotmat(x):
    """geometry.py:243-257 on the host, for constructor-time buffers only (whmr.py:64-65,284-286)."""
    x = x.reshape(-1, 3, 2)
    a1, a2 = x[:, :, 0], x[:, :, 1]
    b1 = a1 / a1.norm(dim=1, keepdim=True).clamp_min(1e-12)
    u = a2 - (b1 * a2).sum(-1, keepdim=True) * b1
    b2 = u / u.norm(dim=1, keepdim=True).clamp_min(1e-12)
    return torch.stack((b1, b2, torch.linalg.cross(b1, b2, dim=-1)), dim=-1)


def load_assets(smpl_mean_params=SMPL_MEAN_PARAMS):
    """The reference's data files (whmr.py:62-100; never shipped here): SMPL pkl, mean params, markers, down-sampling."""
    import scipy.sparse
    mp = np.load(smpl_mean_params)
    g = np.load(MESH_DOWNSAMPLING, allow_pickle=True, encoding='latin1')
    D = [torch.from_numpy(np.asarray(scipy.sparse.coo_matrix(d).todense(), dtype=np.float32)) for d in g['D']]
    return {'smpl': load_smpl_arrays(), 'ssm': torch.from_numpy(np.load(SMPL_Marker).astype(np.int64)),
            'Dmap0': D[0], 'Dmap1': D[1], 'mean_params': {k: mp[k] for k in ('pose', 'shape', 'cam')}}


_H36M_IDX = {}


def h36m_joints(verts, J_regressor):
    """whmr.py:176-180 / :647-651: J_regressor [17,6890] . verts, LSP-14 subset, pelvis-centred (fp32 GEMM kernel)."""
    B = verts.shape[0]
    J = J_regressor.float().to(verts.device).contiguous()
    vt = verts.permute(0, 2, 1).reshape(B * 3, -1).contiguous()
    jj = torch.empty(J.shape[0], B * 3, dtype=torch.float32, device=verts.device)
    L.gemm(J, vt, jj)
    jj = jj.view(J.shape[0], B, 3).permute(1, 0, 2)
    idx = _H36M_IDX.get(verts.device)
    if idx is None:
        idx = _H36M_IDX[verts.device] = torch.tensor(H36M_TO_J14, dtype=torch.long, device=verts.device)
    return jj.index_select(1, idx) - jj[:, :1]


class _Cache:
    """Derived device-side operands (folded / re-ordered / cast weights), rebuilt when a source tensor changes."""

    def __init__(self):
        self.d = {}

    def get(self, key, srcs, fn):
        ver = tuple((t.device, t._version, t.data_ptr()) for t in srcs)
        ent = self.d.get(key)
        if ent is None or ent[0] != ver:
            ent = self.d[key] = (ver, fn())
        return ent[1]


_CAM_STREAMS = {}


def compose_tz_weights(w0, w1):
    """whmr.py:418-421: Conv2d(256, 64, k7, s3, bias=False) followed by Conv2d(64, 5, k7, s2, bias=False) is the single convolution
    Conv2d(256, 5, k25, s6) with Wc[o, ci, A, B] = sum_{c1, 3u + a = A, 3v + b = B} w1[o, c1, u, v] * w0[c1, ci, a, b] (fp64 here).  Returned as the weight
    matrix of the space-to-depth implicit GEMM (``WHMR._tz_tokens_composed``): [128, 36 Ci] fp32, row n = (jA * 5 + jB) * 5 + o (125 used), column
    k = (q * 6 + p) * Ci + ci, entry Wc[o, ci, q + 6 jA, p + 6 jB] (zero where a tap index exceeds 24)."""
    w0, w1 = w0.double(), w1.double()                                                   # [64, Ci, 7, 7], [5, 64, 7, 7]
    Ci = w0.shape[1]
    wc = torch.zeros(5, Ci, 30, 30, dtype=torch.float64, device=w0.device)              # taps 25..29 stay zero (A = q + 6 jA <= 29)
    for u in range(7):
        for v in range(7):
            wc[:, :, 3 * u:3 * u + 7, 3 * v:3 * v + 7] += torch.einsum('oc,cikl->oikl', w1[:, :, u, v], w0)
    g = wc.view(5, Ci, 5, 6, 5, 6).permute(2, 4, 0, 3, 5, 1).reshape(125, 36 * Ci)      # [(jA, jB, o), (q, p, ci)]
    return torch.cat([g, g.new_zeros(3, g.shape[1])], 0).float().contiguous()           # 128 rows


class Regressor(nn.Module):
    """whmr.py:42-269.  fc1 -> fc2 (no nonlinearity) -> decpose/decshape/deccam residual heads -> SMPL -> projections."""

    def __init__(self, feat_dim, smpl_mean_params, smpl=None, assets=None):
        super().__init__()
        npose = 24 * 9
        self.fc1 = nn.Linear(feat_dim + npose + 13 + 5, 1024)
        self.drop1 = nn.Dropout()
        self.fc2 = nn.Linear(1024, 1024)
        self.drop2 = nn.Dropout()
        self.decpose = nn.Linear(1024, npose)
        self.decshape = nn.Linear(1024, 10)
        self.deccam = nn.Linear(1024, 3)
        for m in (self.decpose, self.decshape, self.deccam):
            nn.init.xavier_uniform_(m.weight, gain=0.01)
        self.smpl = smpl
        mp = assets['mean_params']
        init_pose = _cpu_rot6d_to_rotmat(torch.from_numpy(np.asarray(mp['pose'], dtype=np.float32)).reshape(1, 24, 6)).reshape(1, -1)
        self.register_buffer('init_pose', init_pose)
        self.register_buffer('init_shape', torch.from_numpy(np.asarray(mp['shape'], dtype=np.float32)).unsqueeze(0))
        self.register_buffer('init_cam', torch.from_numpy(np.asarray(mp['cam'], dtype=np.float32)).unsqueeze(0))
        self.register_buffer('Dmap0', assets['Dmap0'].float().clone())      # dense, like whmr.py:95-98
        self.register_buffer('Dmap1', assets['Dmap1'].float().clone())
        self.ssm = np.asarray(assets['ssm'])
        self._cache = _Cache()

    def _collapsed(self):
        """fc1 -> fc2 -> [decpose | decshape | deccam] has no nonlinearity between its layers (whmr.py:118-126; the dropouts are
        identities in eval), so the stage is ONE affine map: W = Wh.W2.W1 [229, in], b = Wh.(W2.b1 + b2) + bh.  Built once per
        weight version in fp64 and rounded to fp32: the regressor then streams 2.2 MB instead of 13.5 MB per iteration and is
        one launch instead of three.  (Different fp32 association than the reference's three sgemm calls: ~1e-7 relative.)"""
        ws = [self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, self.decpose.weight, self.decshape.weight,
              self.deccam.weight, self.decpose.bias, self.decshape.bias, self.deccam.bias]

        def build():
            w1, b1, w2, b2 = (t.detach().double() for t in ws[:4])
            wh = torch.cat([w.detach().double() for w in ws[4:7]], 0)
            bh = torch.cat([b.detach().double() for b in ws[7:]], 0)
            w21 = w2 @ w1
            return (wh @ w21).float().contiguous(), (wh @ (w2 @ b1 + b2) + bh).float().contiguous()
        return self._cache.get('collapsed', ws, build)

    def _downsample(self, verts):
        """whmr.py:182-183 (train view only): sub_verts = Dmap0 . verts, temp_verts = Dmap1 . sub_verts.  The reference multiplies the
        densified matrices; here they are compressed once per buffer version (CSR) and applied as a gather (whmr_csr_apply3)."""
        from ..train.heads_autograd import downsample_csr
        c = downsample_csr(self.Dmap0, self.Dmap1, self.__dict__.setdefault('_ds_cache', {}))
        sub = L.csr_apply3(c['d0'], verts, self.Dmap0.shape[0])
        return sub, L.csr_apply3(c['d1'], sub, self.Dmap1.shape[0])

    def _outputs(self, out, state, scale, J_regressor, with_aux, Tz=None, orig_shape=None, center=None, bbox_height=None):
        """state [B,229] = [pose(216) | shape(10) | cam(3)] rows (possibly a strided view).  whmr.py:139-209."""
        verts, joints = out.vertices, out.joints
        pose_flat, shape, cam = state[:, :216], state[:, 216:226], state[:, 226:]
        d = {'verts': verts, 'kp_3d': joints, 'smpl_kp_3d': out.smpl_joints, 'rotmat': out.rotmat, 'pred_cam': cam,
             'pred_shape': shape, 'pred_pose': pose_flat, 'pose': out.pose_aa,
             'pelvis': out.smpl_joints[:, :1] if out.smpl_joints is not None else None, 'markers': out.markers}
        if Tz is not None:      # theta, kp_2d, focal (whmr.py:147-149), cam_t, kp_2d_w (whmr.py:165-173): computed by the stage-tail launch
            theta, kp_2d, kp_w, cam_t, focal = out.post if out.post is not None else L.regressor_post(
                state, out.pose_aa, joints, Tz, bbox_height, center, orig_shape, 1000.0, float(cfg.IMG_RES.WIDTH), float(cfg.IMG_RES.HEIGHT))
            d.update(theta=theta, kp_2d=kp_2d, kp_2d_w=kp_w, pred_cam_t=cam_t, scale=scale, focal_length=focal)
        else:
            d['theta'] = torch.cat([cam, shape, out.pose_aa], dim=1)
            d['kp_2d'] = L.weak_projection(joints, cam.contiguous(), 1000.0, float(cfg.IMG_RES.WIDTH), float(cfg.IMG_RES.HEIGHT))
        if J_regressor is not None:                                                   # whmr.py:176-180
            d['kp_3d'] = h36m_joints(verts, J_regressor)
        if with_aux:
            d['sub_verts'], d['temp_verts'] = self._downsample(verts)
        return d

    def forward(self, x, bbox_info, Tz, orig_shape, center, scale, bbox_height, init_pose=None, init_shape=None,
                init_cam=None, is_train=False, n_iter=1, J_regressor=None, with_aux=True, xc=None, xc_next=None, state_ready=False):
        """x [B, feat] (or pre-filled xc buffer [B, feat+5+229] whose first ``feat`` columns hold x) -> (dict, body_feat)."""
        if is_train:       # whmr.py:102-209 in training: autograd nodes with HIP forward + backward (whmr_amd.train.whmr_train)
            from ..train.whmr_train import regressor_train
            assert n_iter == 1 and J_regressor is None, 'the training graph is built for n_iter = 1 without J_regressor (core/trainer.py:410)'
            B = bbox_info.shape[0]
            pose = self.init_pose.expand(B, -1) if init_pose is None else init_pose
            shape = self.init_shape.expand(B, -1) if init_shape is None else init_shape
            cam = self.init_cam.expand(B, -1) if init_cam is None else init_cam
            return regressor_train(self, x, bbox_info.float(), Tz, orig_shape.float(), center.float(), scale, bbox_height.float(), pose, shape,
                                   cam, self.__dict__.setdefault('_train_cache', {}))
        with torch.no_grad():
            return self._forward_eval(x, bbox_info, Tz, orig_shape, center, scale, bbox_height, init_pose, init_shape, init_cam, n_iter,
                                      J_regressor, with_aux, xc, xc_next, state_ready)

    def _forward_eval(self, x, bbox_info, Tz, orig_shape, center, scale, bbox_height, init_pose, init_shape, init_cam, n_iter, J_regressor,
                      with_aux, xc, xc_next=None, state_ready=False):
        """``xc_next`` = (buffer, F) of the NEXT stage: its state columns [bbox_info | rotmat | shape | cam] are written by this stage's tail launch;
        ``state_ready``: this stage's own state columns were written that way by the previous stage (no regressor_state launch)."""
        B = bbox_info.shape[0]
        dev = bbox_info.device
        F = self.fc1.in_features - 229 - 5
        if xc is None:
            xc = torch.empty(B, F + 5 + 229, dtype=torch.float32, device=dev)
            xc[:, :F] = x
        pose = (self.init_pose.expand(B, -1) if init_pose is None else init_pose).reshape(B, -1)
        shape = self.init_shape.expand(B, -1) if init_shape is None else init_shape
        cam = self.init_cam.expand(B, -1) if init_cam is None else init_cam
        if pose.stride(-1) != 1 or pose.dtype != torch.float32:
            pose = pose.float().contiguous()
        if not state_ready:
            L.regressor_state(xc, F, bbox_info.float().contiguous(), pose, shape.float(), cam.float())     # whmr.py:105,119, one launch
        new = torch.empty(B, 229, dtype=torch.float32, device=dev)
        w_eff, b_eff = self._collapsed()
        for _ in range(n_iter):
            L.gemm(xc, w_eff, new, bias=b_eff, residual=xc[:, F + 5:])                # whmr.py:118-126 as one affine map (+ residual state)
            if n_iter > 1:
                xc[:, F + 5:] = new
        post = nxt = None
        if Tz is not None:
            post = dict(state=new, Tz=Tz, bbox_h=bbox_height, center=center, orig_shape=orig_shape, focal0=1000.0,
                        res_w=float(cfg.IMG_RES.WIDTH), res_h=float(cfg.IMG_RES.HEIGHT))
            if xc_next is not None:
                nxt = dict(bbox_info=bbox_info, xc=xc_next[0], F=xc_next[1])
        out = self.smpl.run(new[:, 216:226], new[:, :216], gram_schmidt=True, want_aa=True, want_smpl_joints=True,
                            want_markers=True, post=post, nxt=nxt)                    # whmr.py:128-137,174,184-187
        d = self._outputs(out, new, scale, J_regressor, with_aux, Tz, orig_shape, center, bbox_height)
        self._last_stage = (new, out.pose_aa, out.joints)          # for WHMR's deferred Tz finalize (Tz head on a side stream)
        return d, xc[:, :F + 5]

    @torch.no_grad()
    def forward_init(self, x, init_pose=None, init_shape=None, init_cam=None, n_iter=1, J_regressor=None, with_aux=True):
        """whmr.py:211-269: the mean-pose mesh (no Gram-Schmidt, rotmats = init_pose as stored)."""
        B = x.shape[0]
        state = torch.cat([self.init_pose.expand(B, -1) if init_pose is None else init_pose.reshape(B, -1),
                           self.init_shape.expand(B, -1) if init_shape is None else init_shape,
                           self.init_cam.expand(B, -1) if init_cam is None else init_cam], dim=1).float().contiguous()
        out = self.smpl.run(state[:, 216:226], state[:, :216], gram_schmidt=False, want_aa=True, want_smpl_joints=True,
                            want_markers=True)
        return self._outputs(out, state, None, J_regressor, with_aux)


class Global_Orient_Regressor(nn.Module):
    """whmr.py:272-305.  In eval the three 'iterations' are identical (local_orient is never fed back): one pass."""

    def __init__(self, smpl_mean_params=None, assets=None):
        super().__init__()
        self.fc1 = nn.Linear(2149 + 6 + 9, 2048)
        self.drop1 = nn.Dropout()
        self.fc2 = nn.Linear(2048, 2048)
        self.drop2 = nn.Dropout()
        self.decrot = nn.Linear(2048, 9)
        nn.init.xavier_uniform_(self.decrot.weight, gain=0.01)
        mp = assets['mean_params']
        init_pose = _cpu_rot6d_to_rotmat(torch.from_numpy(np.asarray(mp['pose'], dtype=np.float32)).reshape(1, 24, 6)).reshape(1, 24, 9)
        self.register_buffer('init_pose', init_pose[:, 0])
        self._cache = _Cache()

    def _collapsed(self):
        """fc1 -> fc2 -> decrot without a nonlinearity (whmr.py:295-301) = one [9, 2164] affine map (see Regressor._collapsed)."""
        ws = [self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, self.decrot.weight, self.decrot.bias]

        def build():
            w1, b1, w2, b2, wd, bd = (t.detach().double() for t in ws)
            return (wd @ (w2 @ w1)).float().contiguous(), (wd @ (w2 @ b1 + b2) + bd).float().contiguous()
        return self._cache.get('collapsed', ws, build)

    def forward(self, x, cam_rotmat, local_orient, is_train=False, xc=None, raw=False):
        """xc (optional): a buffer with >= 2164 columns whose first 2149 already hold x (the last regressor stage's input buffer);
        columns 2149..2163 are overwritten in place instead of copying x.  local_orient: [B, 3, 3] or the stage's whole rotmat [B, 24, 3, 3]
        (its root block is read in place).  raw: return the head's output before the Gram-Schmidt step.
        is_train=True (whmr.py:289-305 as called at :631 in training): the three Linear layers as autograd nodes with HIP forward / backward
        (train.heads_autograd.LinearFn), Dropout active in ``.train()``, no Gram-Schmidt -> pred_rot [B, 1, 3, 3] with a graph into fc1 / fc2 /
        decrot, ``x`` (the last stage's body_feat) and ``local_orient``.  The reference runs the stack three times on the SAME input and keeps
        the last result (local_orient is never fed back, :293-301): one pass, one dropout draw."""
        if is_train:
            return self._forward_train(x, cam_rotmat, local_orient)
        with torch.no_grad():
            return self._forward_eval(x, cam_rotmat, local_orient, xc, raw)

    def _forward_train(self, x, cam_rotmat, local_orient):
        from ..train.heads_autograd import LinearFn
        B = x.shape[0]
        rot6 = cam_rotmat.detach()[:, :, :2].reshape(B, 6)                             # rotmat_to_rot6d (geometry.py:275-286); cam_rotmat is detached (whmr.py:509-524)
        lo = local_orient.reshape(B, -1)
        xc = torch.cat([x, rot6.to(x.dtype), lo], dim=1)
        h = self.drop1(LinearFn.apply(xc, self.fc1.weight, self.fc1.bias))
        h = self.drop2(LinearFn.apply(h, self.fc2.weight, self.fc2.bias))
        return (LinearFn.apply(h, self.decrot.weight, self.decrot.bias) + lo).reshape(-1, 1, 3, 3)

    def _forward_eval(self, x, cam_rotmat, local_orient, xc=None, raw=False):
        B, dev = cam_rotmat.shape[0], cam_rotmat.device
        if xc is None:
            xc = torch.empty(B, 2149 + 6 + 9, dtype=torch.float32, device=dev)
            xc[:, :2149] = x
        xc = xc[:, :2164]
        rot = local_orient if local_orient.dim() == 4 else local_orient.reshape(B, 1, 3, 3)          # the stage's rotmat [B, 24, 3, 3] or its root block
        L.orient_state(cam_rotmat, rot, xc, 2149)                                        # [rot6d(cam_rotmat) | local_orient], geometry.py:275-286
        r = torch.empty(B, 9, dtype=torch.float32, device=dev)
        w_eff, b_eff = self._collapsed()
        L.gemm(xc, w_eff, r, bias=b_eff, residual=xc[:, 2155:], lda=xc.stride(0))
        if raw:
            return r                                                                     # WHMR.forward finishes with L.orient_tail (one launch)
        return unbiased_gram_schmidt(r.reshape(-1, 1, 3, 3))


class IUV_predict_layer(nn.Module):
    """Parameter container of models/iuv_predictor.py:71-91 (dp_head).  Its output is discarded by the released
    forward (whmr.py:656-658,663-678), so no inference view evaluates it; kept for state_dict parity."""

    def __init__(self, feat_dim=256, final_cov_k=3, part_out_dim=25):
        super().__init__()
        pad = 1 if final_cov_k == 3 else 0
        self.predict_u = nn.Conv2d(feat_dim, 25, final_cov_k, 1, pad)
        self.predict_v = nn.Conv2d(feat_dim, 25, final_cov_k, 1, pad)
        self.predict_ann_index = nn.Conv2d(feat_dim, 15, final_cov_k, 1, pad)
        self.predict_uv_index = nn.Conv2d(feat_dim, 25, final_cov_k, 1, pad)


class WHMR(nn.Module):
    """whmr.py:308-678 (vitpose branch)."""

    def __init__(self, smpl_mean_params=SMPL_MEAN_PARAMS, pretrained=True, assets=None, numerics='bf16x3',
                 return_view='vis', cam_ckpt='data/pretrained_model/camcalib_sa_biased_l2.ckpt'):
        super().__init__()
        assert cfg.MODEL.PyMAF.BACKBONE == 'vitpose', 'only the vitpose branch is on the hot path (SURVEY 2.1)'
        assert cfg.MODEL.PyMAF.N_ITER == 3
        if assets is None:
            assets = load_assets(smpl_mean_params)
        self.numerics = numerics
        self.return_view = return_view
        self.feature_extractor = get_vitpose_encoder(cfg, numerics=numerics)
        self.inplanes = 768
        self.deconv_with_bias = cfg.RES_MODEL.DECONV_WITH_BIAS
        self.deconv_layers = self._make_deconv_layer(cfg.RES_MODEL.NUM_DECONV_LAYERS, cfg.RES_MODEL.NUM_DECONV_FILTERS,
                                                     cfg.RES_MODEL.NUM_DECONV_KERNELS)
        dmap = torch.matmul(assets['Dmap1'].float(), assets['Dmap0'].float())
        self.maf_extractor = nn.ModuleList([MAF_Extractor(Dmap=dmap.clone()) for _ in range(cfg.MODEL.PyMAF.N_ITER)])
        ma_feat_len = 67 * cfg.MODEL.PyMAF.MLP_DIM[-1]
        xv, yv = torch.meshgrid([torch.linspace(-1, 1, 7), torch.linspace(-1, 1, 9)], indexing='ij')   # whmr.py:341-347
        self.register_buffer('points_grid', torch.stack([xv.reshape(-1), yv.reshape(-1)]).unsqueeze(0))
        grid_feat_len = 7 * 9 * cfg.MODEL.PyMAF.MLP_DIM[-1]
        smpl = SMPL(arrays=assets['smpl'], marker_ids=assets['ssm'])                  # one instance, shared (3 key prefixes)
        self.regressor = nn.ModuleList([Regressor(grid_feat_len if i == 0 else ma_feat_len, smpl_mean_params, smpl, assets)
                                        for i in range(3)])
        self.transformer = nn.ModuleList()                                            # dead in the reference (whmr.py:362-394)
        if cfg.MODEL.PyMAF.AUX_SUPV_ON:
            self.dp_head = IUV_predict_layer(feat_dim=256)
        self.conv = nn.Sequential(nn.Conv2d(256, 64, 7, 3, 0, bias=False), nn.Conv2d(64, 5, 7, 2, 0, bias=False))
        td = _Holder()                                                                # timm Block(dim=216, heads=2) names
        _ln_params(td, 'norm1', 216)
        td.attn = _Holder()
        _linear_params(td.attn, 'qkv', 648, 216, bias=False)
        _linear_params(td.attn, 'proj', 216, 216)
        _ln_params(td, 'norm2', 216)
        td.mlp = _Holder()
        _linear_params(td.mlp, 'fc1', 864, 216)
        _linear_params(td.mlp, 'fc2', 216, 864)
        self.transformer_decoder = td
        self.avgpool = nn.AvgPool1d(kernel_size=5)
        self.est_Tz = nn.Sequential(nn.Linear(18 * 12, 12), nn.Linear(12, 1), nn.BatchNorm1d(1), nn.Sigmoid())
        self.cam_model = CameraRegressorNetwork(backbone='resnet50', num_fc_layers=1, num_fc_channels=1024)
        self.cam_model.numerics = numerics
        if pretrained and cam_ckpt and os.path.exists(cam_ckpt):
            sd = torch.load(cam_ckpt, map_location='cpu')['state_dict']
            self.cam_model.load_state_dict({k.replace('model.', '', 1): v for k, v in sd.items()}, strict=True)
        self.global_orient = Global_Orient_Regressor(smpl_mean_params, assets)
        self._cache = _Cache()
        self._init_cache = None
        self.overlap_camera = True          # cam_model on a side stream beside the backbone (joined before the global-orientation head)
        self.camera_launch = os.environ.get('WHMR_CAM_LAUNCH', 'early')    # where the camera branch's launches are issued: 'early' | 'vit' | 'loop' (A/B, see _forward_eval)
        self.overlap_tz = True              # Tz head on a side stream beside the regressor loop (its outputs finalized after the join)
        self.smpl_offsets_x3 = os.environ.get('WHMR_SMPL_X3', '1') != '0'   # A/B: 0 keeps the exact-f32 offsets in every numerics mode
        self.compose_tz = os.environ.get('WHMR_COMPOSE_TZ', '1') != '0'   # inference: the two Tz-head convolutions as ONE composed k25 / s6 convolution (False: the two-convolution form)
        self._tz_gemm_kw = {}               # explicit tile / split-K of the composed convolution's GEMM (A/B probes)
        self._tz_ones = {}
        # torch's DistributedDataParallel (the reference's wrap, core/trainer.py:84-86) re-broadcasts every module buffer before each forward.  Here
        # that is > 100 MB of CONSTANT tables per step (SMPL arrays, the dense down-sampling matrices, mean parameters, the frozen camera network's
        # BatchNorm statistics) in ~600 small copies -- and each in-place copy bumps the buffer's version, which invalidates the operand forms cached
        # per version (CSR of the down-sampling matrices, folded tables): measured 33.6 vs 22.9 ms per step at world 1.  DDP's own opt-out
        # (`_ddp_params_and_buffers_to_ignore`, read from the wrapped module) keeps them out of the broadcast; the BatchNorm statistics that DO change
        # (3 deconv layers + the Tz head) stay in.  Every rank builds these tables from the same files / checkpoint, so nothing is lost.
        self._ddp_params_and_buffers_to_ignore = [n for n, _ in self.named_buffers()
                                                  if n.startswith('cam_model.') or not ('running_' in n or 'num_batches_tracked' in n)]
        self.eval()

    def _make_deconv_layer(self, num_layers, num_filters, num_kernels):
        """whmr.py:459-501: parameter containers for 3x (ConvTranspose2d k4 s2 p1, BatchNorm2d, ReLU)."""
        assert num_layers == len(num_filters) == len(num_kernels)
        layers = []
        for i in range(num_layers):
            assert num_kernels[i] == 4, 'the sub-pixel GEMM decomposition is built for kernel 4 / stride 2 / pad 1'
            layers += [nn.ConvTranspose2d(self.inplanes, num_filters[i], 4, 2, 1, 0, bias=self.deconv_with_bias),
                       nn.BatchNorm2d(num_filters[i], momentum=BN_MOMENTUM), nn.ReLU(inplace=True)]
            self.inplanes = num_filters[i]
        return nn.Sequential(*layers)

    # ------------------------------------------------------------------ derived operands
    @property
    def _dt(self):
        return torch.bfloat16 if self.numerics == 'bf16' else torch.float32

    def _deconv_operands(self, i):
        """4 sub-pixel phase matrices [Cout, 4*Cin] (k = (a, b, ci)) with the eval BatchNorm scale folded in, + shift."""
        ct, bn = self.deconv_layers[3 * i], self.deconv_layers[3 * i + 1]
        srcs = [ct.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var] + ([ct.bias] if ct.bias is not None else [])

        def build():
            w = ct.weight.detach().float()                                            # [Cin, Cout, 4, 4]
            s = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
            t = bn.bias.detach() - bn.running_mean * s
            if ct.bias is not None:
                t = t + ct.bias.detach() * s
            phases = []
            for py in range(2):
                for px in range(2):
                    taps = [w[:, :, 3 - py - 2 * a, 3 - px - 2 * b] for a in range(2) for b in range(2)]   # each [Cin, Cout]
                    wp = torch.stack(taps, 0).permute(2, 0, 1).reshape(w.shape[1], -1) * s[:, None]
                    phases.append(wp.contiguous())
            if self.numerics == 'bf16x3':
                # per tap [W_hi | W_hi | W_lo] against the activation's [x_hi | x_lo | x_hi] (L.split3): one bf16 launch, three split products
                ph = torch.stack(phases, 0).reshape(4, w.shape[1], 4, w.shape[0])
                return L.split3_weight(ph).reshape(4, w.shape[1], 12 * w.shape[0]).contiguous(), t.float().contiguous()
            if self._dt == torch.float32:
                return phases, t.float().contiguous()
            return L.cast_bf16(torch.stack(phases, 0).contiguous()), t.float().contiguous()        # [4, Cout, 4*Cin]
        return self._cache.get(('deconv', i, self.numerics), srcs, build)

    def _tz_operands(self):
        c0, c1 = self.conv[0].weight, self.conv[1].weight

        def build():
            w0 = c0.detach().permute(0, 2, 3, 1)                                             # [64, ky, kx, ci]
            if self.numerics == 'bf16x3':
                # N = 64 makes this convolution A-traffic bound, so the W_lo product goes into 64 extra OUTPUT columns instead of a third K slice:
                # activation [x_hi | x_lo] (2 C per pixel) against rows n < 64: [W_hi | W_hi] (hi.hi + lo.hi), rows 64 + n: [W_lo | 0] (hi.lo);
                # tz_conv1 adds the column halves as it reads them.  Then the chunk-major K order of the bf16 kernel.
                hi, lo = L.split_bf16(w0.float().contiguous())
                w2 = torch.cat([torch.cat([hi, hi], -1), torch.cat([lo, torch.zeros_like(lo)], -1)], 0)        # [128, ky, kx, 2 ci]
                n, kh, kw, ci = w2.shape
                w0 = w2.reshape(n, kh * kw, ci // 64, 64).permute(0, 2, 1, 3).reshape(n, -1).contiguous()
            elif self._dt == torch.float32:
                w0 = w0.reshape(c0.shape[0], -1).contiguous()                              # [64, (ky,kx,ci)]
            else:   # chunk-major K order (epi_flags bit 3): (ci chunk of 64, ky, kx, ci in chunk) -- window overlap re-read from cache
                n, kh, kw, ci = w0.shape
                w0 = L.cast_bf16(w0.reshape(n, kh * kw, ci // 64, 64).permute(0, 2, 1, 3).reshape(n, -1).contiguous())
            w1 = c1.detach().float().permute(0, 2, 3, 1).reshape(c1.shape[0], 49, 64).contiguous()        # [5, (ky,kx), ci] fp32
            bn = self.est_Tz[2]
            bn4 = torch.stack([bn.weight.detach()[0], bn.bias.detach()[0], bn.running_mean[0], bn.running_var[0]]).float().contiguous()
            return w0, w1, bn4
        bn = self.est_Tz[2]
        return self._cache.get(('tz', self.numerics), [c0, c1, bn.weight, bn.bias, bn.running_mean, bn.running_var], build)

    def _tz_composed_operands(self):
        """The two Tz-head convolutions (whmr.py:418-421: Conv2d(256, 64, k7, s3) -> Conv2d(64, 5, k7, s2), no bias, nothing in between) composed into
        ONE Conv2d(256, 5, k25, s6): Wc[o, ci, A, B] = sum_{c1, 3u + a = A, 3v + b = B} w1[o, c1, u, v] w0[c1, ci, a, b], built in fp64.  Stored as the weight
        of the space-to-depth implicit GEMM of ``_tz_head``: rows n = (jA, jB, o) (125, padded to 128), columns k = (q, p, ci) with A = q + 6 jA,
        B = p + 6 jB (zero where A or B > 24).  Inference views only -- the training graph keeps the two convolutions (their weights get gradients)."""
        c0, c1 = self.conv[0].weight, self.conv[1].weight

        def build():
            Ci = c0.shape[1]
            g = compose_tz_weights(c0.detach(), c1.detach())                             # [128, 36 Ci] fp32
            if self.numerics == 'bf16x3':
                # map operand [x_hi | x_lo] per pixel (2 C channels); rows n < 128: [W_hi | W_hi] (hi.hi + lo.hi), rows 128 + n: [W_lo | 0] (hi.lo)
                hi, lo = L.split_bf16(g)
                hi, lo = hi.view(128, 36, Ci), lo.view(128, 36, Ci)
                g = torch.cat([torch.cat([hi, hi], -1), torch.cat([lo, torch.zeros_like(lo)], -1)], 0).reshape(256, -1).contiguous()
            elif self._dt != torch.float32:
                g = L.cast_bf16(g)
            bn = self.est_Tz[2]
            bn4 = torch.stack([bn.weight.detach()[0], bn.bias.detach()[0], bn.running_mean[0], bn.running_var[0]]).float().contiguous()
            return g, bn4
        bn = self.est_Tz[2]
        return self._cache.get(('tzc', self.numerics), [c0, c1, bn.weight, bn.bias, bn.running_mean, bn.running_var], build)

    # ------------------------------------------------------------------ stages
    def _deconv(self, i, x_nhwc, x_split=None):
        """-> (map [B, 2H, 2W, Cout], its split-bf16 operand form or None).  ``x_split``: the [hi | lo | hi] operand form of ``x_nhwc`` that the
        producing stage's epilogue left (bf16x3 only); handed over EXPLICITLY by the caller -- it used to ride as an attribute on the map tensor,
        which nothing tied to the map's contents."""
        B, H, W, Cin = x_nhwc.shape
        phases, shift = self._deconv_operands(i)
        Cout = phases[0].shape[0]
        out = torch.empty(B, 2 * H, 2 * W, Cout, dtype=self._dt, device=x_nhwc.device)
        if self.numerics == 'bf16x3':          # fp32 map in, fp32 map out; the bf16 kernel on the K-concatenated split operands (Cin' = 3 Cin)
            # the split-bf16 operand form of the OUTPUT ([hi | lo | hi], what the next deconv stage / the Tz convolution multiplies) leaves the
            # epilogue next to the fp32 map (epi_flags bit 8): the split pass over each map (0.45 ms per forward at batch 64) is gone
            xs = x_split if x_split is not None else L.split3(x_nhwc)
            parts = 2 if i == len(self.deconv_layers) // 3 - 1 else 3      # the last map feeds the Tz convolution only: [hi | lo]
            out_s3 = torch.empty(B, 2 * H, 2 * W, parts * Cout, dtype=torch.bfloat16, device=x_nhwc.device)
            L.gemm(xs, phases, out, bias=shift, act=L.ACT_RELU, split3_out=out_s3, split_parts=parts,
                   conv=dict(IH=H, IW=W, Cin=3 * Cin, OH=H, OW=W, KW=2, SH=1, SW=1, PH=1, PW=1),
                   scatter=dict(c_off=0, osb=4 * H * W * Cout, osy=4 * W * Cout, osx=2 * Cout),
                   phases=dict(cy=2 * W * Cout, cx=Cout))
            return out, out_s3                 # parts == 3: [hi | lo | hi] for the next deconv stage; 2: [hi | lo] for the Tz convolution
        if self._dt != torch.float32:          # all 4 sub-pixel phases in one launch (4x the tiles to fill the CUs)
            L.gemm(x_nhwc, phases, out, bias=shift, act=L.ACT_RELU,
                   conv=dict(IH=H, IW=W, Cin=Cin, OH=H, OW=W, KW=2, SH=1, SW=1, PH=1, PW=1),
                   scatter=dict(c_off=0, osb=4 * H * W * Cout, osy=4 * W * Cout, osx=2 * Cout),
                   phases=dict(cy=2 * W * Cout, cx=Cout))
            return out, None
        for py in range(2):
            for px in range(2):
                L.gemm(x_nhwc, phases[py * 2 + px], out, bias=shift, act=L.ACT_RELU,
                       conv=dict(IH=H, IW=W, Cin=Cin, OH=H, OW=W, KW=2, SH=1, SW=1, PH=1 - py, PW=1 - px),
                       scatter=dict(c_off=(py * 2 * W + px) * Cout, osb=4 * H * W * Cout, osy=4 * W * Cout, osx=2 * Cout))
        return out, None

    def _tz_head(self, f_nhwc, f_split=None):
        """whmr.py:567-577.  ``f_split`` (bf16x3): the [hi | lo] operand form of the map, as the last deconv stage's epilogue wrote it."""
        B, H, W, C = f_nhwc.shape
        dev = f_nhwc.device
        assert (C, self.conv[1].weight.shape[0], self.conv[0].weight.shape[0]) == (256, 5, 64)
        H1, W1 = (H - 7) // 3 + 1, (W - 7) // 3 + 1
        if self.compose_tz and W % 6 == 0:
            return self._tz_tokens_tail(self._tz_tokens_composed(f_nhwc, f_split), B, dev, self._tz_composed_operands()[1])
        w0, w1, bn4 = self._tz_operands()
        y0 = torch.empty(B, H1, W1, 128 if self.numerics == 'bf16x3' else 64, dtype=self._dt, device=dev)     # NHWC, mode dtype (feeds the 2nd conv)
        if self.numerics == 'bf16x3':
            fs = f_split if f_split is not None else torch.cat(L.split_bf16(f_nhwc), -1).contiguous()
            L.gemm(fs, w0, y0.view(-1, 128), conv=dict(IH=H, IW=W, Cin=2 * C, OH=H1, OW=W1, KW=7, SH=3, SW=3, PH=0, PW=0, chunk_major=True))
        else:
            L.gemm(f_nhwc, w0, y0.view(-1, 64), conv=dict(IH=H, IW=W, Cin=C, OH=H1, OW=W1, KW=7, SH=3, SW=3, PH=0, PW=0,
                                                          chunk_major=self._dt != torch.float32))
        H2, W2 = (H1 - 7) // 2 + 1, (W1 - 7) // 2 + 1
        D = H2 * W2
        t = torch.empty(B * 5, D, dtype=torch.float32, device=dev)                  # == conv1(...).reshape(B, 5, -1), whmr.py:571
        L.tz_conv1(y0, w1, t.view(B, 5, D))                                           # N = 5 output channels: one wave per pixel
        return self._tz_tokens_tail(t, B, dev, bn4)

    def _tz_tokens_composed(self, f_nhwc, f_split):
        """whmr.py:567-571 as ONE convolution: tokens [B * 5, 18 * 12] = Conv2d(256, 5, k25, s6)(map), see ``_tz_composed_operands``.  The map
        [B, H, W, C] is read as [B, H, W / 6, 6 C]: an implicit GEMM with a 6 x 1 kernel at stride 6 x 1 (rows (b, Y, m), K = (q, p, ci)) that touches
        every map byte once, then ``L.tz_fold`` adds the 25 shifted partial sums of each output pixel."""
        B, H, W, C = f_nhwc.shape
        g, _ = self._tz_composed_operands()
        H1, W1 = (H - 7) // 3 + 1, (W - 7) // 3 + 1
        H2, W2 = (H1 - 7) // 2 + 1, (W1 - 7) // 2 + 1
        OHp, OWp = (H + 5) // 6, W // 6
        assert OHp >= H2 + 4 and OWp >= W2 + 4
        x3 = self.numerics == 'bf16x3'
        if x3:
            fs = f_split if f_split is not None else torch.cat(L.split_bf16(f_nhwc), -1).contiguous()
            a, Cp = fs, 2 * C
        else:
            a, Cp = f_nhwc, C
        # N = 128 (256) columns only: the chooser's 176 (88) tiles leave CUs idle while each walks 144 (288) K steps alone.  Two raw split-K planes
        # that tz_fold adds (tools/r6_tz_probe.py, batch 64: bf16 174 -> 124 us on the 128 x 128 tile, bf16x3 444 -> 270 us on the 192 x 256 tile);
        # small batches keep the chooser (it slices K itself when the grid is small)
        kw = dict(self._tz_gemm_kw)
        M = B * OHp * OWp
        if not kw and self.numerics != 'fp32' and M >= 8192:
            kw = dict(tile=192 if x3 else 64, raw_splits=2)
        ns = kw.get('raw_splits') or 1
        P = torch.empty(ns, M, g.shape[0], dtype=torch.float32, device=f_nhwc.device)
        L.gemm(a.view(B, H, OWp, 6 * Cp), g, P if ns > 1 else P[0], conv=dict(IH=H, IW=OWp, Cin=6 * Cp, OH=OHp, OW=OWp, KW=1, SH=6, SW=1, PH=0, PW=0), **kw)
        t = torch.empty(B * 5, H2 * W2, dtype=torch.float32, device=f_nhwc.device)
        return L.tz_fold(P, t, B, OHp, OWp, H2, W2, halves=2 if x3 else 1, nsplit=ns, split_stride=M * g.shape[0])

    def _tz_tokens_tail(self, t, B, dev, bn4):
        """whmr.py:572-577: the timm Block over the 5 tokens, mean over tokens, est_Tz."""
        D = t.shape[1]
        td = self.transformer_decoder
        h = torch.empty_like(t)
        qkv = torch.empty(B * 5, 3 * D, dtype=torch.float32, device=dev)
        att = torch.empty_like(t)
        hid = torch.empty(B * 5, td.mlp.fc1.weight.shape[0], dtype=torch.float32, device=dev)
        L.layernorm(t, td.norm1.weight, td.norm1.bias, h, 1e-5)
        L.gemm(h, td.attn.qkv.weight.detach(), qkv, bias=td.attn.qkv.bias)
        L.attention(qkv, att, B, 5, 2, D // 2, (D // 2) ** -0.5)
        L.gemm(att, td.attn.proj.weight.detach(), t, bias=td.attn.proj.bias.detach(), residual=t)
        L.layernorm(t, td.norm2.weight, td.norm2.bias, h, 1e-5)
        L.gemm(h, td.mlp.fc1.weight.detach(), hid, bias=td.mlp.fc1.bias.detach(), act=L.ACT_GELU)
        L.gemm(hid, td.mlp.fc2.weight.detach(), t, bias=td.mlp.fc2.bias.detach(), residual=t)
        tz = torch.empty(B, dtype=torch.float32, device=dev)
        L.tz_tail(t.view(B, 5, D), self.est_Tz[0].weight.detach(), self.est_Tz[0].bias.detach(), self.est_Tz[1].weight.detach(),
                  self.est_Tz[1].bias.detach(), bn4, self.est_Tz[2].eps, tz)
        return tz

    def _init_mesh(self, B, J_regressor, with_aux):
        """whmr.py:548-550 recomputes a constant every call: computed once at batch 1 and expanded."""
        reg = self.regressor[0]
        expand = lambda d: {k: (v.expand(B, *v.shape[1:]) if torch.is_tensor(v) else v) for k, v in d.items()}
        x1 = torch.empty(1, 1, device=self.points_grid.device)
        if J_regressor is not None:
            return expand(reg.forward_init(x1, J_regressor=J_regressor, with_aux=with_aux))
        key = (reg.init_pose.device, reg.init_pose._version, reg.init_shape._version, reg.init_cam._version, with_aux)
        if self._init_cache is None or self._init_cache[0] != key:
            self._init_cache = (key, reg.forward_init(x1, with_aux=with_aux))
        return expand(self._init_cache[1])

    def _grid_points(self, B):
        """the stage-0 sampling grid as [B, 63, 2] points (whmr.py:586-590: points_grid expanded and transposed): a constant, built once per batch size"""
        g = self.points_grid
        key = (B, g.device, g._version, g.data_ptr())
        ent = self.__dict__.get('_grid_pts')
        if ent is None or ent[0] != key:
            ent = self.__dict__['_grid_pts'] = (key, g.expand(B, -1, -1).transpose(1, 2).contiguous())
        return ent[1]

    def _tz_placeholder(self, B, dev):
        t = self._tz_ones.get((dev, B))
        if t is None:
            t = self._tz_ones[(dev, B)] = torch.ones(B, dtype=torch.float32, device=dev)
        return t

    @staticmethod
    def _camera_stream(dev, tag=None):
        st = _CAM_STREAMS.get((dev, tag))          # override hook of the stream experiments (tools/r6_cu_mask_probe.py, r6_stream_priority_probe.py)
        return st if st is not None else L.side_stream(dev, 0 if tag == 'tz' else 1)      # the package's shared pool (module-level, not in the module: deepcopy / pickling)

    @torch.no_grad()
    def _camera(self, full_x, cam_rotmat, B, dev):
        """whmr.py:509-524: camera-calibration head on the full image -> (cam_rotmat, render_rotmat); detached in the reference too."""
        render_rotmat = None
        if cam_rotmat is None:
            if full_x is not None:
                # demo/tester.py:161 replicates the full image once per person; a batch-1 full_x is accepted here and its
                # camera prediction broadcast (identical result, the ~80 GFLOP ResNet-50 runs once per image -- SURVEY 8f N1)
                pred, _ = self.cam_model(full_x)
                # soft-argmax -> (pitch, roll) -> R([pitch, 0, roll]), R([-pitch, 0, roll]) (whmr.py:513-522) in one launch
                # (convert_preds_to_angles + 2 x batch_euler2matrix were ~160 element-wise launches)
                # the head's three Linear layers run as ONE GEMM into a [Bf, 3 D] matrix that forward() also publishes as `fused_logits`;
                # it is taken only when it provably IS what the three views slice (same storage, offsets 0 / D / 2 D), else the views are joined
                D = pred[0].shape[1]
                logits = getattr(self.cam_model, 'fused_logits', None)
                if not (logits is not None and logits.dim() == 2 and logits.shape == (pred[0].shape[0], 3 * D) and logits.is_contiguous()
                        and logits.dtype == torch.float32 and all(pr.data_ptr() == logits.data_ptr() + 4 * i * D and pr.stride() == (3 * D, 1) and pr.shape[1] == D
                                                                  for i, pr in enumerate(pred))):
                    logits = torch.cat([pr.float() for pr in pred], 1)
                from .cam_model import PITCH_RANGE, ROLL_RANGE
                cam_rotmat, render_rotmat = L.cam_head(logits, pred[0].shape[1], PITCH_RANGE, ROLL_RANGE, B)
            else:
                cam_rotmat = torch.eye(3, device=dev).unsqueeze(0).expand(B, -1, -1).float()
        if render_rotmat is None:
            render_rotmat = cam_rotmat
        return cam_rotmat, render_rotmat

    # ------------------------------------------------------------------ forward
    def forward(self, x, meta_masks=None, center=None, scale=None, bbox_height=None, orig_shape=None, bbox_info=None,
                is_train=False, J_regressor=None, full_x=None, cam_rotmat=None, view=None):
        if not x.is_cuda:
            raise RuntimeError('whmr_amd.WHMR runs on a HIP device only (no CPU fallback)')
        if is_train:       # training graph: HIP forward kernels + hand-written HIP backward behind autograd nodes (whmr_amd.train)
            from ..train.whmr_train import whmr_forward_train
            return whmr_forward_train(self, x, center, scale, bbox_height, orig_shape, bbox_info, J_regressor=J_regressor,
                                      full_x=full_x, cam_rotmat=cam_rotmat)
        with torch.no_grad():
            return self._forward_eval(x, center, scale, bbox_height, orig_shape, bbox_info, J_regressor, full_x, cam_rotmat, view)

    def _forward_eval(self, x, center, scale, bbox_height, orig_shape, bbox_info, J_regressor, full_x, cam_rotmat, view):
        view = view or self.return_view
        with_aux = view == 'train'
        B, dev = x.shape[0], x.device
        # SMPL's pose-corrective offsets: split-bf16 MFMA operands in the bf16 / bf16x3 numerics (~3e-7 of a vertex), the exact f32 chain in fp32
        self.regressor[0].smpl.offsets_x3 = self.numerics != 'fp32' and self.smpl_offsets_x3
        # The camera-calibration ResNet-50 (whmr.py:509-522) only feeds the global-orientation head at the very end (whmr.py:630): it runs on a
        # SIDE stream beside the backbone / deconvs / regressor loop and is joined just before that head.  Its few-tile launches slot into the CUs
        # the big GEMM grids leave idle (tile-grid tails); under GraphedForward the fork / join become two branches of the captured graph.
        # ``camera_launch`` (A/B switch, round 6): 'early' issues the branch first (default), 'vit' right behind the backbone's launches, 'loop' just
        # before the join -- its only dependency is the start of the forward either way.  Measured under the HIP-graph replay at batch 64
        # (profiles/r06_experiments_that_did_not_pay.txt): early 4.30 ms, vit 4.87, loop 4.89 (the graph then folds the branch onto the regressor
        # loop's queue); without any camera work 3.97, the camera chain alone 0.79 -- i.e. 'early' already hides 0.46 of its 0.79 ms, and so does a
        # SEPARATE camera graph replayed on a side stream (4.34): what remains is the chain's own CU time and power beside the backbone.
        side, cam_pending = None, None
        if full_x is not None and cam_rotmat is None and self.overlap_camera:
            main = torch.cuda.current_stream(dev)
            side = self._camera_stream(dev)
            if self.camera_launch == 'early':
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    cam_rotmat, render_rotmat = self._camera(full_x, None, B, dev)
            else:
                cam_pending = torch.cuda.Event()
                cam_pending.record(main)
        else:
            cam_rotmat, render_rotmat = self._camera(full_x, cam_rotmat, B, dev)

        def launch_camera():
            side.wait_event(cam_pending)
            with torch.cuda.stream(side):
                return self._camera(full_x, None, B, dev)

        # backbone (tokens are NHWC already) -> deconv pyramid in NHWC
        vit = self.feature_extractor.backbone
        tok, (_, Hp, Wp) = vit.forward_tokens(x)
        if cam_pending is not None and self.camera_launch == 'vit':
            cam_rotmat, render_rotmat = launch_camera()
            cam_pending = None
        s_feat = tok.view(B, Hp, Wp, vit.embed_dim)
        f = s_feat if self._dt == torch.float32 else L.cast_bf16(s_feat)
        fmaps = []
        # Regressor iteration i only needs feature map i, and the Tz head (7x7 s3 conv over the last map + the timm Block, ~0.45 ms at batch 64)
        # only enters a stage's OUTPUTS (focal length, pred_cam_t, kp_2d_w: whmr.py:147-173), never the next stage's input.  So after the first
        # deconv stage the heavy chain (deconv 2, deconv 3, Tz head) moves to a side stream and the regressor loop -- a chain of small
        # latency-bound launches -- runs beside it on the main stream, waiting for map i before iteration i; the loop's tail kernels see a
        # placeholder Tz and the Tz-dependent outputs of the stages a view returns are (re)computed by one small launch each after the join.
        tz_side, map_ready = None, [None, None, None]
        if self.overlap_tz and B >= 4:          # (one or two crops: every launch is latency-bound, the extra finalize launch costs more than it hides)
            f, fsplit = self._deconv(0, f)
            fmaps.append(f)
            main_tz = torch.cuda.current_stream(dev)
            tz_side = self._camera_stream(dev, 'tz')
            tz_side.wait_stream(main_tz)
            with torch.cuda.stream(tz_side):
                for i in (1, 2):
                    f, fsplit = self._deconv(i, f, fsplit)
                    fmaps.append(f)
                    map_ready[i] = torch.cuda.Event()
                    map_ready[i].record(tz_side)
                Tz_true = self._tz_head(fmaps[-1], fsplit)
            Tz = self._tz_placeholder(B, dev)
        else:
            fsplit = None
            for i in range(3):
                f, fsplit = self._deconv(i, f, fsplit)
                fmaps.append(f)
            Tz = Tz_true = self._tz_head(fmaps[-1], fsplit)
        for i in range(3):
            self.maf_extractor[i].im_feat = fmaps[i].permute(0, 3, 1, 2)              # logical NCHW view (whmr.py:564)
        stage_state = []

        smpl_output = self._init_mesh(B, J_regressor, with_aux)
        outs = [smpl_output]
        body_feat = None
        center, scale, bbox_height = center.float().contiguous(), scale.float(), bbox_height.float().contiguous()
        orig_shape, bbox_info = orig_shape.float().contiguous(), bbox_info.float().contiguous()
        # one input buffer per stage, allocated up front: stage i's tail launch writes stage i+1's state columns (whmr.py:105,119)
        Fs = [self.regressor[i].fc1.in_features - 234 for i in range(3)]
        xcs = [torch.empty(B, Fs[i] + 234, dtype=torch.float32, device=dev) for i in range(3)]
        for i in range(3):                                                            # whmr.py:580-627
            reg, ext = self.regressor[i], self.maf_extractor[i]
            if map_ready[i] is not None:
                torch.cuda.current_stream(dev).wait_event(map_ready[i])               # feature map i comes from the side stream
            cam, shp, pose = smpl_output['pred_cam'], smpl_output['pred_shape'], smpl_output['rotmat']
            ext.cam = cam
            xc = xcs[i]
            if i == 0:
                pts = self._grid_points(B)
                ext.sampling(pts, out=xc, want_point_feat=False)
            else:
                ext(smpl_output['markers'], cam=cam, out=xc, want_point_feat=False)   # cam: a strided column slice of the state (no copy)
            smpl_output, body_feat = reg(None, bbox_info, Tz, orig_shape, center, scale, bbox_height, pose, shp, cam,
                                         is_train=False, n_iter=1, J_regressor=J_regressor, with_aux=with_aux, xc=xc,
                                         xc_next=(xcs[i + 1], Fs[i + 1]) if i < 2 else None, state_ready=i > 0)
            outs.append(smpl_output)
            stage_state.append(reg._last_stage)

        if cam_pending is not None:                                                   # camera_launch == 'loop': issued behind every other launch
            cam_rotmat, render_rotmat = launch_camera()
        if side is not None:                                                          # join: the camera rotation is needed from here on
            main.wait_stream(side)
            if not torch.cuda.is_current_stream_capturing():                          # (a capture's private pool never recycles)
                for t in (cam_rotmat, render_rotmat):
                    t.record_stream(main)                                             # allocated on the side stream, read (and returned) on the main one
        # whmr.py:630-654: state columns, the collapsed head, then Gram-Schmidt + angle-axis + the two concatenations as one launch
        r9 = self.global_orient(body_feat, cam_rotmat, smpl_output['rotmat'], False, xc=xc, raw=True)
        g_pose, g_rotmat = L.orient_tail(r9, smpl_output['pose'].contiguous(), smpl_output['rotmat'].contiguous())
        g = self.regressor[0].smpl.run(smpl_output['pred_shape'], g_rotmat)
        g_joints = g.joints
        if J_regressor is not None:
            g_joints = h36m_joints(g.vertices, J_regressor)
        if tz_side is not None:                                                       # join the Tz head LAST (the global-orientation head above does not need it
        # and runs beside the heavy chain too); finalize the stages this view returns
            main_tz.wait_stream(tz_side)
            if not torch.cuda.is_current_stream_capturing():
                for t in (Tz_true, fmaps[1], fmaps[2]):                                # allocated on the side stream, read (or returned) on the main one
                    t.record_stream(main_tz)
            for i in (range(3) if view == 'train' else (2,)):
                st, aa, joints = stage_state[i]
                theta, kp_2d, kp_w, cam_t, focal = L.regressor_post(st, aa, joints, Tz_true, bbox_height, center, orig_shape, 1000.0,
                                                                    float(cfg.IMG_RES.WIDTH), float(cfg.IMG_RES.HEIGHT))
                outs[i + 1].update(theta=theta, kp_2d=kp_2d, kp_2d_w=kp_w, pred_cam_t=cam_t, focal_length=focal)
        g_out = {'global_pose': g_pose, 'global_shape': smpl_output['pred_shape'], 'global_rotmat': g_rotmat,
                 'global_kp_3d': g_joints, 'global_verts': g.vertices}
        if view == 'eval':
            return {'global_output': g_out}, None
        if view == 'train':
            vis_feat = [s_feat.permute(0, 3, 1, 2)] + [m.permute(0, 3, 1, 2) for m in fmaps]
            return {'smpl_out': outs, 'dp_out': [], 'dpth_out': [], 'global_output': g_out}, vis_feat
        return {'local_smpl_vertices': smpl_output['verts'], 'smpl_vertices': g.vertices,
                'pred_cam_t': smpl_output['pred_cam_t'], 'focal_length': smpl_output['focal_length'],
                'cam_rotmat': cam_rotmat, 'render_rotmat': render_rotmat, 'shape': smpl_output['pred_shape'],
                'global_pose': g_pose, 'local_pose': smpl_output['pose']}


THIRD_PARTY_KEY_PARTS = ('.smpl.', '.vertex_joint_selector.')     # smplx / pare internals inside regressor.N; 'transformer.' = HF leftovers


def _is_third_party_key(k):
    return any(t in k for t in THIRD_PARTY_KEY_PARTS) or k.startswith('transformer.')


def load_reference_state_dict(model, state_dict, verbose=True):
    """Load a reference checkpoint's ``ckpt['model']`` (demo/tester.py:64-65, utils/saver.py:26-64) into ``model``.

    Own-code keys are identical and are loaded strictly.  Keys that belong to third-party classes in the
    reference (``regressor.N.smpl.*`` from smplx, ``regressor.N.vertex_joint_selector.*``) are matched by name where this
    package has the same buffer and otherwise reported, never silently dropped.  Returns (missing, unexpected, skipped).
    """
    own = model.state_dict()
    load, skipped = {}, []
    for k, v in state_dict.items():
        if k in own and tuple(own[k].shape) == tuple(v.shape):
            load[k] = v
        elif '.vertex_joint_selector.' in k or '.smpl.' in k or k.startswith('transformer.'):
            skipped.append(k)                      # smplx / pare internals without a counterpart (e.g. betas, faces_tensor variants)
        else:
            raise KeyError('reference key %s %s has no counterpart of that shape' % (k, tuple(v.shape)))
    res = model.load_state_dict(load, strict=False)
    missing = [k for k in res.missing_keys if '.smpl.' not in k]
    if missing:
        raise KeyError('own-code keys missing from the checkpoint: %s' % missing[:8])
    if verbose:
        print('loaded %d tensors; %d third-party keys skipped; %d smpl buffers kept from the SMPL model file'
              % (len(load), len(skipped), len(res.missing_keys)))
    return res.missing_keys, res.unexpected_keys, skipped


def _whmr_load_state_dict(self, state_dict, strict=True, assign=False):
    """``model.load_state_dict(torch.load(ckpt)['model'], strict=True)`` exactly as the reference's callers write it (demo/tester.py:64-65,
    evaluate/val_results.py:70).  A reference checkpoint carries, besides the own-code keys (identical here), tensors that
    live inside third-party classes there: ``regressor.N.smpl.*`` / ``regressor.N.vertex_joint_selector.*`` (smplx / pare) and possibly
    ``transformer.*``.  Those with a same-named, same-shaped buffer here are loaded; the others have no counterpart (this package reads the
    SMPL model from its own file) and are listed in ``self.ignored_checkpoint_keys`` instead of failing the strict check.  Conversely this
    package's ``.smpl.`` buffers that a checkpoint does not carry keep their values.  Everything else stays strict: a missing or
    shape-mismatched own-code key raises like nn.Module.load_state_dict would."""
    own = nn.Module.state_dict(self)
    load, ignored = {}, []
    for k, v in state_dict.items():
        if _is_third_party_key(k) and not (k in own and tuple(own[k].shape) == tuple(v.shape)):
            ignored.append(k)
        else:
            load[k] = v
    res = nn.Module.load_state_dict(self, load, strict=False, assign=assign)
    self.ignored_checkpoint_keys = ignored
    if strict:
        missing = [k for k in res.missing_keys if not _is_third_party_key(k)]
        if missing or res.unexpected_keys:
            raise RuntimeError('Error(s) in loading state_dict for WHMR: missing own-code keys %s, unexpected keys %s'
                               % (missing[:8], list(res.unexpected_keys)[:8]))
    return res


WHMR.load_state_dict = _whmr_load_state_dict


def whmr_net(smpl_mean_params, pretrained=True, **kwargs):
    """models/whmr.py:681-687."""
    return WHMR(smpl_mean_params, pretrained, **kwargs)
