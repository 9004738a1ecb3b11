"""CPU oracle for the training-mode stages (TEST INFRASTRUCTURE ONLY: imported by tests/, never by w-hmr_amd/).

Plain PyTorch fp32 restatements whose autograd is the reference gradient.  Parity: these are the torch modules the reference
itself instantiates (nn.ConvTranspose2d / nn.BatchNorm2d / nn.ReLU at models/whmr.py:488-498), called functionally.
"""
import torch
import torch.nn.functional as F


def deconv_bn_relu_train(x, weight, gamma, beta, running_mean=None, running_var=None, eps=1e-5, momentum=0.1):
    """models/whmr.py:488-498 in train mode: ConvTranspose2d(k4, s2, p1, output_padding 0, no bias) -> BatchNorm2d -> ReLU.
    x NCHW; running stats (if given) are updated in place like nn.BatchNorm2d does."""
    z = F.conv_transpose2d(x, weight, None, stride=2, padding=1, output_padding=0)
    z = F.batch_norm(z, running_mean, running_var, gamma, beta, training=True, momentum=momentum, eps=eps)
    return F.relu(z)


# ----------------------------------------------------------------------------------------------------------------------
# WHMR.forward in model.train() (models/whmr.py:503-678 as driven by core/trainer.py:410-470), functional over a state dict whose
# tensors may require grad.  Dropout is the identity here (the parity runs set p = 0 on both sides: the masks are random draws).
# Pinned against the imported reference by tests/golden/make_golden_train.py (per-stage outputs and every parameter gradient).
from . import geometry as G
from . import smpl as S
from . import whmr as OW
from .vit import vit_forward

TRAIN_LOSS_KEYS = ('rotmat', 'pred_shape', 'pred_cam', 'kp_2d', 'kp_2d_w', 'kp_3d', 'verts', 'sub_verts', 'temp_verts', 'focal_length')


def _bn_train(x, sd, p, stats):
    rm, rv = sd[p + 'running_mean'].detach().clone(), sd[p + 'running_var'].detach().clone()
    y = F.batch_norm(x, rm, rv, sd[p + 'weight'], sd[p + 'bias'], True, 0.1, OW.BN_EPS)
    if stats is not None:
        stats[p + 'running_mean'], stats[p + 'running_var'] = rm, rv
    return y


def tz_head_train(sd, s_feat, stats=None):
    """whmr.py:567-577, est_Tz's BatchNorm1d with batch statistics."""
    B = s_feat.shape[0]
    y = F.conv2d(s_feat, sd['conv.0.weight'], None, stride=3)
    y = F.conv2d(y, sd['conv.1.weight'], None, stride=2).reshape(B, 5, -1)
    y = OW.timm_block(sd, y, 'transformer_decoder.', 2).transpose(1, 2)
    y = F.avg_pool1d(y, 5).squeeze(-1)
    y = F.linear(F.linear(y, sd['est_Tz.0.weight'], sd['est_Tz.0.bias']), sd['est_Tz.1.weight'], sd['est_Tz.1.bias'])
    return 10.0 * torch.sigmoid(_bn_train(y, sd, 'est_Tz.2.', stats)).squeeze(-1)


def regressor_forward_train(sd, assets, i, x, bbox_info, Tz, orig_shape, center, scale, bbox_height, pose, shape, cam, stage=2):
    """whmr.py:102-209 with is_train=True, n_iter=1, J_regressor=None (no Gram-Schmidt; TRAIN.STAGE detach rules at :142-163)."""
    p = 'regressor.%d.' % i
    B = x.shape[0]
    x = torch.cat((x, bbox_info), dim=1)
    pose = pose.reshape(B, -1)
    xc = torch.cat([x, pose, shape, cam], 1)
    xc = F.linear(F.linear(xc, sd[p + 'fc1.weight'], sd[p + 'fc1.bias']), sd[p + 'fc2.weight'], sd[p + 'fc2.bias'])
    pose = F.linear(xc, sd[p + 'decpose.weight'], sd[p + 'decpose.bias']) + pose
    shape = F.linear(xc, sd[p + 'decshape.weight'], sd[p + 'decshape.bias']) + shape
    cam = F.linear(xc, sd[p + 'deccam.weight'], sd[p + 'deccam.bias']) + cam
    rotmat = pose.view(B, 24, 3, 3)
    verts, joints = S.smpl_forward(shape, rotmat, assets['smpl'])
    kp_2d = G.projection(joints if stage == 1 else joints.detach(), cam)
    focal = cam[:, 0].detach() * bbox_height * Tz / 2.0
    cam_center = orig_shape[:, [1, 0]] / 2.0
    cam_t = G.convert_pare_to_full_img_cam(cam.detach(), bbox_height, center, orig_shape[:, 1], orig_shape[:, 0], Tz)
    kp_w = G.perspective_projection(joints.detach() if stage == 1 else joints, torch.eye(3).unsqueeze(0), cam_t, focal, cam_center)
    kp_w = kp_w / cam_center.unsqueeze(1) - 1
    aa = G.rotation_matrix_to_angle_axis(rotmat.reshape(-1, 3, 3)).reshape(-1, 72)
    sub, temp, markers, smpl_j = OW._smpl_aux(verts, assets)
    out = {'theta': torch.cat([cam, shape, aa], dim=1), 'verts': verts, 'sub_verts': sub, 'temp_verts': temp, 'kp_2d': kp_2d,
           'kp_2d_w': kp_w, 'kp_3d': joints, 'smpl_kp_3d': smpl_j, 'rotmat': rotmat, 'pred_cam': cam, 'pred_cam_t': cam_t,
           'pred_shape': shape, 'pred_pose': pose, 'pose': aa, 'pelvis': smpl_j[:, :1], 'scale': scale, 'focal_length': focal,
           'markers': markers}
    return out, x


def global_orient_forward_train(sd, x, cam_rotmat, local_orient):
    """whmr.py:289-305 with is_train=True (Dropout p = 0 in the parity runs): no Gram-Schmidt (:303-304); the three loop passes are identical."""
    B = x.shape[0]
    lo = local_orient.reshape(B, -1)
    xc = torch.cat([x, G.rotmat_to_rot6d(cam_rotmat), lo], dim=1)
    xc = F.linear(xc, sd['global_orient.fc1.weight'], sd['global_orient.fc1.bias'])
    xc = F.linear(xc, sd['global_orient.fc2.weight'], sd['global_orient.fc2.bias'])
    return (F.linear(xc, sd['global_orient.decrot.weight'], sd['global_orient.decrot.bias']) + lo).reshape(-1, 1, 3, 3)


def global_output_train(sd, assets, smpl_out, body_feat, cam_rotmat):
    """whmr.py:630-654 in training mode, with its graph: the global-orientation head on the last stage's body_feat / root rotation (cam_rotmat is
    detached in the reference: it comes out of torch.no_grad(), whmr.py:509-524), angle-axis, SMPL of [global root | the other 23 rotations]."""
    g_rot = global_orient_forward_train(sd, body_feat, cam_rotmat.detach(), smpl_out['rotmat'][:, 0])
    g_aa = G.rotation_matrix_to_angle_axis(g_rot.reshape(-1, 3, 3)).reshape(-1, 3)
    g_rotmat = torch.cat([g_rot, smpl_out['rotmat'][:, 1:]], dim=1)
    g_verts, g_joints = S.smpl_forward(smpl_out['pred_shape'], g_rotmat, assets['smpl'])
    return {'global_pose': torch.cat([g_aa, smpl_out['pose'][:, 3:]], dim=1), 'global_shape': smpl_out['pred_shape'], 'global_rotmat': g_rotmat,
            'global_kp_3d': g_joints, 'global_verts': g_verts}


GLOBAL_LOSS_KEYS = ('global_verts', 'global_pose', 'global_kp_3d')


def global_cotangent_loss(g_out, seed=2, dev=None):
    """Fixed random linear functional of ``global_output`` (same role as cotangent_loss): reaches global_orient.*, and through global_pose the
    angle-axis conversion of the stage-3 rotations."""
    g = torch.Generator().manual_seed(seed)
    total = 0.0
    for k in GLOBAL_LOSS_KEYS:
        t = g_out[k]
        c = torch.randn(t.shape, generator=g, dtype=torch.float32) / float(max(1, t[0].numel())) ** 0.5
        total = total + (t * (c.to(dev) if dev is not None else c)).sum()
    return total


def whmr_forward_train(sd, assets, x, center, scale, bbox_height, orig_shape, bbox_info, stage=2, stats=None, dp_out=None,
                       drop_masks=None, drop_path_rate=0.0, relu_gates=None, fmaps_out=None, global_out=None, cam_rotmat=None):
    """-> list of the 4 ``smpl_out`` dicts (mean-pose mesh + 3 stages).  ``stats`` (dict, optional) receives the updated BN running stats,
    ``dp_out`` (list, optional) the IUV head's output dict, ``fmaps_out`` (list, optional) the three deconv maps, ``global_out`` (list, optional)
    the ``global_output`` dict of whmr.py:630-654 with its graph, computed with ``cam_rotmat`` [B, 3, 3] (identity when None).
    ``relu_gates`` (tests only; None = the reference's arithmetic): three boolean NCHW masks that REPLACE the ReLU decisions of the deconv stages
    (whmr.py:488-498) -- a parity test injects the gates another evaluation took, so that pre-activations within rounding of zero stop showing up
    as O(1/sqrt(map size)) differences of every gradient behind them (tests/test_train_gpu.py)."""
    B = x.shape[0]
    s_feat = vit_forward(sd, x, 'feature_extractor.backbone.', drop_masks=drop_masks, drop_path_rate=drop_path_rate)   # stochastic depth (vit.py:132-139)
    smpl_out = OW.regressor_forward_init(sd, assets, B)
    outs, fmaps = [smpl_out], []
    for i in range(3):
        w = sd['deconv_layers.%d.weight' % (3 * i)]
        z = _bn_train(F.conv_transpose2d(s_feat, w, None, stride=2, padding=1), sd, 'deconv_layers.%d.' % (3 * i + 1), stats)
        s_feat = F.relu(z) if relu_gates is None else z * relu_gates[i].to(z.dtype)
        fmaps.append(s_feat)
        if fmaps_out is not None:
            fmaps_out.append(s_feat)
    Tz = tz_head_train(sd, s_feat.detach() if stage == 1 else s_feat, stats)
    if dp_out is not None:
        dp_out.append(dp_head_forward(sd, s_feat))
    for i in range(3):
        cam, shape = smpl_out['pred_cam'].detach(), smpl_out['pred_shape'].detach()
        pose, markers = smpl_out['rotmat'].detach(), smpl_out['markers'].detach()
        pts = sd['points_grid'].expand(B, -1, -1).transpose(1, 2) if i == 0 else G.projection(markers, cam)
        ref, _ = OW.maf_sampling(sd, pts, fmaps[i], 'maf_extractor.%d.' % i)
        smpl_out, body_feat = regressor_forward_train(sd, assets, i, ref, bbox_info, Tz, orig_shape, center, scale, bbox_height, pose, shape, cam, stage)
        outs.append(smpl_out)
    if global_out is not None:
        cr = cam_rotmat if cam_rotmat is not None else torch.eye(3).unsqueeze(0).expand(B, -1, -1)
        global_out.append(global_output_train(sd, assets, smpl_out, body_feat, cr))
    return outs


def dp_head_forward(sd, s_feat):
    """IUV_predict_layer.forward (models/iuv_predictor.py:71-91) on the last feature map (whmr.py:656-658, AUX_SUPV_ON)."""
    c = lambda n: F.conv2d(s_feat, sd['dp_head.%s.weight' % n], sd['dp_head.%s.bias' % n], stride=1, padding=1)
    return {'predict_uv_index': c('predict_uv_index'), 'predict_ann_index': c('predict_ann_index'), 'predict_u': c('predict_u'),
            'predict_v': c('predict_v')}


def dp_cotangent_loss(dp, seed=1, dev=None):
    """Fixed random linear functional of the IUV head's four outputs (same role as cotangent_loss)."""
    g = torch.Generator().manual_seed(seed)
    total = 0.0
    for k in ('predict_u', 'predict_v', 'predict_uv_index', 'predict_ann_index'):
        t = dp[k]
        c = torch.randn(t.shape, generator=g, dtype=torch.float32).to(t.dtype if dev is None else torch.float32) / float(t[0].numel()) ** 0.5
        total = total + (t * (c.to(dev) if dev is not None else c)).sum()
    return total


def cotangent_loss(outs, seed=0, dev=None):
    """A fixed random linear functional of the stage outputs core/trainer.py:500-600 puts losses on (TRAIN_LOSS_KEYS of stages 1-3):
    sum_k <c_k, out_k> with unit-scale cotangents normalised by each tensor's size.  Shared by the reference run, the oracle and the
    HIP test so that all three differentiate the same scalar."""
    g = torch.Generator().manual_seed(seed)
    total = 0.0
    for l in range(1, len(outs)):
        for k in TRAIN_LOSS_KEYS:
            t = outs[l][k]
            c = torch.randn(t.shape, generator=g, dtype=torch.float32).to(t.dtype if dev is None else torch.float32) / float(max(1, t[0].numel())) ** 0.5
            if k in ('kp_2d', 'kp_2d_w'):
                c = c * 0.05
            if k == 'focal_length':
                c = c * 1e-3
            total = total + (t * (c.to(dev) if dev is not None else c)).sum()
    return total
