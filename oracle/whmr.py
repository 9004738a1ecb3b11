"""Oracle restatement of WHMR.forward (eval) on CPU.  TEST INFRASTRUCTURE ONLY.

Functional over ``sd`` (state dict with the reference's key names)
and ``assets`` (synthetic SMPL model + marker ids, oracle/synth.py).  Follows
models/whmr.py:503-678 (orchestration), :42-269 (Regressor), :272-305
(Global_Orient_Regressor), :417-430,567-577 (Tz head), :459-501 (deconv pyramid),
models/maf_extractor.py:75-143 (sampler + point MLP), models/cam_model.py:72-81 and
utils/cam_utils.py:114-145 (camera calibration head).
"""
import math

import torch
import torch.nn.functional as F

from . import geometry as G
from . import smpl as S
from .vit import vit_forward

BN_EPS = 1e-5


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + 'running_mean'], sd[p + 'running_var'], sd[p + 'weight'], sd[p + 'bias'],
                        False, 0.0, BN_EPS)


# ----------------------------------------------------------------------------- deconv pyramid
def deconv_stage(sd, x, i):
    """whmr.py:488-498: ConvTranspose2d(k4,s2,p1,no bias) -> BatchNorm2d (eval) -> ReLU.  Keys deconv_layers.{3i,3i+1}."""
    y = F.conv_transpose2d(x, sd['deconv_layers.%d.weight' % (3 * i)], None, stride=2, padding=1)
    return F.relu(_bn(y, sd, 'deconv_layers.%d.' % (3 * i + 1)))


# ----------------------------------------------------------------------------- Tz head
def timm_block(sd, x, p, num_heads):
    """timm==0.4.9 vision_transformer.Block [3P, restated]: pre-LN eps 1e-5, qkv without bias, GELU MLP x4."""
    B, N, C = x.shape
    hd = C // num_heads
    h = F.layer_norm(x, (C,), sd[p + 'norm1.weight'], sd[p + 'norm1.bias'], 1e-5)
    qkv = F.linear(h, sd[p + 'attn.qkv.weight'], sd.get(p + 'attn.qkv.bias'))
    qkv = qkv.reshape(B, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    a = ((qkv[0] @ qkv[1].transpose(-2, -1)) * hd ** -0.5).softmax(dim=-1)
    h = (a @ qkv[2]).transpose(1, 2).reshape(B, N, C)
    x = x + F.linear(h, sd[p + 'attn.proj.weight'], sd[p + 'attn.proj.bias'])
    h = F.layer_norm(x, (C,), sd[p + 'norm2.weight'], sd[p + 'norm2.bias'], 1e-5)
    h = F.gelu(F.linear(h, sd[p + 'mlp.fc1.weight'], sd[p + 'mlp.fc1.bias']))
    return x + F.linear(h, sd[p + 'mlp.fc2.weight'], sd[p + 'mlp.fc2.bias'])


def tz_head(sd, s_feat):
    """whmr.py:418-430,567-577 (vitpose branch).  s_feat [B,256,128,96] -> Tz [B] in (0,10)."""
    B = s_feat.shape[0]
    y = F.conv2d(s_feat, sd['conv.0.weight'], None, stride=3)
    y = F.conv2d(y, sd['conv.1.weight'], None, stride=2)
    y = y.reshape(B, 5, -1)
    y = timm_block(sd, y, 'transformer_decoder.', 2).transpose(1, 2)       # [B,216,5]
    y = F.avg_pool1d(y, 5).squeeze(-1)                                      # [B,216]
    y = F.linear(y, sd['est_Tz.0.weight'], sd['est_Tz.0.bias'])
    y = F.linear(y, sd['est_Tz.1.weight'], sd['est_Tz.1.bias'])
    y = F.batch_norm(y, sd['est_Tz.2.running_mean'], sd['est_Tz.2.running_var'], sd['est_Tz.2.weight'],
                     sd['est_Tz.2.bias'], False, 0.0, BN_EPS)
    return 10.0 * torch.sigmoid(y).squeeze(-1)


# ----------------------------------------------------------------------------- MAF sampler
def maf_reduce_dim(sd, feat, p):
    """maf_extractor.py:75-101: 1x1-conv MLP 256->128->(+256)->64->(+256)->32, leaky_relu(0.01) x2, ReLU, channel-major flatten."""
    y = F.leaky_relu(F.conv1d(feat, sd[p + 'conv0.weight'], sd[p + 'conv0.bias']))
    y = F.leaky_relu(F.conv1d(torch.cat([y, feat], 1), sd[p + 'conv1.weight'], sd[p + 'conv1.bias']))
    y = F.relu(F.conv1d(torch.cat([y, feat], 1), sd[p + 'conv2.weight'], sd[p + 'conv2.bias']))
    return y.reshape(y.shape[0], -1)


def maf_sampling(sd, points, im_feat, p):
    """maf_extractor.py:103-124: bilinear grid_sample(align_corners=True, zeros) then the point MLP."""
    pf = F.grid_sample(im_feat, points.unsqueeze(2), align_corners=True)[..., 0]
    return maf_reduce_dim(sd, pf, p), pf


# ----------------------------------------------------------------------------- regressor
def _smpl_aux(verts, assets):
    """whmr.py:182-187: Dmap matmuls, markers, SMPL joints + selector."""
    sub = torch.matmul(assets['Dmap0'], verts)
    temp = torch.matmul(assets['Dmap1'], sub)
    markers = verts[:, assets['ssm']]
    j = torch.einsum('bik,ji->bjk', verts, assets['smpl']['J_regressor'])
    return sub, temp, markers, S.vertex_joint_selector(verts, j)


def regressor_forward(sd, assets, i, x, bbox_info, Tz, orig_shape, center, scale, bbox_height,
                      pose, shape, cam, J_regressor=None, stage=2):
    """whmr.py:102-209 with is_train=False, n_iter=1.  ``pose`` is [B,24,3,3] or [B,216]."""
    p = 'regressor.%d.' % i
    B = x.shape[0]
    x = torch.cat((x, bbox_info), dim=1)
    pose = pose.reshape(B, -1)
    xc = torch.cat([x, pose, shape, cam], 1)
    xc = F.linear(xc, sd[p + 'fc1.weight'], sd[p + 'fc1.bias'])
    xc = F.linear(xc, sd[p + 'fc2.weight'], sd[p + 'fc2.bias'])          # no nonlinearity, dropout = identity in eval
    pose = F.linear(xc, sd[p + 'decpose.weight'], sd[p + 'decpose.bias']) + pose
    shape = F.linear(xc, sd[p + 'decshape.weight'], sd[p + 'decshape.bias']) + shape
    cam = F.linear(xc, sd[p + 'deccam.weight'], sd[p + 'deccam.bias']) + cam
    rotmat = G.unbiased_gram_schmidt(pose.view(B, 24, 3, 3))
    verts, joints = S.smpl_forward(shape, rotmat, assets['smpl'])
    kp_2d = G.projection(joints, cam)
    s = cam[:, 0]
    focal = s * bbox_height * Tz / 2.0
    cam_center = orig_shape[:, [1, 0]] / 2.0
    cam_t = G.convert_pare_to_full_img_cam(cam, bbox_height, center, orig_shape[:, 1], orig_shape[:, 0], Tz)
    kp_w = G.perspective_projection(joints, torch.eye(3).unsqueeze(0), cam_t, focal, cam_center)
    kp_w = kp_w / cam_center.unsqueeze(1) - 1
    aa = G.rotation_matrix_to_angle_axis(rotmat.reshape(-1, 3, 3)).reshape(-1, 72)
    kp_3d = joints
    if J_regressor is not None:
        jj = torch.matmul(J_regressor, verts)
        kp_3d = jj[:, S.H36M_TO_J14] - jj[:, [0]]
    sub, temp, markers, smpl_j = _smpl_aux(verts, assets)
    out = {'theta': torch.cat([cam, shape, aa], dim=1), 'verts': verts, 'sub_verts': sub, 'temp_verts': temp,
           'kp_2d': kp_2d, 'kp_2d_w': kp_w, 'kp_3d': kp_3d, 'smpl_kp_3d': smpl_j, 'rotmat': rotmat,
           'pred_cam': cam, 'pred_cam_t': cam_t, 'pred_shape': shape, 'pred_pose': pose, 'pose': aa,
           'pelvis': smpl_j[:, :1], 'scale': scale, 'focal_length': focal, 'markers': markers}
    return out, x


def regressor_forward_init(sd, assets, B, J_regressor=None):
    """whmr.py:211-269 (mean-pose mesh; constant per model)."""
    p = 'regressor.0.'
    pose = sd[p + 'init_pose'].expand(B, -1)
    shape = sd[p + 'init_shape'].expand(B, -1)
    cam = sd[p + 'init_cam'].expand(B, -1)
    rotmat = pose.reshape(B, 24, 3, 3)
    verts, joints = S.smpl_forward(shape, rotmat, assets['smpl'])
    kp_2d = G.projection(joints, cam)
    aa = G.rotation_matrix_to_angle_axis(rotmat.reshape(-1, 3, 3)).reshape(-1, 72)
    kp_3d = joints
    if J_regressor is not None:
        jj = torch.matmul(J_regressor, verts)
        kp_3d = jj[:, S.H36M_TO_J14] - jj[:, [0]]
    sub, temp, markers, smpl_j = _smpl_aux(verts, assets)
    return {'theta': torch.cat([cam, shape, aa], dim=1), 'verts': verts, 'sub_verts': sub, 'temp_verts': temp,
            'kp_2d': kp_2d, 'kp_3d': kp_3d, 'smpl_kp_3d': smpl_j, 'rotmat': rotmat, 'pred_cam': cam,
            'pred_shape': shape, 'pred_pose': pose, 'pose': aa, 'pelvis': smpl_j[:, :1], 'markers': markers}


def global_orient_forward(sd, x, cam_rotmat, local_orient):
    """whmr.py:289-305, eval: the three 'iterations' are identical (local_orient never updated)."""
    B = x.shape[0]
    lo = local_orient.reshape(B, -1)
    xc = torch.cat([x, G.rotmat_to_rot6d(cam_rotmat), lo], dim=1)
    xc = F.linear(xc, sd['global_orient.fc1.weight'], sd['global_orient.fc1.bias'])
    xc = F.linear(xc, sd['global_orient.fc2.weight'], sd['global_orient.fc2.bias'])
    r = F.linear(xc, sd['global_orient.decrot.weight'], sd['global_orient.decrot.bias']) + lo
    return G.unbiased_gram_schmidt(r.reshape(-1, 1, 3, 3))


# ----------------------------------------------------------------------------- camera calibration head
def _bottleneck(sd, x, p, stride):
    y = F.relu(_bn(F.conv2d(x, sd[p + 'conv1.weight']), sd, p + 'bn1.'))
    y = F.relu(_bn(F.conv2d(y, sd[p + 'conv2.weight'], stride=stride, padding=1), sd, p + 'bn2.'))
    y = _bn(F.conv2d(y, sd[p + 'conv3.weight']), sd, p + 'bn3.')
    if (p + 'downsample.0.weight') in sd:
        x = _bn(F.conv2d(x, sd[p + 'downsample.0.weight'], stride=stride), sd, p + 'downsample.1.')
    return F.relu(y + x)


def resnet50_features(sd, x, p):
    """pare.models.backbone.resnet50 [3P]: torchvision-style ResNet-50 returning the layer4 map."""
    x = F.relu(_bn(F.conv2d(x, sd[p + 'conv1.weight'], stride=2, padding=3), sd, p + 'bn1.'))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, (n, stride) in enumerate(zip([3, 4, 6, 3], [1, 2, 2, 2])):
        for bi in range(n):
            x = _bottleneck(sd, x, p + 'layer%d.%d.' % (li + 1, bi), stride if bi == 0 else 1)
    return x


def softargmax1d(logits):
    """pare.models.layers.softargmax.softargmax1d(normalize_keypoints=True) [3P, restated]:
    softmax over the D bins, expectation of the bin index, mapped k/(D-1)*2-1."""
    D = logits.shape[-1]
    w = F.softmax(logits, dim=-1)
    idx = (w * torch.arange(D, dtype=logits.dtype)).sum(-1)
    return idx / (D - 1) * 2 - 1


VFOV_RANGE = (0.2617, 2.1)       # utils/cam_utils.py:56
PITCH_RANGE = (-0.6, 0.6)        # utils/cam_utils.py:38
ROLL_RANGE = (-0.6, 0.6)         # utils/cam_utils.py:139


def cam_model_forward(sd, full_x, taps=None):
    """cam_model.py:72-81 + cam_utils.py:121-145 (softargmax_l2) + whmr.py:515-522 -> (cam_rotmat, render_rotmat).
    ``taps`` (dict) receives the pooled features [B,2048], the three [B,256] logit vectors and the angles."""
    f = resnet50_features(sd, full_x, 'cam_model.backbone.')
    f = f.mean(dim=(2, 3))
    ang = []
    if taps is not None:
        taps['feat'] = f
    for name, (lo, hi) in (('vfov', VFOV_RANGE), ('pitch', PITCH_RANGE), ('roll', ROLL_RANGE)):
        logits = F.linear(f, sd['cam_model.fc_%s.weight' % name], sd['cam_model.fc_%s.bias' % name])
        sidx = softargmax1d(logits)
        ang.append((hi - lo) * ((sidx + 1) / 2) + lo)
        if taps is not None:
            taps['logits_' + name], taps['angle_' + name] = logits, ang[-1]
    pitch, roll = ang[1].unsqueeze(-1), ang[2].unsqueeze(-1)
    z = torch.zeros_like(pitch)
    return (G.batch_euler2matrix(torch.cat([pitch, z, roll], 1).float()),
            G.batch_euler2matrix(torch.cat([-pitch, z, roll], 1).float()))


# ----------------------------------------------------------------------------- full forward
def whmr_forward(sd, assets, x, center, scale, bbox_height, orig_shape, bbox_info,
                 J_regressor=None, full_x=None, cam_rotmat=None, view='vis', taps=None):
    """WHMR.forward (whmr.py:503-678), is_train=False.  view in {'vis','train','eval'}.

    Unlike the released reference, full_x=None / cam_rotmat=... works: render_rotmat
    falls back to cam_rotmat (the reference raises NameError there).
    """
    B = x.shape[0]
    render_rotmat = None
    if cam_rotmat is None:
        if full_x is not None:
            cam_rotmat, render_rotmat = cam_model_forward(sd, full_x)
        else:
            cam_rotmat = torch.eye(3).unsqueeze(0).expand(B, -1, -1)
    if render_rotmat is None:
        render_rotmat = cam_rotmat

    s_feat = vit_forward(sd, x, 'feature_extractor.backbone.')
    smpl_out = regressor_forward_init(sd, assets, B, J_regressor)
    outs = [smpl_out]
    fmaps = [s_feat]
    for i in range(3):
        s_feat = deconv_stage(sd, s_feat, i)
        fmaps.append(s_feat)
    Tz = tz_head(sd, s_feat)
    if taps is not None:
        taps.update(s_feat0=fmaps[0], fmaps=fmaps[1:], Tz=Tz, ref_feature=[])

    body_feat = None
    for i in range(3):
        cam, shape, pose, markers = smpl_out['pred_cam'], smpl_out['pred_shape'], smpl_out['rotmat'], smpl_out['markers']
        if i == 0:
            pts = sd['points_grid'].expand(B, -1, -1).transpose(1, 2)
        else:
            pts = G.projection(markers, cam)
        ref, _ = maf_sampling(sd, pts, fmaps[i + 1], 'maf_extractor.%d.' % i)
        if taps is not None:
            taps['ref_feature'].append(ref)
        smpl_out, body_feat = regressor_forward(sd, assets, i, ref, bbox_info, Tz, orig_shape, center, scale,
                                                bbox_height, pose, shape, cam, J_regressor)
        outs.append(smpl_out)

    g_rot = global_orient_forward(sd, body_feat, cam_rotmat, smpl_out['rotmat'][:, 0])
    g_aa = G.rotation_matrix_to_angle_axis(g_rot.reshape(-1, 3, 3)).reshape(-1, 3)
    g_pose = torch.cat([g_aa, smpl_out['pose'][:, 3:]], dim=1)
    g_rotmat = torch.cat([g_rot, smpl_out['rotmat'][:, 1:]], dim=1)
    g_verts, g_joints = S.smpl_forward(smpl_out['pred_shape'], g_rotmat, assets['smpl'])
    if J_regressor is not None:
        jj = torch.matmul(J_regressor, g_verts)
        g_joints = jj[:, S.H36M_TO_J14] - jj[:, [0]]
    g_out = {'global_pose': g_pose, 'global_shape': smpl_out['pred_shape'], 'global_rotmat': g_rotmat,
             'global_kp_3d': g_joints, 'global_verts': g_verts}
    if view == 'eval':
        return {'global_output': g_out}, None
    if view == 'train':
        return {'smpl_out': outs, 'dp_out': [], 'dpth_out': [], 'global_output': g_out}, fmaps
    return {'local_smpl_vertices': smpl_out['verts'], 'smpl_vertices': g_verts, 'pred_cam_t': smpl_out['pred_cam_t'],
            'focal_length': smpl_out['focal_length'], 'cam_rotmat': cam_rotmat, 'render_rotmat': render_rotmat,
            'shape': smpl_out['pred_shape'], 'global_pose': g_pose, 'local_pose': smpl_out['pose']}
