"""Deterministic synthetic weights / SMPL model / inputs for benchmarks, smoke runs and tests (a data generator: no compute path).

The reference ships neither weights nor the licensed SMPL model, so
parity is established on synthetic tensors of the true shapes, like the
vendored mmpose tests do (models/ViTPose/tests/utils/mesh_utils.py:9-38) but with random
non-zero values.  Everything is drawn from numpy's PCG64 ``default_rng`` keyed by
crc32(tensor name) ^ seed, so values do not depend on torch's RNG, on the order of
generation, or on the machine -- the GPU box regenerates bit-identical tensors.
"""
import math
import zlib

import numpy as np
import torch

NV = 6890


def _rng(seed, key):
    return np.random.default_rng([zlib.crc32(key.encode()), seed])


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def _rot6d_to_rotmat_cpu(x):
    """utils/geometry.py:243-257 on the host (F.normalize eps 1e-12, cross product), for the init_pose buffers of the state dict."""
    x = x.reshape(-1, 3, 2)
    a1, a2 = x[:, :, 0], x[:, :, 1]
    b1 = a1 / a1.norm(dim=1, keepdim=True).clamp_min(1e-12)
    b2 = a2 - (b1 * a2).sum(-1, keepdim=True) * b1
    b2 = b2 / b2.norm(dim=1, keepdim=True).clamp_min(1e-12)
    return torch.stack((b1, b2, torch.linalg.cross(b1, b2, dim=-1)), dim=-1)


# ----------------------------------------------------------------------------- SMPL + mesh assets
def make_assets(seed=0):
    """Synthetic SMPL body + marker ids + down-sampling maps + mean params."""
    r = _rng(seed, 'smpl')
    centers = r.normal(0, 0.3, (24, 3))
    home = np.arange(NV) % 24
    v_template = centers[home] + r.normal(0, 0.06, (NV, 3))
    shapedirs = r.normal(0, 0.01, (NV, 3, 10))
    posedirs = r.normal(0, 0.003, (NV, 3, 207))
    Jreg = np.zeros((24, NV))
    for j in range(24):
        idx = r.choice(np.nonzero(home == j)[0], 40, replace=False)
        w = r.uniform(0.1, 1.0, 40)
        Jreg[j, idx] = w / w.sum()
    W = r.uniform(0, 1, (NV, 24)) ** 8
    W[np.arange(NV), home] += 1.0
    W /= W.sum(1, keepdims=True)
    Jextra = np.zeros((9, NV))
    for j in range(9):
        idx = r.choice(NV, 30, replace=False)
        w = r.uniform(0.1, 1.0, 30)
        Jextra[j, idx] = w / w.sum()
    ssm = np.sort(r.choice(NV, 67, replace=False))
    D0 = np.zeros((1723, NV))
    for i in range(1723):
        D0[i, r.choice(NV, 3, replace=False)] = 1.0 / 3
    D1 = np.zeros((431, 1723))
    for i in range(431):
        D1[i, r.choice(1723, 3, replace=False)] = 1.0 / 3
    pose6 = np.tile(np.array([1, 0, 0, 1, 0, 0], dtype=np.float64), 24) + r.normal(0, 0.2, 144)
    mean = {'pose': pose6.astype(np.float32), 'shape': r.normal(0, 0.5, 10).astype(np.float32),
            'cam': np.array([0.9, 0.0, 0.0], dtype=np.float32)}
    smpl = {'v_template': _t(v_template), 'shapedirs': _t(shapedirs),
            'posedirs': _t(posedirs.reshape(NV * 3, 207).T),           # smplx layout [207, 20670]
            'J_regressor': _t(Jreg), 'lbs_weights': _t(W), 'J_regressor_extra': _t(Jextra)}
    return {'smpl': smpl, 'ssm': torch.from_numpy(ssm.astype(np.int64)), 'Dmap0': _t(D0), 'Dmap1': _t(D1),
            'mean_params': mean, 'faces': torch.zeros(1, 3, dtype=torch.int64)}


# ----------------------------------------------------------------------------- weights
def _normal(seed, key, shape, std, mean=0.0):
    return _t(_rng(seed, key).normal(mean, std, shape))


def _uniform(seed, key, shape, lo, hi):
    return _t(_rng(seed, key).uniform(lo, hi, shape))


def _linear(sd, seed, key, out_f, in_f, std=None, bias=True, xavier_gain=None):
    if xavier_gain is not None:
        a = xavier_gain * math.sqrt(6.0 / (in_f + out_f))
        sd[key + '.weight'] = _uniform(seed, key + '.weight', (out_f, in_f), -a, a)
    else:
        std = std if std is not None else 1.0 / math.sqrt(3.0 * in_f)
        sd[key + '.weight'] = _normal(seed, key + '.weight', (out_f, in_f), std)
    if bias:
        sd[key + '.bias'] = _normal(seed, key + '.bias', (out_f,), 0.02)


def _ln(sd, seed, key, n):
    sd[key + '.weight'] = _normal(seed, key + '.weight', (n,), 0.1, 1.0)
    sd[key + '.bias'] = _normal(seed, key + '.bias', (n,), 0.1)


def _bn(sd, seed, key, n, g=(0.5, 1.5)):
    sd[key + '.weight'] = _uniform(seed, key + '.weight', (n,), *g)
    sd[key + '.bias'] = _normal(seed, key + '.bias', (n,), 0.1)
    sd[key + '.running_mean'] = _normal(seed, key + '.running_mean', (n,), 0.1)
    sd[key + '.running_var'] = _uniform(seed, key + '.running_var', (n,), 0.5, 1.5)
    sd[key + '.num_batches_tracked'] = torch.zeros((), dtype=torch.int64)


def make_vit_state(seed=0, img_size=(256, 192), embed_dim=768, depth=12, prefix='', patch=16, mlp_ratio=4):
    """ViTPose backbone weights.  Key names: vit.py:143-341 / SURVEY App. B.

    qkv std is chosen so attention logits have O(1) spread (the softmax is actually exercised).
    """
    sd = {}
    p = prefix
    hp = (img_size[0] + 4 - patch) // patch + 1
    wp = (img_size[1] + 4 - patch) // patch + 1
    # pos_embed is sized from img_size // patch (vit.py:150,231), not from the padded conv output.
    n = (img_size[0] // patch) * (img_size[1] // patch)
    assert n == hp * wp
    sd[p + 'pos_embed'] = _normal(seed, p + 'pos_embed', (1, n + 1, embed_dim), 0.02)
    sd[p + 'patch_embed.proj.weight'] = _normal(seed, p + 'patch_embed.proj.weight', (embed_dim, 3, patch, patch),
                                                1.0 / math.sqrt(3 * patch * patch))
    sd[p + 'patch_embed.proj.bias'] = _normal(seed, p + 'patch_embed.proj.bias', (embed_dim,), 0.02)
    hid = int(embed_dim * mlp_ratio)
    for i in range(depth):
        b = p + 'blocks.%d.' % i
        _ln(sd, seed, b + 'norm1', embed_dim)
        _linear(sd, seed, b + 'attn.qkv', 3 * embed_dim, embed_dim, std=1.25 / math.sqrt(embed_dim))
        _linear(sd, seed, b + 'attn.proj', embed_dim, embed_dim, std=0.02)
        _ln(sd, seed, b + 'norm2', embed_dim)
        _linear(sd, seed, b + 'mlp.fc1', hid, embed_dim, std=0.03)
        _linear(sd, seed, b + 'mlp.fc2', embed_dim, hid, std=0.02)
    _ln(sd, seed, p + 'last_norm', embed_dim)
    return sd


def _resnet50(sd, seed, p):
    def conv(key, o, i, k, gain=2.0):
        sd[key + '.weight'] = _normal(seed, key + '.weight', (o, i, k, k), math.sqrt(gain / (i * k * k)))
    conv(p + 'conv1', 64, 3, 7)
    _bn(sd, seed, p + 'bn1', 64)
    inpl = 64
    for li, (n, planes) in enumerate(zip([3, 4, 6, 3], [64, 128, 256, 512])):
        for bi in range(n):
            b = p + 'layer%d.%d.' % (li + 1, bi)
            conv(b + 'conv1', planes, inpl, 1)
            _bn(sd, seed, b + 'bn1', planes, (0.5, 1.0))
            conv(b + 'conv2', planes, planes, 3)
            _bn(sd, seed, b + 'bn2', planes, (0.5, 1.0))
            conv(b + 'conv3', planes * 4, planes, 1)
            _bn(sd, seed, b + 'bn3', planes * 4, (0.1, 0.3))
            if bi == 0:
                conv(b + 'downsample.0', planes * 4, inpl, 1, gain=1.0)
                _bn(sd, seed, b + 'downsample.1', planes * 4, (0.5, 1.0))
            inpl = planes * 4


def make_state_dict(seed=0, assets=None, img_size=(256, 192), with_cam_model=True):
    """Full WHMR state dict (own-code keys of SURVEY App. B; 3P sub-module keys use
    timm / torchvision naming).  SMPL buffers live in ``assets`` for the oracle."""
    assets = assets if assets is not None else make_assets(seed)
    sd = make_vit_state(seed, img_size, prefix='feature_extractor.backbone.')
    xv, yv = torch.meshgrid([torch.linspace(-1, 1, 7), torch.linspace(-1, 1, 9)], indexing='ij')
    sd['points_grid'] = torch.stack([xv.reshape(-1), yv.reshape(-1)]).unsqueeze(0)       # whmr.py:345-347
    cin = 768
    for i in range(3):
        k = 'deconv_layers.%d.weight' % (3 * i)
        sd[k] = _normal(seed, k, (cin, 256, 4, 4), 1.0 / math.sqrt(2.0 * cin))
        _bn(sd, seed, 'deconv_layers.%d' % (3 * i + 1), 256)
        cin = 256
    dmap = torch.matmul(assets['Dmap1'], assets['Dmap0'])
    for i in range(3):
        m = 'maf_extractor.%d.' % i
        sd[m + 'Dmap'] = dmap
        for l, (o, c) in enumerate([(128, 256), (64, 384), (32, 320)]):
            sd[m + 'conv%d.weight' % l] = _normal(seed, m + 'conv%d.weight' % l, (o, c, 1), 1.0 / math.sqrt(c))
            sd[m + 'conv%d.bias' % l] = _normal(seed, m + 'conv%d.bias' % l, (o,), 0.05)
    mp = assets['mean_params']
    init_pose = _rot6d_to_rotmat_cpu(torch.from_numpy(mp['pose']).reshape(1, 24, 6)).reshape(1, -1)   # whmr.py:64-65
    for i in range(3):
        r = 'regressor.%d.' % i
        feat = 63 * 32 if i == 0 else 67 * 32
        sd[r + 'init_pose'] = init_pose
        sd[r + 'init_shape'] = torch.from_numpy(mp['shape']).unsqueeze(0)
        sd[r + 'init_cam'] = torch.from_numpy(mp['cam']).unsqueeze(0)
        sd[r + 'Dmap0'] = assets['Dmap0']
        sd[r + 'Dmap1'] = assets['Dmap1']
        _linear(sd, seed, r + 'fc1', 1024, feat + 216 + 13 + 5)
        _linear(sd, seed, r + 'fc2', 1024, 1024)
        _linear(sd, seed, r + 'decpose', 216, 1024, xavier_gain=0.01)
        _linear(sd, seed, r + 'decshape', 10, 1024, xavier_gain=0.01)
        _linear(sd, seed, r + 'deccam', 3, 1024, xavier_gain=0.01)
    for k, o in (('predict_u', 25), ('predict_v', 25), ('predict_uv_index', 25), ('predict_ann_index', 15)):
        sd['dp_head.%s.weight' % k] = _normal(seed, 'dp_head.%s.weight' % k, (o, 256, 3, 3), 0.02)
        sd['dp_head.%s.bias' % k] = _normal(seed, 'dp_head.%s.bias' % k, (o,), 0.02)
    sd['conv.0.weight'] = _normal(seed, 'conv.0.weight', (64, 256, 7, 7), math.sqrt(2.0 / (256 * 49)))
    sd['conv.1.weight'] = _normal(seed, 'conv.1.weight', (5, 64, 7, 7), math.sqrt(1.0 / (64 * 49)))
    t = 'transformer_decoder.'
    _ln(sd, seed, t + 'norm1', 216)
    _linear(sd, seed, t + 'attn.qkv', 648, 216, std=0.1, bias=False)
    _linear(sd, seed, t + 'attn.proj', 216, 216)
    _ln(sd, seed, t + 'norm2', 216)
    _linear(sd, seed, t + 'mlp.fc1', 864, 216)
    _linear(sd, seed, t + 'mlp.fc2', 216, 864)
    _linear(sd, seed, 'est_Tz.0', 12, 216)
    _linear(sd, seed, 'est_Tz.1', 1, 12)
    _bn(sd, seed, 'est_Tz.2', 1)
    if with_cam_model:
        _resnet50(sd, seed, 'cam_model.backbone.')
        for n in ('vfov', 'pitch', 'roll'):
            _linear(sd, seed, 'cam_model.fc_%s' % n, 256, 2048, std=0.01)
        # peaked bin biases: with the plain init all three soft-argmaxes sit mid-range (pitch = roll = 0 to 1e-3), which would leave the
        # euler -> rotation convention (Rx(pitch) . Rz(roll), whmr.py:521-522) invisible to the end-to-end fixture.  Bumps at bins 202 / 64
        # put pitch near +0.35 rad and roll near -0.30 rad.
        k = torch.arange(256, dtype=torch.float32)
        for n, k0 in (('pitch', 202.0), ('roll', 64.0)):
            sd['cam_model.fc_%s.bias' % n] = sd['cam_model.fc_%s.bias' % n] + 8.0 * torch.exp(-((k - k0) / 6.0) ** 2)
    sd['global_orient.init_pose'] = init_pose.reshape(1, 24, 9)[:, 0]
    _linear(sd, seed, 'global_orient.fc1', 2048, 2149 + 6 + 9)
    _linear(sd, seed, 'global_orient.fc2', 2048, 2048)
    _linear(sd, seed, 'global_orient.decrot', 9, 2048, xavier_gain=0.01)
    return sd


def make_densepose_tables(seed=0, assets=None):
    """Synthetic stand-in for the DensePose tables of data/UV_data/UV_Processed.mat (licensed, not shipped): 7829 vertices = the 6890 SMPL
    ones + 939 seam duplicates, 13774 small faces from two spatial orderings of the synthetic rest pose, per-vertex (part / 24, U, V) textures
    with the part index constant per face neighbourhood."""
    assets = assets if assets is not None else make_assets(seed)
    vt = assets['smpl']['v_template']
    r = _rng(seed, 'densepose')
    vmap = torch.cat([torch.arange(6890), _t(r.integers(0, 6890, 939)).long()])
    o1 = torch.argsort((vt[:, 1] * 20).floor() * 1000 + vt[:, 0] * 10)
    o2 = torch.argsort((vt[:, 1] * 20).floor() * 1000 + vt[:, 2] * 10)
    faces = torch.cat([torch.stack([o1[:-2], o1[1:-1], o1[2:]], 1), torch.stack([o2[:-2], o2[2:], o2[1:-1]], 1)])[:13774].to(torch.int32).contiguous()
    part = ((vt[vmap, 1] - vt[:, 1].min()) / (vt[:, 1].max() - vt[:, 1].min() + 1e-6) * 23.99).floor() + 1        # 24 horizontal bands
    tex = torch.stack([part / 24.0, _t(r.uniform(0, 1, 7829)), _t(r.uniform(0, 1, 7829))], 1).float()
    return dict(vert_mapping=vmap, faces=faces, textures_vts=tex)


def make_hmr_state(seed=0, assets=None):
    """HMR (models/hmr.py:164-213) weights: torchvision-style R50 trunk + fc1/fc2/decpose/decshape/deccam + init buffers."""
    assets = assets if assets is not None else make_assets(seed)
    sd = {}
    _resnet50(sd, seed + 100, '')
    _linear(sd, seed, 'fc1', 1024, 2048 + 144 + 13)
    _linear(sd, seed, 'fc2', 1024, 1024)
    _linear(sd, seed, 'decpose', 144, 1024, xavier_gain=0.01)
    _linear(sd, seed, 'decshape', 10, 1024, xavier_gain=0.01)
    _linear(sd, seed, 'deccam', 3, 1024, xavier_gain=0.01)
    mp = assets['mean_params']
    sd['init_pose'] = torch.from_numpy(mp['pose']).unsqueeze(0)
    sd['init_shape'] = torch.from_numpy(mp['shape']).unsqueeze(0)
    sd['init_cam'] = torch.from_numpy(mp['cam']).unsqueeze(0)
    return sd


# ----------------------------------------------------------------------------- inputs
def make_inputs(B, seed=0, img_size=(256, 192), full_size=None):
    """Synthetic crops + bbox metadata (bbox_info per demo/tester.py:136-145)."""
    r = _rng(seed, 'inputs%d' % B)
    H, W = 720.0, 1280.0
    x = _t(r.normal(0, 1, (B, 3) + tuple(img_size)))
    center = _t(np.stack([r.uniform(0.2 * W, 0.8 * W, B), r.uniform(0.2 * H, 0.8 * H, B)], 1))
    scale = _t(r.uniform(0.8, 2.5, B))
    orig_shape = _t(np.tile(np.array([[H, W]]), (B, 1)))
    focal = math.sqrt(H * H + W * W)
    bbox_info = torch.stack([center[:, 0] - W / 2, center[:, 1] - H / 2, 200 * scale,
                             torch.full((B,), W), torch.full((B,), H)], 1) / focal
    out = {'x': x, 'center': center, 'scale': scale, 'bbox_height': 200 * scale, 'orig_shape': orig_shape,
           'bbox_info': bbox_info.float()}
    if full_size is not None:
        out['full_x'] = _t(r.normal(0, 1, (B, 3) + tuple(full_size)))
    return out
