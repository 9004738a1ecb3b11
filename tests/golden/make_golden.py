"""Generate golden fixtures by importing the REAL reference (yw0208/W-HMR) -- build container only.

Usage (from the repo root, in the container that has /root/reference):
    python tests/golden/make_golden.py

What it does (SURVEY Appendix A recipe):
  1. installs small stub modules for the third-party packages the reference imports but
     that are absent here (yacs, timm, mmcv, smplx, pare, torchvision, ...); third-party
     ARITHMETIC (SMPL LBS, softargmax1d, batch_euler2matrix, timm Block, PARE resnet50) is
     supplied by this repo's own restatements -- those pieces stay "parity unpinned";
  2. imports the reference's own files UNMODIFIED from /root/reference (models/whmr.py,
     models/maf_extractor.py, utils/geometry.py, .../backbones/vit.py) and builds WHMR;
  3. loads the deterministic synthetic weights of oracle/synth.py into it, runs
     WHMR.forward on seeded inputs (B=2) and the geometry helpers on edge-case vectors;
  4. asserts the oracle (oracle/whmr.py) reproduces the reference, then writes
     tests/golden/whmr_b2.npz and tests/golden/geometry.npz (data only: inputs + outputs).
Nothing from /root/reference is copied; the fixtures are plain arrays.
"""
import importlib.util
import os
import pickle
import sys
import tempfile
import types

import numpy as np
import scipy.sparse
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)

from oracle import geometry as OG            # noqa: E402
from oracle import smpl as OS                # noqa: E402
from oracle import synth, whmr as OW         # noqa: E402
from oracle.vit import vit_forward           # noqa: E402


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


# ----------------------------------------------------------------------------- stubs
class CfgNode(dict):
    def __init__(self, init=None, new_allowed=False):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def _merge(self, d):
        import ast
        for k, v in d.items():
            if isinstance(v, dict):
                if k not in self or not isinstance(self[k], CfgNode):
                    self[k] = CfgNode()
                self[k]._merge(v)
            else:
                if isinstance(v, str):
                    try:
                        v = ast.literal_eval(v)
                    except Exception:
                        pass
                self[k] = v

    def merge_from_file(self, path):
        import yaml
        with open(path) as f:
            self._merge(yaml.safe_load(f))

    def merge_from_list(self, lst):
        for k, v in zip(lst[0::2], lst[1::2]):
            node = self
            parts = k.split('.')
            for p in parts[:-1]:
                node = node[p]
            node[parts[-1]] = v


class TimmBlock(nn.Module):
    """timm==0.4.9 Block stand-in (pre-LN eps 1e-5, qkv no bias, MLP x4 GELU)."""

    def __init__(self, dim, num_heads, **kw):
        super().__init__()
        self.num_heads = num_heads
        self.norm1 = nn.LayerNorm(dim)
        self.attn = nn.Module()
        self.attn.qkv = nn.Linear(dim, 3 * dim, bias=False)
        self.attn.proj = nn.Linear(dim, dim)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = nn.Module()
        self.mlp.fc1 = nn.Linear(dim, 4 * dim)
        self.mlp.fc2 = nn.Linear(4 * dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        hd = C // self.num_heads
        qkv = self.attn.qkv(self.norm1(x)).reshape(B, N, 3, self.num_heads, hd).permute(2, 0, 3, 1, 4)
        a = ((qkv[0] @ qkv[1].transpose(-2, -1)) * hd ** -0.5).softmax(-1)
        x = x + self.attn.proj((a @ qkv[2]).transpose(1, 2).reshape(B, N, C))
        return x + self.mlp.fc2(F.gelu(self.mlp.fc1(self.norm2(x))))


class Bottleneck(nn.Module):
    def __init__(self, inpl, planes, stride, down):
        super().__init__()
        self.conv1 = nn.Conv2d(inpl, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = None
        if down:
            self.downsample = nn.Sequential(nn.Conv2d(inpl, planes * 4, 1, stride, bias=False),
                                            nn.BatchNorm2d(planes * 4))

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        if self.downsample is not None:
            x = self.downsample(x)
        return F.relu(y + x)


class ResNet50(nn.Module):
    """PARE resnet50 stand-in: torchvision layout, returns the layer4 feature map."""

    def __init__(self, pretrained=False):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        inpl = 64
        for li, (n, planes) in enumerate(zip([3, 4, 6, 3], [64, 128, 256, 512])):
            blocks = []
            for bi in range(n):
                blocks.append(Bottleneck(inpl, planes, (1 if li == 0 else 2) if bi == 0 else 1, bi == 0))
                inpl = planes * 4
            setattr(self, 'layer%d' % (li + 1), nn.Sequential(*blocks))

    def forward(self, x):
        x = F.max_pool2d(F.relu(self.bn1(self.conv1(x))), 3, 2, 1)
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))


ASSETS = synth.make_assets(0)


class SMPLStub(nn.Module):
    """pare.models.SMPL stand-in: forwards to this repo's restatement (oracle/smpl.py)."""

    def __init__(self, *a, **k):
        super().__init__()
        self.faces = np.zeros((1, 3), dtype=np.int64)

    def forward(self, betas=None, body_pose=None, global_orient=None, pose2rot=False, **kw):
        assert not pose2rot
        rot = torch.cat([global_orient, body_pose], dim=1)
        v, j = OS.smpl_forward(betas, rot, ASSETS['smpl'])
        return types.SimpleNamespace(vertices=v, joints=j)


class VJS(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()

    def forward(self, vertices, joints):
        return OS.vertex_joint_selector(vertices, joints)


DROP_QUEUE = []      # make_golden_train.py: per-call [B] keep masks for the stub of timm's drop_path (empty -> identity, as in eval mode)


def drop_path_stub(x, drop_prob=0., training=False):
    """timm.models.layers.drop_path [3P timm 0.4.9, restated]: mask = floor(keep_prob + rand(B, 1, ..)); x / keep_prob * mask.  The random
    draw is replaced by a queue of pre-drawn masks so that the oracle can be run on exactly the same masks."""
    if drop_prob == 0. or not training or not DROP_QUEUE:
        return x
    mask = DROP_QUEUE.pop(0).to(x.dtype).view((x.shape[0],) + (1,) * (x.ndim - 1))
    return x.div(1 - drop_prob) * mask


def install_stubs():
    _mod('yacs')
    _mod('yacs.config', CfgNode=CfgNode)
    _mod('timm')
    _mod('timm.models')
    _mod('timm.models.layers', drop_path=drop_path_stub,
         to_2tuple=lambda v: v if isinstance(v, tuple) else (v, v), trunc_normal_=nn.init.trunc_normal_)
    _mod('timm.models.vision_transformer', Block=TimmBlock)
    _mod('torchvision')
    _mod('torchvision.models')
    tr = _mod('torchvision.models.resnet')
    tr.BasicBlock = tr.Bottleneck = object
    _mod('mmcv', Config=types.SimpleNamespace(fromfile=lambda path: types.SimpleNamespace(backbone=dict(
        img_size=(256, 192), patch_size=16, embed_dim=768, depth=12, num_heads=12, ratio=1, use_checkpoint=False,
        mlp_ratio=4, qkv_bias=True, drop_path_rate=0.3))))
    _mod('smplx')
    _mod('smplx.lbs', vertices2joints=lambda J, v: torch.einsum('bik,ji->bjk', v, J), batch_rodrigues=OG.batch_rodrigues)

    class Struct(object):
        def __init__(self, **kw):
            self.__dict__.update(kw)

    def to_np(a, dtype=np.float32):
        return np.array(a.todense() if scipy.sparse.issparse(a) else a, dtype=dtype)
    _mod('smplx.utils', Struct=Struct, to_tensor=lambda a, dtype=torch.float32: torch.tensor(a, dtype=dtype), to_np=to_np)
    _mod('smplx.vertex_ids', vertex_ids={'smplh': {}})
    _mod('smplx.vertex_joint_selector', VertexJointSelector=VJS)
    _mod('pare')
    _mod('pare.models', SMPL=SMPLStub)
    _mod('pare.models.head', HMRHead=None, SMPLHead=None, SMPLCamHead=None)
    _mod('pare.core', config=types.SimpleNamespace(SMPL_MODEL_DIR='data/smpl'))
    _mod('pare.core.config', SMPL_MODEL_DIR='data/smpl')
    _mod('pare.core.constants', JOINT_MAP={}, JOINT_NAMES=[])
    _mod('pare.utils')
    _mod('pare.utils.geometry', batch_euler2matrix=OG.batch_euler2matrix)
    _mod('pare.utils.train_utils', load_pretrained_model=lambda model, *a, **k: model)
    _mod('pare.models.layers')
    _mod('pare.models.layers.softargmax',
         softargmax1d=lambda h, normalize_keypoints=True: (OW.softargmax1d(h).unsqueeze(-1), None))
    bb = _mod('pare.models.backbone', resnet50=ResNet50)
    bb.__all__ = ['resnet50']
    _mod('pare.models.backbone.utils', get_backbone_info=lambda n: {'n_output_channels': 2048})
    # reference package 'models' without running models/__init__.py (torchvision import)
    m = _mod('models')
    m.__path__ = [os.path.join(REF, 'models')]
    _mod('models.bert')
    _mod('models.bert.modeling_graphormer', Graphormer=None)
    _mod('models.bert.transformers')
    _mod('models.bert.transformers.pytorch_transformers', BertConfig=None)
    _mod('models.e2e_body_network', Graphormer_Body_Network=None)
    # the vendored ViT, loaded by path under a fake package
    _mod('refbb')
    _mod('refbb.builder', BACKBONES=types.SimpleNamespace(register_module=lambda: (lambda c: c)))
    _mod('refbb.backbones')

    class BaseBackbone(nn.Module):
        pass
    _mod('refbb.backbones.base_backbone', BaseBackbone=BaseBackbone)
    spec = importlib.util.spec_from_file_location(
        'refbb.backbones.vit', os.path.join(REF, 'models/ViTPose/mmpose/models/backbones/vit.py'))
    vit = importlib.util.module_from_spec(spec)
    sys.modules['refbb.backbones.vit'] = vit
    spec.loader.exec_module(vit)
    _mod('models.ViTPose')
    _mod('models.ViTPose.mmpose')
    _mod('models.ViTPose.mmpose.models', build_backbone=lambda c: vit.ViT(**c))
    return vit


def patch_torch_cuda():
    for name in ('eye', 'zeros', 'ones', 'tensor'):
        orig = getattr(torch, name)

        def wrap(*a, _o=orig, **k):
            if str(k.get('device', '')).startswith('cuda'):
                k['device'] = 'cpu'
            return _o(*a, **k)
        setattr(torch, name, wrap)


def write_data_tree(d):
    os.makedirs(os.path.join(d, 'data/smpl'))
    os.makedirs(os.path.join(d, 'data/pretrained_model'))
    D = np.empty(2, dtype=object)
    D[0] = scipy.sparse.csr_matrix(ASSETS['Dmap0'].numpy())
    D[1] = scipy.sparse.csr_matrix(ASSETS['Dmap1'].numpy())
    A = np.empty(1, dtype=object)
    A[0] = scipy.sparse.eye(2).tocsr()
    np.savez(os.path.join(d, 'data/mesh_downsampling.npz'), A=A, U=A, D=D)
    s = ASSETS['smpl']
    with open(os.path.join(d, 'data/smpl/SMPL_NEUTRAL.pkl'), 'wb') as f:
        pickle.dump({'J_regressor': scipy.sparse.csc_matrix(s['J_regressor'].numpy())}, f)
    np.save(os.path.join(d, 'data/smpl/smpl_ssm.npy'), ASSETS['ssm'].numpy())
    np.savez(os.path.join(d, 'data/smpl_mean_params.npz'), **ASSETS['mean_params'])


# ----------------------------------------------------------------------------- main
def main():
    vit = install_stubs()
    patch_torch_cuda()
    sys.path.insert(0, REF)
    tmp = tempfile.mkdtemp()
    write_data_tree(tmp)
    os.chdir(tmp)
    from core.cfgs import cfg
    cfg.merge_from_file(os.path.join(REF, 'configs/pymaf_config.yaml'))
    import utils.geometry as RG

    real_load, real_lsd = torch.load, nn.Module.load_state_dict
    torch.load = lambda *a, **k: {'state_dict': {}}
    nn.Module.load_state_dict = lambda self, sd, strict=True: None
    import models.maf_extractor as RM
    init0 = RM.MAF_Extractor.__init__
    RM.MAF_Extractor.__init__ = lambda self, device=torch.device('cpu'): init0(self, device)
    import models.whmr as RW
    net = RW.whmr_net('data/smpl_mean_params.npz')
    torch.load, nn.Module.load_state_dict = real_load, real_lsd

    sd = synth.make_state_dict(0, ASSETS)
    res = net.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    assert not res.missing_keys, res.missing_keys          # the stubs register no extra params
    vit.ViT.train = lambda self, mode=True: nn.Module.train(self, mode)     # vit.py:338-341 returns None
    net.eval()

    # ---- full forward, B=2, with capture hooks
    inp = synth.make_inputs(2, 0, full_size=(224, 256))
    cap = {'ref_feature': [], 'reg': []}
    net.feature_extractor.register_forward_hook(lambda m, i, o: cap.__setitem__('s_feat', o.detach().clone()))
    fm = []
    for idx in (2, 5, 8):
        net.deconv_layers[idx].register_forward_hook(lambda m, i, o: fm.append(o.detach().clone()))
    net.est_Tz.register_forward_hook(lambda m, i, o: cap.__setitem__('Tz', 10.0 * o.detach().squeeze(-1)))
    for ext in net.maf_extractor:
        rd = ext.reduce_dim
        ext.reduce_dim = (lambda f, _rd=rd: (lambda y: (cap['ref_feature'].append(y.detach().clone()), y)[1])(_rd(f)))
    for reg in net.regressor:
        reg.register_forward_hook(lambda m, i, o: cap['reg'].append(o[0]))
    with torch.no_grad():
        out = net(inp['x'], None, inp['center'], inp['scale'], inp['bbox_height'], inp['orig_shape'], inp['bbox_info'],
                  is_train=False, J_regressor=None, full_x=inp['full_x'])
        taps = {}
        mine = OW.whmr_forward(sd, ASSETS, inp['x'], inp['center'], inp['scale'], inp['bbox_height'],
                               inp['orig_shape'], inp['bbox_info'], full_x=inp['full_x'], taps=taps)
        mine_train, _ = OW.whmr_forward(sd, ASSETS, inp['x'], inp['center'], inp['scale'], inp['bbox_height'],
                                        inp['orig_shape'], inp['bbox_info'], full_x=inp['full_x'], view='train')

    def chk(name, a, b, tol=2e-5):
        err = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)
        print('  %-28s max-rel %.2e' % (name, err))
        assert err < tol, name
    print('oracle vs imported reference:')
    for k in out:
        chk(k, mine[k], out[k])
    chk('s_feat', taps['s_feat0'], cap['s_feat'])
    for i in range(3):
        chk('fmap%d' % i, taps['fmaps'][i], fm[i])
        chk('ref_feature%d' % i, taps['ref_feature'][i], cap['ref_feature'][i])
        for k in ('theta', 'verts', 'kp_2d', 'kp_2d_w', 'kp_3d', 'smpl_kp_3d', 'rotmat', 'pred_cam_t', 'focal_length',
                  'markers', 'sub_verts', 'temp_verts'):
            chk('iter%d.%s' % (i, k), mine_train['smpl_out'][i + 1][k], cap['reg'][i][k])
    chk('Tz', taps['Tz'], cap['Tz'])

    g = {'in_' + k: v.numpy() for k, v in inp.items()}
    g.update({'out_' + k: v.numpy() for k, v in out.items()})
    g['s_feat'] = cap['s_feat'].numpy()
    g['Tz'] = cap['Tz'].numpy()
    pos = np.random.default_rng(0).integers(0, 10 ** 9, 1024)
    for i in range(3):
        flat = fm[i].reshape(-1)
        idx = torch.from_numpy(pos % flat.numel())
        g['fmap%d_idx' % i] = idx.numpy()
        g['fmap%d_val' % i] = flat[idx].numpy()
        g['fmap%d_sum' % i] = np.array([fm[i].double().sum().item(), fm[i].double().abs().sum().item()])
        g['ref_feature%d' % i] = cap['ref_feature'][i].numpy()
        for k in ('theta', 'verts', 'kp_2d', 'kp_2d_w', 'kp_3d', 'rotmat', 'pred_cam_t', 'focal_length', 'pose'):
            g['iter%d_%s' % (i, k)] = cap['reg'][i][k].numpy()
    # weight checksums: detects any drift of the synthetic generator between machines
    keys = sorted(k for k in sd if sd[k].dtype == torch.float32)
    g['weight_keys'] = np.array(keys)
    g['weight_sums'] = np.array([sd[k].double().sum().item() for k in keys])
    np.savez_compressed(os.path.join(HERE, 'whmr_b2.npz'), **g)

    # ---- ViT alone at 224x224 (BASELINE config #2 shape), B=2
    vsd = synth.make_vit_state(1, (224, 224))
    v = vit.ViT(img_size=(224, 224), patch_size=16, embed_dim=768, depth=12, num_heads=12, ratio=1, mlp_ratio=4,
                qkv_bias=True, drop_path_rate=0.3)
    v.load_state_dict(vsd, strict=True)
    nn.Module.train(v, False)
    xin = synth.make_inputs(2, 1, (224, 224))['x']
    with torch.no_grad():
        ref = v(xin)
        chk('vit224', vit_forward(vsd, xin), ref)
    np.savez_compressed(os.path.join(HERE, 'vit224_b2.npz'), x=xin.numpy(), s_feat=ref.numpy())

    # ---- BASELINE config #1: HMR (R50 trunk + iterative regressor), 1 x 224 x 224, CPU
    import models.hmr as RH
    from oracle.hmr import hmr_forward
    hsd = synth.make_hmr_state(0, ASSETS)
    hnet = RH.HMR(RH.Bottleneck, [3, 4, 6, 3], 'data/smpl_mean_params.npz')
    res = hnet.load_state_dict(hsd, strict=True)
    hnet.eval()
    xh = synth.make_inputs(1, 11, (224, 224))['x']
    with torch.no_grad():
        rot_r, shape_r, cam_r = hnet(xh)
        rot_m, shape_m, cam_m = hmr_forward(hsd, xh)
        vr = OS.smpl_forward(shape_r, rot_r, ASSETS['smpl'])[0]
    chk('hmr.rotmat', rot_m, rot_r)
    chk('hmr.shape', shape_m, shape_r)
    chk('hmr.cam', cam_m, cam_r)
    np.savez_compressed(os.path.join(HERE, 'hmr_b1.npz'), x=xh.numpy(), rotmat=rot_r.numpy(), shape=shape_r.numpy(),
                        cam=cam_r.numpy(), verts=vr.numpy())

    # ---- geometry helpers on edge cases
    gen = torch.Generator().manual_seed(0)
    R = OG.batch_rodrigues(torch.randn(64, 3, generator=gen) * 1.5)
    # 180-degree rotations and each quaternion branch
    special = torch.stack([torch.diag(torch.tensor(d)) for d in
                           ([1., 1., 1.], [1., -1., -1.], [-1., 1., -1.], [-1., -1., 1.])])
    near = OG.batch_rodrigues(torch.tensor([[3.1, 0.02, 0.01], [0.01, 3.12, 0.0], [0.0, 0.01, 3.13],
                                            [1e-4, 0, 0], [0, 0, 0], [2.0, -2.0, 0.5]]))
    Rall = torch.cat([R, special, near])
    aa_in = torch.cat([torch.randn(32, 3, generator=gen), torch.zeros(1, 3), torch.tensor([[1e-9, 0, 0], [3.14159, 0, 0]])])
    r6 = torch.randn(48, 6, generator=gen)
    m33 = torch.randn(2, 24, 3, 3, generator=gen) * 0.3 + torch.eye(3)
    pts = torch.randn(4, 50, 3, generator=gen) * 0.3
    cam = torch.cat([torch.rand(4, 1, generator=gen) + 0.5, torch.randn(4, 2, generator=gen) * 0.2], 1)
    tr = torch.cat([torch.randn(4, 2, generator=gen), torch.rand(4, 1, generator=gen) * 5 + 2], 1)
    fl = torch.rand(4, generator=gen) * 1000 + 500
    cc = torch.rand(4, 2, generator=gen) * 500
    # estimate_translation (trainer-only host stall): 3-D joints in front of the camera, 2-D joints = a noisy
    # projection, confidences with zeros (undetected joints)
    et_S = torch.randn(6, 49, 3, generator=gen) * 0.4
    et_t = torch.cat([torch.randn(6, 2, generator=gen) * 0.3, torch.rand(6, 1, generator=gen) * 6 + 3], 1)
    et_p = et_S + et_t[:, None]
    et_xy = 5000. * et_p[..., :2] / et_p[..., 2:] + 112. + torch.randn(6, 49, 2, generator=gen)
    et_conf = (torch.rand(6, 49, 1, generator=gen) > 0.3).float() * torch.rand(6, 49, 1, generator=gen)
    et_j2d = torch.cat([et_xy, et_conf], -1)
    geo = {'R': Rall, 'aa_in': aa_in, 'r6': r6, 'm33': m33, 'pts': pts, 'cam': cam, 'tr': tr, 'fl': fl, 'cc': cc, 'et_S': et_S,
           'et_j2d': et_j2d}
    ref = {
        'aa': RG.rotation_matrix_to_angle_axis(Rall),
        'rod': RG.batch_rodrigues(aa_in),
        'r6_to_R': RG.rot6d_to_rotmat(r6),
        'gs': RG.unbiased_gram_schmidt(m33),
        'proj': RG.projection(pts, cam),
        'persp': RG.perspective_projection(pts, torch.eye(3).unsqueeze(0).expand(4, -1, -1), tr, fl, cc),
        'full_cam': RG.convert_pare_to_full_img_cam(cam, fl, cc, torch.full((4,), 1280.), torch.full((4,), 720.), Tz=tr[:, 2]),
        'rot6d': RG.rotmat_to_rot6d(Rall),
        'est_trans': RG.estimate_translation(et_S, et_j2d, focal_length=5000., img_size=[224., 224.]),
    }
    my = {
        'aa': OG.rotation_matrix_to_angle_axis(Rall), 'rod': OG.batch_rodrigues(aa_in), 'r6_to_R': OG.rot6d_to_rotmat(r6),
        'gs': OG.unbiased_gram_schmidt(m33), 'proj': OG.projection(pts, cam),
        'persp': OG.perspective_projection(pts, torch.eye(3).unsqueeze(0), tr, fl, cc),
        'full_cam': OG.convert_pare_to_full_img_cam(cam, fl, cc, torch.full((4,), 1280.), torch.full((4,), 720.), tr[:, 2]),
        'rot6d': OG.rotmat_to_rot6d(Rall),
        'est_trans': OG.estimate_translation(et_S, et_j2d, 5000., (224., 224.)),
    }
    for k in ref:
        chk('geometry.' + k, my[k], ref[k], 1e-6)
    # MAF sampler incl. points outside [-1,1] (zero padding) and exact-texel coordinates
    ext = net.maf_extractor[1]
    fmap = torch.randn(2, 256, 8, 6, generator=gen)
    p2 = torch.rand(2, 40, 2, generator=gen) * 2.6 - 1.3
    p2[:, :6] = torch.tensor([[-1., -1.], [1., 1.], [-1., 1.], [0., 0.], [1.0001, 0.], [-0.2, 1.5]])
    ext.reduce_dim = rd                       # un-hook (rd is extractor 2's; same weights are not needed here)
    with torch.no_grad():
        pf = F.grid_sample(fmap, p2.unsqueeze(2), align_corners=True)[..., 0]
        y_ref = RM.MAF_Extractor.reduce_dim(ext, pf)
        y_my, pf_my = OW.maf_sampling(sd, p2, fmap, 'maf_extractor.1.')
    chk('maf.point_feat', pf_my, pf, 1e-6)
    chk('maf.reduce_dim', y_my, y_ref, 1e-5)
    geo.update({'maf_fmap': fmap, 'maf_pts': p2})
    ref.update({'maf_pf': pf, 'maf_y': y_ref})
    np.savez_compressed(os.path.join(HERE, 'geometry.npz'), **{'in_' + k: v.numpy() for k, v in geo.items()},
                        **{'out_' + k: v.numpy() for k, v in ref.items()})
    print('fixtures written to', HERE)


if __name__ == '__main__':
    main()
