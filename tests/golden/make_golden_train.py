"""Pin the TRAINING-mode oracle (oracle/train.py) against the REAL reference and write tests/golden/whmr_train_b2.npz.

Build container only (needs /root/reference).  Same stub recipe as make_golden.py (whose helpers are reused): the reference's own
models/whmr.py is imported unmodified, put in ``.train()`` with the Dropout probabilities set to 0 (the masks are random draws),
run on the seeded B=2 inputs with ``is_train=True`` for both ``cfg.TRAIN.STAGE`` layouts; forward hooks on the three Regressor
modules hand out the per-stage output dicts WITH their autograd graph (the released forward only returns ``vis_dict``).  The scalar
``oracle.train.cotangent_loss`` of those outputs is back-propagated through the reference, and
  * every per-stage output, ``global_output`` (whmr.py:630-654: hooks on the global-orientation head and the last SMPL call), the BatchNorm running
    statistics after the step and every parameter gradient -- ``global_orient.*`` included, through the third cotangent
    ``oracle.train.global_cotangent_loss`` -- are asserted equal to the oracle's (functional restatement + torch autograd);
  * the fixture stores the loss, per-parameter gradient (L2 norm, sum) pairs for all trained parameters, a few small gradients in full
    and the updated running statistics -- plain arrays only.
Usage:  python tests/golden/make_golden_train.py
"""
import importlib.util
import os
import sys
import tempfile

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location('make_golden', os.path.join(HERE, 'make_golden.py'))
MG = importlib.util.module_from_spec(spec)
spec.loader.exec_module(MG)

from oracle import synth                     # noqa: E402
from oracle import train as OT               # noqa: E402

FULL_KEYS = ('dp_head.predict_ann_index.bias', 'regressor.2.deccam.weight', 'regressor.0.decshape.bias', 'est_Tz.0.weight', 'deconv_layers.7.weight', 'deconv_layers.1.bias',
             'maf_extractor.2.conv2.bias', 'maf_extractor.0.conv2.weight', 'transformer_decoder.norm1.weight', 'conv.1.weight',
             'feature_extractor.backbone.last_norm.weight', 'feature_extractor.backbone.blocks.0.attn.qkv.bias', 'global_orient.decrot.weight',
             'global_orient.fc2.bias')
SKIP = ('running', 'cam_model', 'smpl', 'Dmap', 'points_grid', 'init_', 'num_batches')


def grad_keys(sd):
    return [k for k, v in sd.items() if v.is_floating_point() and not any(s in k for s in SKIP)]


def main():
    vit = MG.install_stubs()
    MG.patch_torch_cuda()
    sys.path.insert(0, MG.REF)
    tmp = tempfile.mkdtemp()
    MG.write_data_tree(tmp)
    os.chdir(tmp)
    from core.cfgs import cfg
    cfg.merge_from_file(os.path.join(MG.REF, 'configs/pymaf_config.yaml'))
    real_load, real_lsd = torch.load, nn.Module.load_state_dict
    torch.load = lambda *a, **k: {'state_dict': {}}
    nn.Module.load_state_dict = lambda self, sd, strict=True: None
    import models.maf_extractor as RM
    init0 = RM.MAF_Extractor.__init__
    RM.MAF_Extractor.__init__ = lambda self, device=torch.device('cpu'): init0(self, device)
    import models.whmr as RW
    net = RW.whmr_net('data/smpl_mean_params.npz')
    torch.load, nn.Module.load_state_dict = real_load, real_lsd
    vit.ViT.train = lambda self, mode=True: nn.Module.train(self, mode)     # vit.py:338-341 returns None
    sd = synth.make_state_dict(0, MG.ASSETS)
    inp = synth.make_inputs(2, 0, full_size=(224, 256))
    keys = grad_keys(sd)
    cap = []
    for reg in net.regressor:
        reg.register_forward_hook(lambda m, i, o: cap.append(o[0]))
    dp_cap = []
    net.dp_head.register_forward_hook(lambda m, i, o: dp_cap.append(o))
    # global_output (whmr.py:630-654) is built inside forward and not returned in training: the global-orientation head's output (with graph), its
    # cam_rotmat input and the LAST call of regressor[0].smpl (the global mesh) come from hooks; global_pose is re-assembled below with the
    # reference's own rotation_matrix_to_angle_axis, exactly as whmr.py:632-633 does
    go_cap, smpl_cap = [], []
    net.global_orient.register_forward_hook(lambda m, i, o: go_cap.append((i[1], o)))
    net.regressor[0].smpl.register_forward_hook(lambda m, i, o: smpl_cap.append(o))
    from utils.geometry import rotation_matrix_to_angle_axis as ref_mat_to_aa
    fixture = {'grad_keys': np.array(keys)}

    def rel(a, b):
        return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)

    # third run: TRAIN.STAGE 2 WITH stochastic depth (vit.py:132-139,233; drop_path_rate 0.3): seeded keep masks, fed to the reference through
    # the drop_path stub's queue (block 0 has dpr = 0 -> nn.Identity, no call) and to the oracle as ``drop_masks``
    depth, rate = 12, 0.3
    gen = torch.Generator().manual_seed(11)
    dpr = torch.linspace(0, rate, depth)
    drop_masks = torch.floor((1 - dpr).repeat_interleave(2).view(-1, 1) + torch.rand(2 * depth, 2, generator=gen))
    assert 0 < drop_masks[2:].sum() < drop_masks[2:].numel()              # some branches dropped, some kept
    fixture['drop_masks'] = drop_masks.numpy().copy()
    for stage, with_dp in ((2, False), (1, False), (2, True)):
        cfg.TRAIN.STAGE = stage
        tag = 'stage%d%s' % (stage, '_droppath' if with_dp else '')
        del MG.DROP_QUEUE[:]
        if with_dp:
            MG.DROP_QUEUE.extend(drop_masks[i] for i in range(2, 2 * depth))
        res = net.load_state_dict(sd, strict=False)
        assert not res.unexpected_keys and not res.missing_keys
        net.train()
        net.cam_model.eval()                          # frozen calibration head (its output is detached, whmr.py:509-524)
        for mod in net.modules():
            if isinstance(mod, nn.Dropout):
                mod.p = 0.0
        net.zero_grad()
        del cap[:]
        del dp_cap[:]
        del go_cap[:]
        del smpl_cap[:]
        net(inp['x'], None, inp['center'], inp['scale'], inp['bbox_height'], inp['orig_shape'], inp['bbox_info'], is_train=True,
            J_regressor=None, full_x=inp['full_x'])
        assert len(cap) == 3
        loss_ref = OT.cotangent_loss([None] + cap)
        assert len(dp_cap) == 1                      # AUX_SUPV_ON: the IUV head ran on the last feature map (whmr.py:656-658)
        loss_dp_ref = OT.dp_cotangent_loss(dp_cap[0])
        assert len(go_cap) == 1 and go_cap[0][1].requires_grad and not go_cap[0][0].requires_grad       # cam_rotmat comes out of torch.no_grad()
        cam_rotmat_ref, g_rot_ref = go_cap[0]
        g_ref = {'global_pose': torch.cat([ref_mat_to_aa(g_rot_ref.reshape(-1, 3, 3)).reshape(-1, 3), cap[2]['pose'][:, 3:]], dim=1),
                 'global_kp_3d': smpl_cap[-1].joints, 'global_verts': smpl_cap[-1].vertices}
        loss_g_ref = OT.global_cotangent_loss(g_ref)
        (loss_ref + loss_dp_ref + loss_g_ref).backward()
        ref_named = dict(net.named_parameters())
        ref_state = net.state_dict()

        p = {k: (v.clone().requires_grad_(True) if k in keys else v) for k, v in sd.items()}
        stats, dp, gout = {}, [], []
        assert not MG.DROP_QUEUE                     # the reference consumed every mask: 2 per block with dpr > 0
        from oracle import whmr as OW
        with torch.no_grad():
            cam_rotmat, _ = OW.cam_model_forward(sd, inp['full_x'])          # the oracle's own camera head (pinned by make_golden.py); whmr.py:509-524
        assert rel(cam_rotmat, cam_rotmat_ref) < 2e-5
        outs = OT.whmr_forward_train(p, MG.ASSETS, inp['x'], inp['center'], inp['scale'], inp['bbox_height'], inp['orig_shape'],
                                     inp['bbox_info'], stage=stage, stats=stats, dp_out=dp,
                                     drop_masks=drop_masks if with_dp else None, drop_path_rate=rate if with_dp else 0.0,
                                     global_out=gout, cam_rotmat=cam_rotmat_ref)
        loss = OT.cotangent_loss(outs)
        loss_dp = OT.dp_cotangent_loss(dp[0])
        loss_g = OT.global_cotangent_loss(gout[0])
        (loss + loss_dp + loss_g).backward()
        for k in OT.GLOBAL_LOSS_KEYS:
            assert rel(gout[0][k].detach(), g_ref[k].detach()) < 2e-5, k
        assert abs(loss_g.item() - loss_g_ref.item()) < 1e-5 * max(1.0, abs(loss_g_ref.item()))
        for k in dp[0]:
            assert rel(dp[0][k].detach(), dp_cap[0][k].detach()) < 2e-5, k
        assert abs(loss_dp.item() - loss_dp_ref.item()) < 1e-5 * max(1.0, abs(loss_dp_ref.item()))
        print('TRAIN.STAGE %d%s: loss reference %.8f oracle %.8f' % (stage, ' + stochastic depth' if with_dp else '', loss_ref.item(), loss.item()))
        assert abs(loss_ref.item() - loss.item()) < 1e-5 * max(1.0, abs(loss_ref.item()))
        for l in range(3):
            for k in OT.TRAIN_LOSS_KEYS + ('theta', 'pred_cam_t', 'smpl_kp_3d', 'markers'):
                e = rel(outs[l + 1][k].detach(), cap[l][k].detach())
                assert e < 2e-5, (stage, l, k, e)
        for k, v in stats.items():
            assert rel(v, ref_state[k]) < 2e-5, k
        worst = 0.0
        for k in keys:
            g_ref = ref_named[k].grad
            if g_ref is None:
                assert p[k].grad is None or p[k].grad.abs().max() == 0, k
                continue
            if g_ref.abs().max() < 1e-8:              # bias in front of a batch-statistics BatchNorm: zero up to rounding
                assert p[k].grad.abs().max() < 1e-6, k
                continue
            e = rel(p[k].grad, g_ref)
            worst = max(worst, e)
            assert e < 5e-4, (stage, k, e)
        untouched = [k for k, q in ref_named.items() if q.grad is None and not k.startswith('cam_model')]
        print('  outputs, running stats and %d parameter gradients agree (worst max-rel %.2e); no gradient in the reference for: %s'
              % (len(keys), worst, sorted(set(k.split('.')[0] for k in untouched))))
        fixture['loss_%s' % tag] = np.array(loss_ref.item())
        fixture['loss_dp_%s' % tag] = np.array(loss_dp_ref.item())
        fixture['loss_global_%s' % tag] = np.array(loss_g_ref.item())
        fixture['cam_rotmat'] = cam_rotmat_ref.numpy().copy()
        fixture['grad_norm_sum_%s' % tag] = np.array([[ref_named[k].grad.double().norm().item(), ref_named[k].grad.double().sum().item()]
                                                             for k in keys])
        for k in FULL_KEYS:
            fixture['grad_%s/%s' % (tag, k)] = ref_named[k].grad.numpy().copy()
        for k in stats:
            fixture['stat_%s/%s' % (tag, k)] = ref_state[k].numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'whmr_train_b2.npz'), **fixture)
    print('wrote whmr_train_b2.npz')


if __name__ == '__main__':
    main()
