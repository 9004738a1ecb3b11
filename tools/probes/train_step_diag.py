"""Diagnostic: per-parameter gradient error (max-rel and RMS-rel) of WHMR.forward(is_train=True) vs the CPU oracle, both TRAIN.STAGE layouts."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import torch
from oracle import synth, train as OT
from whmr_amd.core.cfgs import cfg
import test_train_gpu as T
dev = torch.device('cuda:0')
numerics = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
assets = synth.make_assets(0)
sd = synth.make_state_dict(0, assets)
inp = synth.make_inputs(2, 0)
keys = T._grad_keys(sd)
for stage in (2, 1):
    p = {k: (v.clone().requires_grad_(True) if k in keys else v) for k, v in sd.items()}
    outs_ref = OT.whmr_forward_train(p, assets, inp['x'], inp['center'], inp['scale'], inp['bbox_height'], inp['orig_shape'], inp['bbox_info'], stage=stage)
    OT.cotangent_loss(outs_ref).backward()
    m = T._train_model(assets, sd, numerics, dev)
    cfg.TRAIN.STAGE = stage
    d = {k: inp[k].to(dev) for k in ('x', 'center', 'scale', 'bbox_height', 'orig_shape', 'bbox_info')}
    out_list, vis = m(d['x'], None, d['center'], d['scale'], d['bbox_height'], d['orig_shape'], d['bbox_info'], is_train=True)
    OT.cotangent_loss(out_list['smpl_out'], dev=dev).backward()
    named = dict(m.named_parameters())
    rows = []
    for k in keys:
        a, b = named[k].grad.double().cpu(), p[k].grad.double()
        rows.append((((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item(), ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-30)).item(),
                     b.abs().max().item(), a.abs().max().item(), k))
    rows.sort(reverse=True)
    print('stage', stage, numerics)
    for r in rows[:14]:
        print('  max-rel %.2e  rms-rel %.2e  |ref|max %.2e |hip|max %.2e  %s' % r)
    for r in rows:
        if r[4] in ('conv.0.weight', 'conv.1.weight', 'est_Tz.0.weight', 'regressor.2.fc1.weight', 'regressor.0.fc1.weight', 'maf_extractor.2.conv0.weight', 'maf_extractor.0.conv0.weight', 'deconv_layers.7.weight', 'deconv_layers.1.weight'):
            print('  *max-rel %.2e  rms-rel %.2e  |ref|max %.2e |hip|max %.2e  %s' % r)
    import numpy as np
    print('  median max-rel %.2e' % float(np.median([r[0] for r in rows])))
