"""Which parameter gradients of the batch-64 bf16 training step differ between two runs from the same state (same process, fresh model each)?
usage: python tools/r6_det_probe.py [runs]   (switches through the WHMR_TRAIN_* environment variables)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import synth
from oracle import train as OT
from whmr_amd.models import whmr_net

dev = torch.device('cuda:0')
assets = synth.make_assets(0)
sd = synth.make_state_dict(0, assets)
B = 64
inp = synth.make_inputs(B, 3)
d = {k: inp[k].to(dev) for k in ('x', 'center', 'scale', 'bbox_height', 'orig_shape', 'bbox_info')}
runs = []
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    m = whmr_net(None, assets=assets, numerics='bf16')
    m.load_state_dict(sd, strict=False)
    m = m.to(dev).train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    m.feature_extractor.backbone.drop_path_rate = 0.0
    out_list, _ = m(d['x'], None, d['center'], d['scale'], d['bbox_height'], d['orig_shape'], d['bbox_info'], is_train=True)
    loss = OT.cotangent_loss(out_list['smpl_out'], dev=dev) + OT.dp_cotangent_loss(out_list['dp_out'][0], dev=dev)
    loss.backward()
    torch.cuda.synchronize()
    runs.append({k: v.grad.detach().clone() for k, v in m.named_parameters() if v.grad is not None})
for r in range(1, len(runs)):
    bad = []
    for k, g in runs[0].items():
        if not torch.equal(g, runs[r][k]):
            e = ((g.double() - runs[r][k].double()).abs().max() / g.double().abs().max().clamp_min(1e-30)).item()
            bad.append((k, e))
    heads = [b for b in bad if b[0].startswith(('regressor', 'est_Tz', 'dp_head', 'conv', 'transformer_decoder', 'maf_extractor', 'deconv'))]
    print('run %d vs 0: %d of %d differ; head keys:' % (r, len(bad), len(runs[0])), ' '.join('%s=%.1e' % b for b in heads[:60]))
    k = 'regressor.2.decpose.bias'
    if not torch.equal(runs[0][k], runs[r][k]):
        dd = (runs[0][k].double() - runs[r][k].double()).abs().cpu()
        print('   decpose.bias: differing entries', torch.nonzero(dd).flatten().tolist()[:80], 'max', dd.max().item(), 'of', runs[0][k].abs().max().item())
        for k2 in ('regressor.2.deccam.bias', 'regressor.2.decshape.bias'):
            print('   ', k2, torch.equal(runs[0][k2], runs[r][k2]))
