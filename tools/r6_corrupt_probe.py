"""Does the side stream's backward write into memory the regressor loop still owns?  Forward of the batch-64 bf16 step, snapshot of the stage-3 SMPL
node's saved activations, backward of a loss that does NOT reach stage 3 (so they stay alive and unread), compare.  usage: python tools/r6_corrupt_probe.py [iters]"""
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
m = whmr_net(None, assets=assets, numerics='bf16')
m.load_state_dict(sd, strict=False)
m = m.to(dev).train()
for mod in m.modules():
    if isinstance(mod, torch.nn.Dropout):
        mod.p = 0.0
m.feature_extractor.backbone.drop_path_rate = 0.0
names = ('betas', 'rot', 'A', 'pose_off')
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    for p_ in m.parameters():
        p_.grad = None
    out_list, _ = m(d['x'], None, d['center'], d['scale'], d['bbox_height'], d['orig_shape'], d['bbox_info'], is_train=True)
    so = out_list['smpl_out']
    watch = {}
    for l in (1, 2, 3):
        fn = so[l]['verts'].grad_fn
        for n, t in zip(names, fn.saved):
            watch[(l, n)] = (t, t.clone())
    torch.cuda.synchronize()
    # stage 3 (and, in the second half of the iterations, every stage) kept out of the loss: the saved activations are never consumed
    keep = so[:3] if it % 2 == 0 else so[:1]
    loss = OT.dp_cotangent_loss(out_list['dp_out'][0], dev=dev) + (OT.cotangent_loss(keep, dev=dev) if len(keep) > 1 else 0.0)
    # Tz reaches the loss through the stages' projections; with no stage in the loss, add it directly
    if len(keep) <= 1:
        loss = loss + (so[3]['focal_length'] * 1e-3).sum()
    loss.backward()
    torch.cuda.synchronize()
    rep = []
    for (l, n), (t, c) in watch.items():
        if fn is not None and t is not None and not torch.equal(t, c):
            dd = (t - c).abs().flatten()
            nz = torch.nonzero(dd).flatten()
            rep.append('stage %d %s: %d of %d elements changed, first at %d..%d, max |d| %.3e' % (l, n, nz.numel(), dd.numel(), nz[0].item(), nz[-1].item(), dd.max().item()))
    print('iter %d (%s in loss):' % (it, 'stages 1-2' if it % 2 == 0 else 'focal of stage 3 only'), '; '.join(rep) if rep else 'saved SMPL activations intact')
