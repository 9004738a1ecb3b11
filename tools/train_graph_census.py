"""Histogram of the autograd node types in the graph of ONE W-HMR training step's loss (which torch-native nodes sit between the HIP nodes: every
SliceBackward / SelectBackward / IndexBackward is a zero fill + a copy in the backward pass).   python tools/train_graph_census.py"""
import os
import sys
from collections import Counter
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whmr_amd.models import whmr_net
from whmr_amd.utils import synth

dev = torch.device('cuda:0')
assets = synth.make_assets(0)
sd = synth.make_state_dict(0, assets, with_cam_model=False)
m = whmr_net(None, assets=assets, numerics='bf16')
m.load_state_dict(sd, strict=False)
m = m.to(dev).train()
inp = {k: v.to(dev) for k, v in synth.make_inputs(64, 7).items()}
out, _ = m(inp['x'], None, inp['center'], inp['scale'], inp['bbox_height'], inp['orig_shape'], inp['bbox_info'], is_train=True)
keys = ('rotmat', 'pred_shape', 'pred_cam', 'kp_2d', 'kp_2d_w', 'kp_3d', 'verts', 'sub_verts', 'temp_verts')
loss = sum(out['smpl_out'][l][k].float().pow(2).mean() for l in range(1, 4) for k in keys)
seen, todo, cnt = set(), [loss.grad_fn], Counter()
where = {}
while todo:
    fn = todo.pop()
    if fn is None or fn in seen:
        continue
    seen.add(fn)
    name = type(fn).__name__
    cnt[name] += 1
    for nxt, _ in fn.next_functions:
        todo.append(nxt)
print('%d autograd nodes' % len(seen))
for name, n in cnt.most_common(40):
    print('%4d  %s' % (n, name))
