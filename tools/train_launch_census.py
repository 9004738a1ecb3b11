"""Which operator issues the small framework launches (fills, copies, element-wise kernels) of ONE W-HMR training step?  torch.profiler with stacks:
every device kernel that is not one of libwhmr_hip.so's is attributed to its aten operator and to the innermost frame inside this repository.
   python tools/train_launch_census.py [top_n]"""
import os
import sys
from collections import Counter

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = bench.parse(['--workload', 'whmr_train', '--no-cpu'])
step = bench.build_workload(args, torch.device('cuda:0'))[0]
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=False) as prof:
    step()
    torch.cuda.synchronize()
ours = ('gemm_', 'attention', 'layernorm', 'bn_', 'smpl_', 'maf_', 'tn_', 'raster', 'iuv_', 'csr_', 'regressor_', 'colsum', 'transpose', 'cast_f32',
        'im2col', 'col2im', 'splitk', 'tz_', 'gelu', 'patch_', 'mat_to_aa', 'rot_to_mat', 'weights_prepare', 'scale_rows', 'orient', 'cam_head', 'split3')
cnt, tim = Counter(), Counter()
events = prof.events()
for e in events:
    if e.device_type is not None and str(e.device_type).endswith('CUDA') or getattr(e, 'is_legacy', False):
        continue
for e in events:
    ks = getattr(e, 'kernels', None)
    if not ks:
        continue
    fw = [k for k in ks if not any(k.name.startswith(o) or ('void ' + o) in k.name for o in ours)]
    if not fw:
        continue
    where = '?'
    for fr in (e.stack or []):
        if ROOT in fr and '/tools/' not in fr and 'torch/' not in fr:
            where = fr.replace(ROOT + '/', '').split(',')[0].strip()
            break
    key = (e.name, where)
    cnt[key] += len(fw)
    tim[key] += sum(getattr(k, 'duration', 0) for k in fw)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
print('%d framework kernel launches in one step, %.1f us of device time' % (sum(cnt.values()), sum(tim.values())))
for key, c in cnt.most_common(n):
    print('%4d  %8.1f us  %-38s %s' % (c, tim[key], key[0][:38], key[1][:110]))
