import sys; sys.path.insert(0, '/root/repo')
import torch
from torch.profiler import profile, ProfilerActivity
from whmr_amd.utils import synth
from whmr_amd.models import whmr_net
dev = torch.device('cuda:0')
assets = synth.make_assets(0); sd = synth.make_state_dict(0, assets)
m = whmr_net(None, assets=assets, numerics='bf16'); m.load_state_dict(sd, strict=False); m = m.to(dev).eval()
inp = {k: v.to(dev) for k, v in synth.make_inputs(8, 7).items()}
a = (inp['x'], None, inp['center'], inp['scale'], inp['bbox_height'], inp['orig_shape'], inp['bbox_info'])
for _ in range(3): m(*a)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    m(*a)
    torch.cuda.synchronize()
evs = [e for e in prof.events() if 'copy' in e.name.lower() or 'Memcpy' in e.name or 'memcpy' in e.name.lower()]
from collections import Counter
c = Counter()
for e in evs:
    st = [s for s in (e.stack or []) if 'w-hmr_amd' in s or 'whmr_amd' in s]
    c[(e.name, st[0] if st else '?')] += 1
for k, v in c.most_common(40):
    print(v, k)
