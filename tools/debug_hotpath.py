import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = ['maf_nchw', 'maf_direct', 'smpl', 'whmr']
if len(sys.argv) == 1:
    for c in CASES:
        r = subprocess.run([sys.executable, __file__, c], capture_output=True, text=True)
        print('==', c, 'rc', r.returncode, '\n', r.stdout[-1500:], r.stderr[-1500:])
    sys.exit(0)
import numpy as np, torch
from oracle import synth
dev = torch.device('cuda:0')
case = sys.argv[1]
assets = synth.make_assets(0)
if case.startswith('maf'):
    from whmr_amd.models.maf_extractor import MAF_Extractor
    g = {k: torch.from_numpy(v) for k, v in np.load('tests/golden/geometry.npz').items()}
    ext = MAF_Extractor().to(dev)
    fmap, pts = g['in_maf_fmap'].to(dev), g['in_maf_pts'].to(dev)
    if case == 'maf_nchw':
        y, pf = ext.sampling(pts, fmap); torch.cuda.synchronize(); print('ok', y.shape, pf.shape)
    else:
        y = ext.reduce_dim(g['out_maf_pf'].to(dev)); torch.cuda.synchronize(); print('ok', y.shape)
elif case == 'smpl':
    from whmr_amd.models.smpl import SMPL
    m = SMPL(arrays=assets['smpl'], marker_ids=assets['ssm']).to(dev)
    B = 2
    out = m.run(torch.randn(B, 10, device=dev), torch.eye(3, device=dev).expand(B, 24, 3, 3).contiguous(), want_aa=True, want_smpl_joints=True, want_markers=True)
    torch.cuda.synchronize(); print('ok', out.vertices.shape)
else:
    from whmr_amd.models import whmr_net
    sd = synth.make_state_dict(0, assets)
    m = whmr_net(None, assets=assets, numerics='fp32'); m.load_state_dict(sd, strict=False); m = m.to(dev)
    inp = synth.make_inputs(2, 0, full_size=(224, 256))
    kw = {k: v.to(dev) for k, v in inp.items()}
    out = m(kw['x'], None, kw['center'], kw['scale'], kw['bbox_height'], kw['orig_shape'], kw['bbox_info'], full_x=kw['full_x'])
    torch.cuda.synchronize(); print('ok', {k: tuple(v.shape) for k, v in out.items()})
