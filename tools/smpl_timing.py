"""One SMPL call (regressor-stage form) at several batch sizes: the three-launch product form vs the five per-phase launches of round 2, 20 calls per HIP graph
(+ phase stamps of the blend + skin launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whmr_amd.utils import synth
from whmr_amd.models.smpl import SMPL

dev = torch.device('cuda:0')
assets = synth.make_assets(0)
m = SMPL(arrays=assets['smpl'], marker_ids=assets['ssm']).to(dev)
for B in (1, 8, 64, 256):
    g = torch.Generator().manual_seed(B)
    betas = torch.randn(B, 10, generator=g).to(dev)
    rot = (torch.eye(3).expand(B, 24, 3, 3) + 0.1 * torch.randn(B, 24, 3, 3, generator=g)).contiguous().to(dev)
    res = {}
    for fused in (False, 'csr'):
        m.csr_tail, m.blend_skin = fused == 'csr', fused == 'csr'
        fn = lambda: m.run(betas, rot, gram_schmidt=True, want_aa=True, want_smpl_joints=True, want_markers=True)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                fn()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(20):
                fn()
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record()
        torch.cuda.synchronize()
        res[fused] = e0.elapsed_time(e1) / 20 * 1e3
    from whmr_amd import _lib as L
    fn = lambda: m.run(betas, rot, gram_schmidt=True, want_aa=True, want_smpl_joints=True, want_markers=True)
    m.csr_tail, m.blend_skin = True, True
    buf = torch.zeros(16, dtype=torch.int32, device=dev)
    L.lib().whmr_smpl_blend_skin_stamps(buf.data_ptr())
    fn(); torch.cuda.synchronize()
    L.lib().whmr_smpl_blend_skin_stamps(None)
    bs = buf.cpu()[2:10].view(torch.int64).tolist()
    print('      blend+skin launch, workgroup 0 (us): loads + staging %.1f | fp32-MFMA offsets %.1f | shape blend + skinning %.1f' % tuple((bs[i + 1] - bs[i]) / 100.0 for i in range(3)))
    byt = B * 84172.0 + 19.6e6
    print('B %4d: five launches (round 2) %.1f us, three launches (blend+skin, CSR tail) %.1f us = %.1f %% of 8 TB/s' % (B, res[False], res['csr'], byt / res['csr'] / 1e6 / 8 * 100), flush=True)
