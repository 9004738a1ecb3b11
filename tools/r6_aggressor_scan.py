"""Does any kernel of the library disturb `v_pk_fma_f32 ... op_sel:[0,1,0]` on another stream the way the (removed) 64-row TN tile did?  The one-instruction
canary (whmr_debug_pkfma_canary) runs back to back on a side stream while the main stream runs whole workloads: the ViT forward in the three numerics, the
full W-HMR forward, the training step (forward + backward).  usage: python tools/r6_aggressor_scan.py [steps]   -> wrong lanes per workload (0 expected)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import synth
from oracle import train as OT
from whmr_amd import _lib as L
from whmr_amd.models import whmr_net

dev = torch.device('cuda:0')
assets = synth.make_assets(0)
sd = synth.make_state_dict(0, assets)
B = 64
inp = synth.make_inputs(B, 3)
d = {k: inp[k].to(dev) for k in ('x', 'center', 'scale', 'bbox_height', 'orig_shape', 'bbox_info')}
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
side = L.side_stream(dev, 3)
rep = torch.zeros(4, dtype=torch.int32, device=dev)


def scan(name, fn, canaries):
    fn()                                                     # warm-up (caches, workspaces)
    torch.cuda.synchronize()
    rep.zero_()
    lanes = 0
    for _ in range(steps):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(canaries):
                L._check(L.lib().whmr_debug_pkfma_canary(1024, 1500, rep.data_ptr(), L._stream()), 'pkfma')
        lanes += canaries * 1024 * 128
        fn()
        torch.cuda.synchronize()
    r = rep.cpu().tolist()
    print('%-34s: plain form %d / %d wrong low / high lanes, op_sel:[0,1,0] form %d / %d (of %d lanes, %d canary launches beside %d steps)' % (
        name, r[0], r[1], r[2], r[3], lanes, steps * canaries, steps))


def train_step(m):
    def fn():
        for p_ in m.parameters():
            p_.grad = None
        out_list, _ = m(d['x'], None, d['center'], d['scale'], d['bbox_height'], d['orig_shape'], d['bbox_info'], is_train=True)
        loss = OT.cotangent_loss(out_list['smpl_out'], dev=dev) + OT.dp_cotangent_loss(out_list['dp_out'][0], dev=dev)
        loss.backward()
    return fn


if os.environ.get('SCAN') == 'tiles':        # the row-major bf16 GEMM kernel tile by tile (plain operands, ResNet-like shapes), 100 launches per window
    g = torch.Generator().manual_seed(0)
    for tile, (M, N, K) in ((65, (30000, 64, 576)), (65, (30000, 64, 64)), (64, (30000, 128, 1152)), (128, (30000, 256, 1152)), (192, (12544, 768, 768)),
                            (257, (12544, 2304, 768)), (256, (12544, 768, 768))):
        a = (torch.randn(M, K, generator=g) * 0.5).to(dev).bfloat16()
        w = (torch.randn(N, K, generator=g) * 0.05).to(dev).bfloat16()
        o = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        scan('gemm_bf16_big tile %d, %d x %d x %d, x 100' % (tile, M, N, K), lambda: [L.gemm(a, w, o, tile=tile) for _ in range(100)], 60)
    sys.exit(0)
if os.environ.get('SCAN') == 'parts':        # the bf16 forward piece by piece: backbone, one sampler launch, one SMPL call (each repeated to fill the window)
    m = whmr_net(None, assets=assets, numerics='bf16')
    m.load_state_dict(sd, strict=False)
    m = m.to(dev).eval()
    with torch.no_grad():
        out, _ = m(d['x'], None, d['center'], d['scale'], d['bbox_height'], d['orig_shape'], d['bbox_info'], view='train')
        last = out['smpl_out'][2]
        markers, cam = last['markers'].contiguous(), last['pred_cam'].contiguous()
        betas, rot = last['pred_shape'].contiguous(), last['rotmat'].reshape(B, 216).contiguous()
        ext, smpl = m.maf_extractor[2], m.regressor[2].smpl
        xc = torch.empty(B, m.regressor[2].fc1.in_features, dtype=torch.float32, device=dev)
        scan('ViT-B backbone (blocked bf16 kernels)', lambda: m.feature_extractor(d['x']), 80)
        scan('sampler launch x 100', lambda: [ext(markers, cam=cam, out=xc, want_point_feat=False) for _ in range(100)], 40)
        scan('SMPL call x 40', lambda: [smpl.run(betas, rot, gram_schmidt=True, want_aa=True, want_smpl_joints=True, want_markers=True) for _ in range(40)], 50)
        f = m.feature_extractor(d['x']).permute(0, 2, 3, 1).contiguous().to(m._dt)
        def heads():
            x_, sp = f, None
            for i in range(3):
                x_, sp = m._deconv(i, x_, sp)
            return x_
        scan('three deconv stages x 1', heads, 30)
        fm = heads()
        scan('composed Tz convolution x 20', lambda: [m._tz_tokens_composed(fm, None) for _ in range(20)], 60)
        t0 = m._tz_tokens_composed(fm, None)
        g_, bn4 = m._tz_composed_operands()
        scan('Tz tail x 40', lambda: [m._tz_tokens_tail(t0.clone(), B, dev, bn4) for _ in range(40)], 60)
        scan('full forward again (all pieces)', lambda: m(d['x'], None, d['center'], d['scale'], d['bbox_height'], d['orig_shape'], d['bbox_info']), 120)
        try:                                     # one regressor stage alone (state kernel, collapsed affine map, SMPL, tail, projections of stage 3)
            so = out['smpl_out'][2]
            reg = m.regressor[2]
            xc2 = torch.empty(B, reg.fc1.in_features, dtype=torch.float32, device=dev)
            ext(markers, cam=cam, out=xc2, want_point_feat=False)
            tz1 = torch.full((B,), 5.0, device=dev)
            c_, s_, bh_ = d['center'].float().contiguous(), d['scale'].float(), d['bbox_height'].float().contiguous()
            os_, bi_ = d['orig_shape'].float().contiguous(), d['bbox_info'].float().contiguous()
            stage = lambda: [reg(None, bi_, tz1, os_, c_, s_, bh_, so['rotmat'], so['pred_shape'], so['pred_cam'], is_train=False, n_iter=1, J_regressor=None,
                                 with_aux=False, xc=xc2, xc_next=None, state_ready=False) for _ in range(30)]
            scan('regressor stage x 30', stage, 60)
        except Exception as e:                   # noqa: BLE001
            print('regressor stage scan skipped:', type(e).__name__, e)
        full_x = torch.randn(1, 3, 600, 800, generator=torch.Generator().manual_seed(11)).to(dev)
        scan('camera model (ResNet-50, 600x800) x 4', lambda: [m.cam_model(full_x) for _ in range(4)], 120)
        m.overlap_camera = m.overlap_tz = False
        scan('full forward, side streams folded', lambda: m(d['x'], None, d['center'], d['scale'], d['bbox_height'], d['orig_shape'], d['bbox_info']), 140)
    sys.exit(0)
for numerics in ('bf16', 'bf16x3', 'fp32'):
    m = whmr_net(None, assets=assets, numerics=numerics)
    m.load_state_dict(sd, strict=False)
    m = m.to(dev).eval()
    with torch.no_grad():
        scan('full W-HMR forward, %s' % numerics, lambda: m(d['x'], None, d['center'], d['scale'], d['bbox_height'], d['orig_shape'], d['bbox_info']),
             {'bf16': 120, 'bf16x3': 250, 'fp32': 600}[numerics])
    del m
for numerics in ('bf16', 'fp32'):
    m = whmr_net(None, assets=assets, numerics=numerics)
    m.load_state_dict(sd, strict=False)
    m = m.to(dev).train()
    scan('training step, %s' % numerics, train_step(m), {'bf16': 500, 'fp32': 2500}[numerics])
    del m
