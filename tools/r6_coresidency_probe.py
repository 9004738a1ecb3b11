"""smpl_skin_bwd on one stream while large kernels run on another: does its output change?  (round 6: the stage-3 pose gradients of the training step
differed between runs once the side stream's backward overlapped the stage-3 SMPL backward.)   usage: python tools/r6_coresidency_probe.py [rounds]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import synth
from whmr_amd import _lib as L
from whmr_amd.models import whmr_net
from whmr_amd.train.heads_autograd import ConvNHWCFn

dev = torch.device('cuda:0')
assets = synth.make_assets(0)
net = whmr_net(None, assets=assets, numerics='bf16').to(dev)
smpl = net.regressor[0].smpl
m = smpl._model()
g = torch.Generator().manual_seed(0)
B = 64
f32 = dict(dtype=torch.float32, device=dev)
betas = (torch.randn(B, 10, generator=g) * 0.5).to(dev)
A = torch.randn(B, 24, 12, generator=g).to(dev)
pose_off = (torch.randn(B, 6890 * 3, generator=g) * 0.01).to(dev)
dv = torch.randn(B, 6890, 3, generator=g).to(dev)
if os.environ.get('PROBE_ZERO') == 'pose':
    pose_off.zero_()
if os.environ.get('PROBE_ZERO') == 'betas':
    betas.zero_()
if os.environ.get('PROBE_ONEHOT') is not None:      # only beta[l0] is non-zero: the shape sum of component c then reads shapedirs row c * 10 + l0 alone
    betas.zero_()
    betas[:, int(os.environ['PROBE_ONEHOT'])] = 0.7
if os.environ.get('PROBE_ZERO') == 'both':
    pose_off.zero_(); betas.zero_()
dregd = torch.randn(B, 33, 3, generator=g).to(dev)


def skin():
    dvp = torch.empty(B, 6890 * 3, **f32)
    dA = torch.empty(B, 54, 288, **f32)
    L.smpl_skin_bwd(m, betas, A, pose_off, dv, dregd, dvp, dA)
    return dvp, dA


ref = skin()
torch.cuda.synchronize()
side = torch.cuda.Stream()
x = (torch.randn(B, 128, 96, 256, generator=g) * 0.5).to(dev).bfloat16()
w_iuv = (torch.randn(90, 256, 3, 3, generator=g) * 0.02).to(dev).requires_grad_(True)
b_iuv = torch.zeros(90, device=dev, requires_grad=True)
w_tz = (torch.randn(64, 256, 7, 7, generator=g) * 0.02).to(dev).requires_grad_(True)


def load(kind):
    if kind == 'copy':
        for _ in range(8):
            x.clone()
        return
    want_dx, want_dw = kind in ('iuv', 'tz', 'tz_dx'), kind in ('iuv', 'tz', 'tz_dw')
    xx = x.clone().requires_grad_(want_dx)
    if kind == 'iuv':
        y = ConvNHWCFn.apply(xx, w_iuv, 1, torch.bfloat16, 1, b_iuv)
    else:
        with torch.set_grad_enabled(kind != 'tz_fwd'):
            for _ in range(4 if kind == 'tz_fwd' else 1):
                y = ConvNHWCFn.apply(xx, w_tz if want_dw else w_tz.detach(), 3, torch.bfloat16)
    if kind != 'tz_fwd':
        y.backward(torch.ones_like(y))


rep = torch.zeros(4, dtype=torch.int32, device=dev)
for kind in ('none', 'iuv', 'tz_fwd', 'tz_dw', 'tz_dx'):
    rep.zero_()
    for r in range(10):
        if kind != 'none':
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                load(kind)
        for _ in range(6):
            L._check(L.lib().whmr_debug_lds_canary(3456, 20480, 400, rep.data_ptr(), L._stream()), 'canary')
        torch.cuda.synchronize()
    r_ = rep.cpu().tolist()
    print('LDS canary (3456 x 20 KB) beside %-6s: %d dwords changed; one at dword %d: found 0x%08x expected 0x%08x' % (kind, r_[0], r_[1] - 1, r_[2] & 0xffffffff, r_[3] & 0xffffffff))
NVc = 6890
table = (torch.arange(30 * NVc, dtype=torch.int64) * 2654435761 % (1 << 32)).to(torch.int64)
table = (table - (table >= (1 << 31)).to(torch.int64) * (1 << 32)).to(torch.int32).to(dev)
for kind in ('none', 'iuv', 'tz_dw', 'tz_dx'):
    rep.zero_()
    for r in range(10):
        if kind != 'none':
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                load(kind)
        for _ in range(12):
            L._check(L.lib().whmr_debug_global_canary(table.data_ptr(), 30, NVc, 64, rep.data_ptr(), L._stream()), 'gcanary')
        torch.cuda.synchronize()
    r_ = rep.cpu().tolist()
    print('global-read canary (30 x 6890 table) beside %-6s: %d wrong reads; one at dword %d: found 0x%08x expected 0x%08x' % (kind, r_[0], r_[1] - 1, r_[2] & 0xffffffff, r_[3] & 0xffffffff))
sink_, stats_ = torch.zeros(1 << 16, **f32), torch.zeros(4, dtype=torch.int64, device=dev)
for beside in ('nothing', 'mfma ceiling kernel', '32x32x16 mfma stream', 'skin'):
    rep.zero_()
    for r in range(10):
        if beside != 'nothing':
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                if beside == 'mfma ceiling kernel':
                    L._check(L.lib().whmr_mfma_ceiling(512, 40000, sink_.data_ptr(), stats_.data_ptr(), L._stream()), 'ceiling')
                else:
                    L._check(L.lib().whmr_debug_mfma32_stream(245 if beside == 'skin' else 512, 20000, sink_.data_ptr(), L._stream()), 'mfma32')
        if beside == 'skin':                # the victim itself beside the bare 32x32x16 stream
            outs = [skin() for _ in range(12)]
            torch.cuda.synchronize()
            rep[0] += sum(int(not (torch.equal(o[0], ref[0]) and torch.equal(o[1], ref[1]))) for o in outs)
            continue
        for _ in range(8):
            L._check(L.lib().whmr_debug_pkfma_canary(3456, 2000, rep.data_ptr(), L._stream()), 'pkfma')
        torch.cuda.synchronize()
    r_ = rep.cpu().tolist()
    if beside == 'skin':
        print('smpl_skin_bwd_kernel beside the bare 32x32x16 mfma stream (245 workgroups): %d of 120 launches differ' % r_[0])
    else:
        print('v_pk_fma_f32 canary beside %s: %d lanes with a wrong LOW half, %d with a wrong HIGH half; op_sel:[0,1,0] form %d / %d (of %d)' % (beside, r_[0], r_[1], r_[2], r_[3], 10 * 8 * 3456 * 128))
import ctypes as C
# The Tz head's weight gradient called directly (no autograd) from lab builds of gemm_tn.hip that still hold the 64-row tile of the gathering kernel:
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DTN_LAB_ROW64=1 [-DTN_LAB=<n>] -I include w-hmr_amd/csrc/gemm_tn.hip -o tools/lab/build/libtn_row64[_lab<n>].so
# (TN_LAB: 1 = no MFMAs, 2 = no fragment reads, 4 = no LDS-DMA).  The library itself no longer builds that tile.
OH, OW = 42, 31
sig = [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p, C.c_long] + [C.c_int] * 12 + [C.c_void_p, C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]
here = os.path.dirname(os.path.abspath(__file__))
for name, rows in (('libtn_row64.so', 64), ('libtn_row64_lab1.so', 64), ('libtn_row64_lab2.so', 64), ('libtn_row64_lab4.so', 64), ('libtn_row64.so', 128)):
    path = os.path.join(here, 'lab', 'build', name)
    if not os.path.exists(path):
        continue
    fn = C.CDLL(path).whmr_conv_dw_tn_bf16
    fn.argtypes, fn.restype = sig, C.c_int
    dy_tz = (torch.randn(B * OH * OW, rows, generator=g) * 0.1).to(dev).bfloat16()
    dw_tz = torch.empty(rows, 49 * 256, **f32)
    bad = tot = 0
    for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ws = L.splitk_workspace(dev)
            for _ in range(3):
                rc = fn(dy_tz.data_ptr(), dy_tz.stride(0), x.data_ptr(), x.stride(2), dw_tz.data_ptr(), dw_tz.stride(0), rows, dy_tz.shape[0], B, OH, OW, 128, 96, 256,
                        7, 7, 3, 0, L.zero_page(dev).data_ptr(), 0, ws.data_ptr(), ws.numel(), None, L._stream())
                assert rc == 0, rc
        outs = [skin() for _ in range(12)]
        torch.cuda.synchronize()
        for dvp, dA in outs:
            tot += 1
            bad += int(not (torch.equal(dvp, ref[0]) and torch.equal(dA, ref[1])))
    print('Tz weight gradient (7x7 s3, %d-row tile of the gathering TN kernel, %s) on the other stream: %d of %d smpl_skin_bwd launches differ' % (rows, name, bad, tot))
    if name == 'libtn_row64.so':                       # the canaries beside THIS kernel
        rep.zero_()
        rep2 = torch.zeros(4, dtype=torch.int32, device=dev)
        for r in range(10):
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                ws = L.splitk_workspace(dev)
                for _ in range(3):
                    fn(dy_tz.data_ptr(), dy_tz.stride(0), x.data_ptr(), x.stride(2), dw_tz.data_ptr(), dw_tz.stride(0), rows, dy_tz.shape[0], B, OH, OW, 128, 96, 256,
                       7, 7, 3, 0, L.zero_page(dev).data_ptr(), 0, ws.data_ptr(), ws.numel(), None, L._stream())
            for _ in range(8):
                L._check(L.lib().whmr_debug_pkfma_canary(3456, 2000, rep.data_ptr(), L._stream()), 'pkfma')
                L._check(L.lib().whmr_debug_lds_canary(3456, 20480, 100, rep2.data_ptr(), L._stream()), 'lds')
            torch.cuda.synchronize()
        r_, r2_ = rep.cpu().tolist(), rep2.cpu().tolist()
        print('   beside the %d-row tile: v_pk_fma_f32 canary %d wrong LOW halves, %d wrong HIGH halves; with op_sel:[0,1,0] %d / %d; LDS canary %d dwords changed' % (rows, r_[0], r_[1], r_[2], r_[3], r2_[0]))
for kind in ('none', 'tz_dw', 'tz', 'iuv'):
    bad = tot = 0
    worst = 0.0
    for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
        if kind != 'none':
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                load(kind)
        outs = [skin() for _ in range(12)]
        torch.cuda.synchronize()
        for dvp, dA in outs:
            tot += 1
            if not (torch.equal(dvp, ref[0]) and torch.equal(dA, ref[1])):
                bad += 1
                if bad <= 3:
                    for nm, got, want in (('d_vposed', dvp, ref[0]), ('dA_partial', dA, ref[1])):
                        dd = torch.nonzero((got != want).flatten()).flatten()
                        if dd.numel():
                            i0 = dd[0].item()
                            print('    %s: %d of %d elements differ, flat index %d .. %d; at %d got %.6g want %.6g; distinct (b, block) pairs %s' % (
                                nm, dd.numel(), got.numel(), i0, dd[-1].item(), i0, got.flatten()[i0].item(), want.flatten()[i0].item(),
                                sorted({(int(i) // (54 * 288), (int(i) // 288) % 54) for i in dd.tolist()})[:12] if nm == 'dA_partial' else ''))
                            if nm == 'dA_partial':
                                print('      k = e %% 12 of the differing entries:', sorted({int(i) % 12 for i in dd.tolist()}), ' joints:', sorted({(int(i) % 288) // 12 for i in dd.tolist()}))
                worst = max(worst, (dA - ref[1]).abs().max().item() / ref[1].abs().max().item(), (dvp - ref[0]).abs().max().item() / ref[0].abs().max().item())
    print('load on the other stream = %-5s: %d of %d smpl_skin_bwd launches differ from the launch that ran alone (worst rel %.2e)' % (kind, bad, tot, worst))
