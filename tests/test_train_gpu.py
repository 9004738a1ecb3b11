"""Backward pass of the ViT backbone (HIP GEMM / LayerNorm / GELU kernels) against the CPU oracle's autograd."""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


def test_train_helper_kernels(dev):
    from whmr_amd import _lib as L
    g = torch.Generator().manual_seed(0)
    x = torch.randn(333, 200, generator=g)
    for src_dt in (torch.float32, torch.bfloat16):
        for dst_dt in (torch.float32, torch.bfloat16):
            xs = x.to(src_dt)
            t = L.transpose_cast(xs.to(dev), dst_dt)
            assert t.shape == (200, 384) and not t[:, 333:].any()
            assert torch.equal(t[:, :333].cpu(), xs.float().t().to(dst_dt))
    view = x.to(dev)[:, 8:72]                                                       # strided rows
    assert torch.equal(L.transpose_cast(view, torch.float32, pad_to=1).cpu(), x[:, 8:72].t())
    out = torch.empty(200, device=dev)
    assert _rel(L.colsum(x.to(dev), out).cpu(), x.sum(0)) < 2e-6
    L.colsum(x.to(dev), out, accumulate=True)
    assert _rel(out.cpu(), 2 * x.sum(0)) < 2e-6
    # transpose fused with the column sums (bias gradient from the dY transpose), fast path and fallback
    for shape in ((328, 200), (333, 200)):
        xb = torch.randn(*shape, generator=g).bfloat16()
        cs = torch.ones(shape[1], device=dev)
        tt = L.transpose_colsum(xb.to(dev), cs, pad_to=64, accumulate=True)
        assert tt.shape == (shape[1], 384) and torch.equal(tt[:, :shape[0]].cpu(), xb.t()) and not tt[:, shape[0]:].any()
        assert _rel(cs.cpu() - 1, xb.float().sum(0)) < 1e-5
    # LayerNorm backward vs autograd
    C = 768
    xx = torch.randn(50, C, generator=g, requires_grad=True)
    gam, bet = torch.randn(C, generator=g, requires_grad=True), torch.randn(C, generator=g, requires_grad=True)
    dy, dres = torch.randn(50, C, generator=g), torch.randn(50, C, generator=g)
    torch.nn.functional.layer_norm(xx, (C,), gam, bet, 1e-6).backward(dy)
    dx = dres.clone().to(dev)
    dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
    L.layernorm_bwd(xx.detach().to(dev), dy.to(dev), gam.detach().to(dev), dx, dx, dg, db, 1e-6)
    assert _rel(dx.cpu(), xx.grad + dres) < 1e-5 and _rel(dg.cpu(), gam.grad) < 1e-5 and _rel(db.cpu(), bet.grad) < 1e-5
    # GELU forward / backward (exact erf)
    pre = torch.randn(1000, generator=g, requires_grad=True)
    dh = torch.randn(1000, generator=g)
    torch.nn.functional.gelu(pre).backward(dh)
    o = torch.empty(1000, device=dev)
    assert _rel(L.gelu_fwd(pre.detach().to(dev), o).cpu(), torch.nn.functional.gelu(pre.detach())) < 1e-6
    dp = torch.empty(1000, device=dev)
    assert _rel(L.gelu_bwd(pre.detach().to(dev), dh.to(dev), dp).cpu(), pre.grad) < 1e-5
    pb = (torch.randn(4096 + 8, generator=g) * 2).bfloat16()                           # 8-per-thread bf16 path and the scalar tail path
    for n in (4096, 4099):
        ob = torch.empty(n, dtype=torch.bfloat16, device=dev)
        L.gelu_fwd(pb[:n].contiguous().to(dev), ob)
        assert _rel(ob.float().cpu(), torch.nn.functional.gelu(pb[:n].float())) < 4e-3          # one bf16 rounding step


@pytest.mark.parametrize('numerics,tol', [('fp32', 2e-4), ('bf16', 4e-2)])
def test_vit_backward_matches_oracle_autograd(dev, numerics, tol):
    """d(loss)/d(every parameter) of a depth-2 ViT-B at 64x48, B=3, against torch autograd through the CPU oracle."""
    from oracle import synth
    from oracle.vit import vit_forward
    from whmr_amd.models.pose_vit import ViT
    size = (64, 48)
    sd = synth.make_vit_state(3, size, depth=2)
    x = synth.make_inputs(3, 9, size)['x']
    G = torch.randn(3, 768, 4, 3, generator=torch.Generator().manual_seed(4))
    ref_sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    (vit_forward(ref_sd, x, depth=2) * G).sum().backward()
    m = ViT(img_size=size, depth=2, qkv_bias=True, numerics=numerics)
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).train()
    out = m(x.to(dev))
    assert out.requires_grad and out.shape == (3, 768, 4, 3)
    (out * G.to(dev)).sum().backward()
    worst = {}
    for name, p in m.named_parameters():
        assert p.grad is not None and p.grad.shape == p.shape, name
        worst[name] = _rel(p.grad.cpu(), ref_sd[name].grad)
    bad = {k: v for k, v in worst.items() if not v < tol}
    assert not bad, 'gradient mismatch: %s' % sorted(bad.items(), key=lambda kv: -kv[1])[:6]
    # the forward output of the training path equals the inference path up to the GELU form
    m.eval()
    with torch.no_grad():
        assert _rel(m(x.to(dev)), out.detach()) < (1e-5 if numerics == 'fp32' else 2e-2)


def test_default_numerics_model_trains_in_fp32(dev):
    """a DEFAULT-constructed ViT / WHMR is numerics='bf16x3', an inference mode that trains in fp32.  Its training step must
    not depend on what the last eval forward left in the module (``_eff`` used to pick a bf16 weight copy against fp32 activations): run an eval
    forward on the split-bf16 path first (>= 320 tokens), then a training step, and compare every gradient with the numerics='fp32' model --
    same kernels, same operands: bit-identical."""
    from oracle import synth
    from whmr_amd.models.pose_vit import ViT
    size = (256, 192)
    sd = synth.make_vit_state(3, size, depth=2)
    x = synth.make_inputs(2, 9, size)['x'].to(dev)
    G = torch.randn(2, 768, 16, 12, generator=torch.Generator().manual_seed(4)).to(dev)
    grads = {}
    for numerics in ('bf16x3', 'fp32'):
        m = ViT(img_size=size, depth=2, qkv_bias=True) if numerics == 'bf16x3' else ViT(img_size=size, depth=2, qkv_bias=True, numerics='fp32')
        assert m.numerics == numerics
        m.load_state_dict(sd, strict=True)
        m = m.to(dev).eval()
        with torch.no_grad():
            m(x)                                   # 384 tokens: the bf16x3 model takes its split-bf16 kernels here
        m.train()
        out = m(x)
        assert out.requires_grad
        (out * G).sum().backward()
        grads[numerics] = {k: p.grad.clone() for k, p in m.named_parameters()}
        assert all(torch.isfinite(g).all() for g in grads[numerics].values())
    for k, g in grads['fp32'].items():
        assert torch.equal(grads['bf16x3'][k], g), k


@pytest.mark.parametrize('N', [196, 192, 100])
def test_attention_backward_kernel(dev, N):
    """MFMA attention backward vs autograd of the fp32 attention on the same bf16 inputs."""
    from whmr_amd import _lib as L
    B, H, d = 2, 3, 64
    g = torch.Generator().manual_seed(N)
    qkv = (torch.randn(B, N, 3, H, d, generator=g) * 1.5).bfloat16()
    dout = torch.randn(B, N, H * d, generator=g)
    scale = d ** -0.5
    ref_in = qkv.float().requires_grad_(True)
    q, k, v = ref_in.permute(2, 0, 3, 1, 4)
    o_ref = (torch.softmax((q * scale) @ k.transpose(-1, -2), -1) @ v).transpose(1, 2).reshape(B, N, H * d)
    o_ref.backward(dout)
    qd = qkv.view(B * N, 3 * H * d).to(dev)
    out = torch.empty(B * N, H * d, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(B * H * N, device=dev)
    L.attention_fwd_train(qd, out, lse, B, N, H, d, scale)
    assert _rel(out.float().cpu().view(B, N, -1), o_ref.detach()) < 2e-2
    lse_ref = torch.logsumexp((q * scale) @ k.transpose(-1, -2), -1).detach() / math.log(2.0)          # [B, H, N], log2 domain
    assert (lse.cpu().view(B, H, N) - lse_ref).abs().max() < 2e-2
    dqkv = torch.empty_like(qd)
    L.attention_bwd(qd, out, dout.view(B * N, -1).contiguous().to(dev), lse, dqkv, B, N, H, d, scale)
    got = dqkv.float().cpu().view(B, N, 3, H, d)
    for i, name in enumerate('qkv'):
        assert _rel(got[:, :, i], ref_in.grad[:, :, i]) < 3e-2, name


def test_vit_backward_224_hip_attention(dev):
    """Same gradient check at the bench shape (224x224 -> 196 tokens), depth 1: the path with the MFMA attention backward kernel."""
    from oracle import synth
    from oracle.vit import vit_forward
    from whmr_amd.models.pose_vit import ViT
    from whmr_amd.train.vit_autograd import _hip_attention_bwd
    size = (224, 224)
    sd = synth.make_vit_state(5, size, depth=1)
    x = synth.make_inputs(2, 11, size)['x']
    G = torch.randn(2, 768, 14, 14, generator=torch.Generator().manual_seed(6))
    ref_sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    (vit_forward(ref_sd, x, depth=1) * G).sum().backward()
    m = ViT(img_size=size, depth=1, qkv_bias=True, numerics='bf16')
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).train()
    assert _hip_attention_bwd(m, 196)
    (m(x.to(dev)) * G.to(dev)).sum().backward()
    bad = {n: _rel(p.grad.cpu(), ref_sd[n].grad) for n, p in m.named_parameters()}
    bad = {k: v for k, v in bad.items() if not v < 4e-2}
    assert not bad, 'gradient mismatch: %s' % sorted(bad.items(), key=lambda kv: -kv[1])[:6]


def test_vit_backward_grouped_weight_gradients_match_single_launches(dev):
    """the four weight gradients of a layer in ONE grouped launch (whmr_gemm_tn_bf16_group, two K slices, four-wave body) against the four single
    launches (vit_autograd.GROUP_DW off): the same products with another slice count -- equal to fp32 rounding, biases included, and every
    other gradient of the backbone bit-identical (nothing else may change); 256x192 crops at batch 64 = the training step's token count"""
    from oracle import synth
    from whmr_amd.models.pose_vit import ViT
    from whmr_amd.train import vit_autograd as VA
    size = (256, 192)
    sd = synth.make_vit_state(7, size, depth=2)
    x = synth.make_inputs(64, 12, size)['x'].to(dev)
    G = torch.randn(64, 768, 16, 12, generator=torch.Generator().manual_seed(8)).to(dev)
    grads = {}
    old = VA.GROUP_DW
    try:
        for grouped in (False, True):
            VA.GROUP_DW = grouped
            m = ViT(img_size=size, depth=2, qkv_bias=True, numerics='bf16', drop_path_rate=0.0)
            m.load_state_dict(sd, strict=True)
            m = m.to(dev).train()
            (m(x) * G).sum().backward()
            grads[grouped] = {n: p.grad.clone() for n, p in m.named_parameters()}
    finally:
        VA.GROUP_DW = old
    lin = ('attn.qkv.', 'attn.proj.', 'mlp.fc1.', 'mlp.fc2.')
    for n, g1 in grads[True].items():
        g0 = grads[False][n]
        assert torch.isfinite(g1).all()
        if 'blocks.' in n and any(t in n for t in lin):
            assert _rel(g1, g0) < 2e-6, (n, _rel(g1, g0))
        else:
            assert torch.equal(g1, g0), n


def test_vit_large_backward_256x192(dev):
    """ViT-L/16 geometry (dim 1024, 16 heads, 192 tokens), depth 1: gradients of the HIP backward vs the CPU oracle's autograd."""
    from oracle import synth
    from oracle.vit import vit_forward
    from whmr_amd.models.pose_vit import ViT
    size = (256, 192)
    sd = synth.make_vit_state(7, size, embed_dim=1024, depth=1)
    x = synth.make_inputs(2, 13, size)['x']
    G = torch.randn(2, 1024, 16, 12, generator=torch.Generator().manual_seed(8))
    ref_sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    (vit_forward(ref_sd, x, num_heads=16, depth=1) * G).sum().backward()
    m = ViT(img_size=size, embed_dim=1024, depth=1, num_heads=16, qkv_bias=True, numerics='bf16')
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).train()
    (m(x.to(dev)) * G.to(dev)).sum().backward()
    bad = {n: _rel(p.grad.cpu(), ref_sd[n].grad) for n, p in m.named_parameters()}
    bad = {k: v for k, v in bad.items() if not v < 4e-2}
    assert not bad, 'gradient mismatch: %s' % sorted(bad.items(), key=lambda kv: -kv[1])[:6]


def _rms(a, b):
    return ((a.double() - b.double()).pow(2).mean().sqrt() / b.double().pow(2).mean().sqrt().clamp_min(1e-30)).item()


@pytest.mark.parametrize('numerics,tol', [('fp32', 2e-4), ('bf16', 1e-1)])
def test_deconv_bn_relu_train_matches_autograd(dev, numerics, tol):
    """Two chained deconv stages (768 -> 256 -> 256) in train mode: outputs, running stats and every gradient vs CPU autograd.
    fp32: max-abs error / max-abs reference <= 2e-4.  bf16: the maps z are stored in bf16, so ~0.25 % of the ReLU gates sit within
    one rounding step of zero and open / close differently from the fp32 reference; with the white-noise cotangent of this test
    every flipped gate moves a gradient sum by O(1), which bounds the agreement at ~5-7 % RMS (tools/probes/deconv_bwd_diag.py:
    the same figure at every size, also for the pure column sum dbeta) -- gated as RMS error / RMS reference <= 1e-1."""
    _rel = globals()['_rel'] if numerics == 'fp32' else _rms
    from oracle.train import deconv_bn_relu_train
    from whmr_amd.train.deconv_autograd import DeconvBNReLUFn
    dt = torch.float32 if numerics == 'fp32' else torch.bfloat16
    g = torch.Generator().manual_seed(11)
    B, H, W = 3, 4, 3
    x = torch.randn(B, 768, H, W, generator=g)
    ws = [torch.randn(768, 256, 4, 4, generator=g) * 0.02, torch.randn(256, 256, 4, 4, generator=g) * 0.03]
    gam = [torch.rand(256, generator=g) + 0.5 for _ in range(2)]
    bet = [torch.randn(256, generator=g) * 0.2 for _ in range(2)]
    dy = torch.randn(B, 256, 4 * H, 4 * W, generator=g)
    # reference
    rx = x.clone().requires_grad_(True)
    rp = [[t.clone().requires_grad_(True) for t in (ws[i], gam[i], bet[i])] for i in range(2)]
    rstats = [[torch.zeros(256), torch.ones(256)] for _ in range(2)]
    h = rx
    for i in range(2):
        h = deconv_bn_relu_train(h, rp[i][0], rp[i][1], rp[i][2], rstats[i][0], rstats[i][1])
    h.backward(dy)
    # HIP
    bns = []
    for i in range(2):
        bn = torch.nn.BatchNorm2d(256, momentum=0.1).to(dev)
        bns.append(bn)
    dp = [[t.clone().to(dev).requires_grad_(True) for t in (ws[i], gam[i], bet[i])] for i in range(2)]
    dx_in = x.permute(0, 2, 3, 1).contiguous().to(dev).to(dt).requires_grad_(True)
    hh = dx_in
    for i in range(2):
        hh = DeconvBNReLUFn.apply(hh, dp[i][0], dp[i][1], dp[i][2], bns[i], dt)
    assert hh.shape == (B, 4 * H, 4 * W, 256) and hh.dtype == dt
    hh.backward(dy.permute(0, 2, 3, 1).contiguous().to(dev).to(dt))
    assert _rel(hh.detach().float().cpu().permute(0, 3, 1, 2), h.detach()) < tol
    for i in range(2):
        assert _rel(bns[i].running_mean.cpu(), rstats[i][0]) < tol and _rel(bns[i].running_var.cpu(), rstats[i][1]) < tol
        for a, b, name in zip(dp[i], rp[i], ('weight', 'gamma', 'beta')):
            assert a.grad is not None and a.grad.shape == b.grad.shape
            assert _rel(a.grad.float().cpu(), b.grad) < tol, (i, name, _rel(a.grad.float().cpu(), b.grad))
    assert _rel(dx_in.grad.float().cpu().permute(0, 3, 1, 2), rx.grad) < tol


@pytest.mark.parametrize('numerics', ['fp32', 'bf16'])
def test_deconv_gradient_error_is_relu_gate_flips(dev, numerics):
    """the end-to-end training tests gate the deconv / ViT gradients at 1e-2 RMS "because of ReLU gate flips" -- this test shows
    that the explanation holds.  Two chained deconv stages at a size where a few hundred pre-activations sit within rounding of zero, dense
    white-noise cotangent: the HIP gradients against a float64 CPU evaluation (i) as is, and (ii) with the float64 run's ReLU gates REPLACED by
    the gates the HIP forward took (mask = its output > 0).  With the same gates the two agree to the arithmetic's own resolution (fp32: < 2e-5
    RMS; bf16 operands: < 1.5e-2); the gap between (i) and (ii) is the flipped gates alone, and their count is reported."""
    import torch.nn.functional as F
    from whmr_amd.train.deconv_autograd import DeconvBNReLUFn
    dt = torch.float32 if numerics == 'fp32' else torch.bfloat16
    g = torch.Generator().manual_seed(11)
    B, H, W = 8, 16, 12
    x = torch.randn(B, 768, H, W, generator=g)
    ws = [torch.randn(768, 256, 4, 4, generator=g) * 0.02, torch.randn(256, 256, 4, 4, generator=g) * 0.03]
    gam = [torch.rand(256, generator=g) + 0.5 for _ in range(2)]
    bet = [torch.randn(256, generator=g) * 0.2 for _ in range(2)]
    dy = torch.randn(B, 256, 4 * H, 4 * W, generator=g)
    # HIP: keep both stage outputs (their > 0 pattern = the gates the device used)
    bns = [torch.nn.BatchNorm2d(256, momentum=0.1).to(dev) for _ in range(2)]
    dp = [[t.clone().to(dev).requires_grad_(True) for t in (ws[i], gam[i], bet[i])] for i in range(2)]
    xin = x.permute(0, 2, 3, 1).contiguous().to(dev).to(dt).requires_grad_(True)
    ys, hh = [], xin
    for i in range(2):
        hh = DeconvBNReLUFn.apply(hh, dp[i][0], dp[i][1], dp[i][2], bns[i], dt)
        ys.append(hh)
    hh.backward(dy.permute(0, 2, 3, 1).contiguous().to(dev).to(dt))
    hip = [xin.grad.float().cpu().permute(0, 3, 1, 2)] + [t.grad.float().cpu() for r in dp for t in r]
    gates = [(y.detach().float().cpu().permute(0, 3, 1, 2) > 0) for y in ys]

    def f64_chain(masks):
        rx = x.double().requires_grad_(True)
        rp = [[t.double().requires_grad_(True) for t in (ws[i], gam[i], bet[i])] for i in range(2)]
        h, own = rx, []
        for i in range(2):
            z = F.conv_transpose2d(h, rp[i][0], None, stride=2, padding=1)
            z = F.batch_norm(z, None, None, rp[i][1], rp[i][2], training=True, eps=1e-5)
            own.append(z.detach() > 0)
            h = F.relu(z) if masks is None else z * masks[i].double()
        h.backward(dy.double())
        return [rx.grad] + [t.grad for r in rp for t in r], own
    ref, own = f64_chain(None)
    ref_m, _ = f64_chain(gates)
    flips = [int((a != b).sum()) for a, b in zip(gates, own)]
    names = ['dx', 'w0', 'g0', 'b0', 'w1', 'g1', 'b1']
    plain = {n: _rms(a, b) for n, a, b in zip(names, hip, ref)}
    same = {n: _rms(a, b) for n, a, b in zip(names, hip, ref_m)}
    print('%s deconv chain: %d + %d of %d + %d gates differ from float64; RMS error vs float64 %s; with the device gates injected %s'
          % (numerics, flips[0], flips[1], gates[0].numel(), gates[1].numel(), {k: '%.1e' % v for k, v in plain.items()}, {k: '%.1e' % v for k, v in same.items()}))
    tight = 2e-5 if numerics == 'fp32' else 1.5e-2
    assert all(v < tight for v in same.values()), same
    if sum(flips) == 0:
        pytest.skip('no gate flipped at this size in this numerics: nothing to attribute')
    # the stage-1 quantities sit behind BOTH ReLUs: without the injected gates they carry the flip error, far above the same-gates figure
    assert max(plain['dx'], plain['w0']) > 5 * max(same['dx'], same['w0']), (plain, same)


def test_im2col_t_and_bn_kernels(dev):
    """whmr_im2col_t against F.unfold; BN statistics with a large common offset (the shifted two-stage sum must not cancel)."""
    from whmr_amd import _lib as L
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 64, 6, 10, generator=g)                                        # NCHW
    for dt in (torch.float32, torch.bfloat16):
        xs = x.to(dt)
        t = L.im2col_t(xs.permute(0, 2, 3, 1).contiguous().to(dev), 3, 5, 4, 4, 2, 1, pad_to=64)
        M = 2 * 3 * 5
        assert t.shape == (16 * 64, 64) and not t[:, M:].any()
        u = torch.nn.functional.unfold(xs.float(), 4, padding=1, stride=2)            # [B, C*16, L], k = (c, ky, kx)
        u = u.view(2, 64, 16, 15).permute(2, 1, 0, 3).reshape(16 * 64, M)             # [(ky,kx,c), (b, oy, ox)]
        assert torch.equal(t[:, :M].float().cpu(), u)
    z = torch.randn(5000, 256, generator=g) * 0.01 + 100.0
    st = L.bn_stats(z.to(dev), torch.ones(256, device=dev), torch.zeros(256, device=dev), 1e-5).cpu()
    assert _rel(st[0], z.double().mean(0)) < 1e-6
    assert _rel(st[1], 1.0 / torch.sqrt(z.double().var(0, unbiased=False) + 1e-5)) < 1e-4


@pytest.mark.parametrize('B', [1, 5])
def test_smpl_backward_matches_oracle_autograd(dev, assets, B):
    """d(betas), d(rotmats) of the SMPL forward for random cotangents on vertices, 49 joints, 45 SMPL joints and markers, against torch
    autograd through the CPU restatement (oracle/smpl.py).  Raw (non-orthonormal) 3x3 blocks: the training path has no Gram-Schmidt."""
    from oracle import geometry as OG
    from oracle import smpl as OS
    from whmr_amd.models.smpl import SMPL
    from whmr_amd.train.smpl_autograd import SMPLFn
    g = torch.Generator().manual_seed(40 + B)
    betas = torch.randn(B, 10, generator=g)
    rot = OG.batch_rodrigues(torch.randn(B * 24, 3, generator=g) * 0.7).view(B, 24, 3, 3) + 0.05 * torch.randn(B, 24, 3, 3, generator=g)
    cv, cj = torch.randn(B, 6890, 3, generator=g), torch.randn(B, 49, 3, generator=g) * 30
    cs, cm = torch.randn(B, 45, 3, generator=g) * 30, torch.randn(B, len(assets['ssm']), 3, generator=g) * 10
    rb, rr = betas.clone().requires_grad_(True), rot.clone().requires_grad_(True)
    v, j = OS.smpl_forward(rb, rr, assets['smpl'])
    sj = OS.vertex_joint_selector(v, torch.einsum('bik,ji->bjk', v, assets['smpl']['J_regressor']))
    mk = v[:, assets['ssm']]
    ((v * cv).sum() + (j * cj).sum() + (sj * cs).sum() + (mk * cm).sum()).backward()
    m = SMPL(arrays=assets['smpl'], marker_ids=assets['ssm']).to(dev)
    db, dr = betas.clone().to(dev).requires_grad_(True), rot.clone().to(dev).requires_grad_(True)
    hv, hj, hs, hm = SMPLFn.apply(db, dr, m)
    assert _rel(hv.detach().cpu(), v.detach()) < 1e-5 and _rel(hj.detach().cpu(), j.detach()) < 1e-5
    assert _rel(hs.detach().cpu(), sj.detach()) < 1e-5 and _rel(hm.detach().cpu(), mk.detach()) < 1e-6
    ((hv * cv.to(dev)).sum() + (hj * cj.to(dev)).sum() + (hs * cs.to(dev)).sum() + (hm * cm.to(dev)).sum()).backward()
    assert _rel(db.grad.cpu(), rb.grad) < 1e-4, _rel(db.grad.cpu(), rb.grad)
    assert _rel(dr.grad.cpu(), rr.grad) < 1e-4, _rel(dr.grad.cpu(), rr.grad)
    # vertices-only cotangent (no joint gradients at all)
    db2, dr2 = betas.clone().to(dev).requires_grad_(True), rot.clone().to(dev).requires_grad_(True)
    rb.grad = rr.grad = None
    (OS.smpl_forward(rb, rr, assets['smpl'])[0] * cv).sum().backward()
    (SMPLFn.apply(db2, dr2, m)[0] * cv.to(dev)).sum().backward()
    assert _rel(db2.grad.cpu(), rb.grad) < 1e-4 and _rel(dr2.grad.cpu(), rr.grad) < 1e-4


@pytest.mark.parametrize('map_dtype', [torch.float32, torch.bfloat16])
def test_maf_sampler_backward_matches_autograd(dev, map_dtype):
    """d(feature map) and d(Conv1d weights / biases) of the sampler against torch autograd through grid_sample + the point MLP
    (oracle/whmr.py restatement of maf_extractor.py:75-124), for 2-D points (incl. out-of-range ones) and projected 3-D points."""
    import torch.nn.functional as F
    from whmr_amd.models.maf_extractor import MAF_Extractor
    from whmr_amd.train.maf_autograd import MAFSampleFn
    g = torch.Generator().manual_seed(3)
    B, H, W, P = 3, 10, 7, 21
    ext = MAF_Extractor()
    for p_ in ext.parameters():
        p_.data = torch.randn(p_.shape, generator=g) * (0.1 if p_.dim() > 1 else 0.05)
    fmap = torch.randn(B, 256, H, W, generator=g).to(map_dtype).float()
    pts2 = torch.rand(B, P, 2, generator=g) * 2.4 - 1.2
    pts3 = torch.randn(B, P, 3, generator=g) * 0.4
    cam = torch.cat([torch.rand(B, 1, generator=g) * 0.5 + 0.7, torch.randn(B, 2, generator=g) * 0.1], 1)
    cot = torch.randn(B, 32 * P, generator=g)

    def ref(pts):
        fm = fmap.clone().requires_grad_(True)
        ps = [p_.detach().clone().requires_grad_(True) for p_ in ext.parameters()]
        pf = F.grid_sample(fm, pts.unsqueeze(2), align_corners=True)[..., 0]            # maf_extractor.py:118
        y = pf
        for i in range(3):                                                              # maf_extractor.py:84-99
            y = F.conv1d(y, ps[2 * i], ps[2 * i + 1])
            y = F.leaky_relu(y) if i < 2 else F.relu(y)
            if i < 2:
                y = torch.cat([y, pf], 1)
        y = y.reshape(B, -1)
        y.backward(cot)
        return y.detach(), fm.grad, [p_.grad for p_ in ps]

    ext_d = MAF_Extractor().to(dev)
    ext_d.load_state_dict(ext.state_dict())
    for mode in ('2d', '3d'):
        if mode == '2d':
            y_ref, gmap_ref, gp_ref = ref(pts2)
        else:
            tz = 2 * 1000.0 / (256.0 * cam[:, 0] + 1e-9)                              # utils/geometry.py:289-307 with the cfg res 256
            z = pts3[..., 2] + tz[:, None]
            p2 = torch.stack([1000.0 * (pts3[..., 0] + cam[:, 1:2]) / z / 128.0, 1000.0 * (pts3[..., 1] + cam[:, 2:3]) / z / 128.0], -1)
            y_ref, gmap_ref, gp_ref = ref(p2)
        fm_d = fmap.permute(0, 2, 3, 1).contiguous().to(dev).to(map_dtype).permute(0, 3, 1, 2).requires_grad_(True)
        ps_d = [p_.detach().clone().requires_grad_(True) for p_ in ext_d.parameters()]
        args = (pts2.to(dev), None, None) if mode == '2d' else (None, pts3.to(dev), cam.to(dev))
        y = MAFSampleFn.apply(fm_d, *ps_d, ext_d, *args)
        # the weights actually used are ext_d's (same values as ps_d)
        assert _rel(y.detach().cpu(), y_ref) < 1e-5
        y.backward(cot.to(dev))
        # a bf16 map gets a bf16 gradient map, accumulated in bf16 (2^-9 relative per add; this 10 x 7 map makes 21 points share texels
        # far more often than the 32 x 24 ... 128 x 96 maps of the model do)
        map_tol = 1e-4 if map_dtype == torch.float32 else 1e-2
        assert _rel(fm_d.grad.float().cpu(), gmap_ref) < map_tol, (mode, _rel(fm_d.grad.float().cpu(), gmap_ref))
        for a, b_ in zip(ps_d, gp_ref):
            assert a.grad.shape == b_.shape and _rel(a.grad.cpu(), b_) < 1e-4, (mode, a.shape, _rel(a.grad.cpu(), b_))


@pytest.mark.parametrize('side', [False, True])
@pytest.mark.parametrize('map_dtype', [torch.float32, torch.bfloat16])
def test_map_fork_adds_the_sampler_records_to_the_other_consumers_gradient(dev, map_dtype, side):
    """MapForkFn (last feature map of the training graph): the sampler leaves per-point records instead of a dense gradient map and the fork's
    backward adds them, in place, to the gradient of the map's other consumer -- here a weighted sum, evaluated on a side stream when ``side`` -- so
    that the map's gradient equals plain autograd fan-out (sampler's dense scatter + the other gradient) up to the bf16 accumulation order."""
    from whmr_amd.models.maf_extractor import MAF_Extractor
    from whmr_amd.train.maf_autograd import MAFSampleFn, MapForkFn
    g = torch.Generator().manual_seed(11)
    B, H, W, P = 4, 24, 18, 431
    ext = MAF_Extractor().to(dev)
    for p_ in ext.parameters():
        p_.data = (torch.randn(p_.shape, generator=g) * (0.1 if p_.dim() > 1 else 0.05)).to(dev)
    fmap = torch.randn(B, H, W, 256, generator=g).to(dev).to(map_dtype)
    other = torch.randn(B, H, W, 256, generator=g).to(dev).to(map_dtype)
    pts3 = (torch.randn(B, P, 3, generator=g) * 0.4).to(dev)
    cam = torch.cat([torch.rand(B, 1, generator=g) * 0.5 + 0.7, torch.randn(B, 2, generator=g) * 0.1], 1).to(dev)
    cot = torch.randn(B, 32 * P, generator=g).to(dev)
    ps = [p_.detach().clone().requires_grad_(True) for p_ in ext.parameters()]

    def run(fork):
        fm = fmap.clone().requires_grad_(True)
        for p_ in ps:
            p_.grad = None
        main = torch.cuda.current_stream(dev)
        st = torch.cuda.Stream(device=dev) if side else main
        st.wait_stream(main)
        sink = {} if fork else None
        with torch.cuda.stream(st):
            fm_s, fm_c = MapForkFn.apply(fm, sink) if fork else (fm, fm)
            other_loss = (fm_c.float() * other.float()).sum()
        y = MAFSampleFn.apply(fm_s.permute(0, 3, 1, 2), *ps, ext, None, pts3, cam, sink)
        main.wait_stream(st)
        (other_loss + (y * cot).sum()).backward()
        torch.cuda.synchronize(dev)
        assert not sink                                                     # the records were consumed
        return fm.grad.float().cpu(), [p_.grad.clone().cpu() for p_ in ps]

    g_plain, w_plain = run(False)
    g_fork, w_fork = run(True)
    assert g_fork.abs().max() > 0 and (g_fork - other.float().cpu()).abs().max() > 0          # both contributions are there
    # bf16 map: fan-out rounds (sum of the ~4 contributions a texel of this small map gets) + other once, the fork adds them to `other` one at a time
    # -- a few bf16 ulps at the largest values (2^-5 at |g| in 4 .. 8), 2^-9-grade on the whole map
    tol, tol_rms = (1e-6, 1e-6) if map_dtype == torch.float32 else (4e-2, 4e-3)
    assert _rel(g_fork, g_plain) < tol and _rms(g_fork, g_plain) < tol_rms, (_rel(g_fork, g_plain), _rms(g_fork, g_plain))
    for a, b_ in zip(w_fork, w_plain):
        assert torch.equal(a, b_)                                           # the MLP's gradients do not pass through the map


def _train_model(assets, state_dict, numerics, dev):
    from whmr_amd.models import whmr_net
    m = whmr_net(None, assets=assets, numerics=numerics)
    res = m.load_state_dict(state_dict, strict=False)
    assert not res.unexpected_keys and all('.smpl.' in k for k in res.missing_keys)
    m = m.to(dev).train()
    for mod in m.modules():                     # dropout masks are random draws: parity runs use p = 0 on both sides
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    # stochastic depth: the plain parity runs switch it off on both sides (rate 0 = the oracle without masks); test_*_stochastic_depth
    # below injects the SAME keep masks on both sides instead
    m.feature_extractor.backbone.drop_path_rate = 0.0
    return m


# constants added in front of the Tz head's batch-statistics BatchNorm1d: true gradient zero, both sides hold rounding noise (see the fp32 test)
ZERO_GRAD_KEYS = ('est_Tz.0.bias', 'est_Tz.1.bias', 'transformer_decoder.mlp.fc2.bias')


def _grad_keys(sd, with_global=False):
    # global_orient.* receives a gradient when the loss covers global_output (round 5: test_whmr_train_step_fp32_matches_oracle_autograd)
    skip = ('running', 'cam_model', 'smpl', 'Dmap', 'points_grid', 'init_', 'num_batches') + (() if with_global else ('global_orient',))
    return [k for k, v in sd.items() if v.is_floating_point() and not any(s in k for s in skip)]


@pytest.mark.parametrize('stage', [2, 1])
def test_whmr_train_step_fp32_matches_oracle_autograd(dev, assets, state_dict, stage):
    """WHMR.forward(is_train=True) + backward (B=2, 256x192, fp32 parity numerics): the per-stage outputs, the BatchNorm running
    statistics and d(loss)/d(every trained parameter) against torch autograd through the CPU restatement (oracle/train.py, pinned
    against the imported reference by tests/golden/make_golden_train.py), for both TRAIN.STAGE detach layouts."""
    from oracle import synth
    from oracle import train as OT
    from whmr_amd.core.cfgs import cfg
    inp = synth.make_inputs(2, 0)
    keys = _grad_keys(state_dict, with_global=True)
    p = {k: (v.clone().requires_grad_(True) if k in keys else v) for k, v in state_dict.items()}
    stats, dp_ref, g_ref = {}, [], []
    # global_output (whmr.py:630-654) with a non-trivial camera rotation: the one the reference's cam_model gave the fixture's frame
    import numpy as np
    from conftest import GOLDEN
    cam_rotmat = torch.from_numpy(np.load(os.path.join(GOLDEN, 'whmr_train_b2.npz'))['cam_rotmat'])
    outs_ref = OT.whmr_forward_train(p, assets, inp['x'], inp['center'], inp['scale'], inp['bbox_height'], inp['orig_shape'],
                                     inp['bbox_info'], stage=stage, stats=stats, dp_out=dp_ref, global_out=g_ref, cam_rotmat=cam_rotmat)
    (OT.cotangent_loss(outs_ref) + OT.dp_cotangent_loss(dp_ref[0]) + OT.global_cotangent_loss(g_ref[0])).backward()
    m = _train_model(assets, state_dict, 'fp32', dev)
    old = cfg.TRAIN.STAGE
    cfg.TRAIN.STAGE = stage
    try:
        d = {k: inp[k].to(dev) for k in ('x', 'center', 'scale', 'bbox_height', 'orig_shape', 'bbox_info')}
        out_list, vis = m(d['x'], None, d['center'], d['scale'], d['bbox_height'], d['orig_shape'], d['bbox_info'], is_train=True,
                          cam_rotmat=cam_rotmat.to(dev))
        (OT.cotangent_loss(out_list['smpl_out'], dev=dev) + OT.dp_cotangent_loss(out_list['dp_out'][0], dev=dev)
         + OT.global_cotangent_loss(out_list['global_output'], dev=dev)).backward()
    finally:
        cfg.TRAIN.STAGE = old
    assert len(out_list['smpl_out']) == 4 and len(vis) == 4 and len(out_list['dp_out']) == 1
    for k in ('global_pose', 'global_rotmat', 'global_kp_3d', 'global_verts', 'global_shape'):
        e = _rel(out_list['global_output'][k].detach().cpu(), g_ref[0][k].detach())
        assert out_list['global_output'][k].requires_grad and e < 1e-4, (k, e)
    for k, v in dp_ref[0].items():                                                     # IUV head outputs (NCHW), iuv_predictor.py:71-91
        assert out_list['dp_out'][0][k].shape == v.shape and _rel(out_list['dp_out'][0][k].detach().cpu(), v.detach()) < 1e-4, k
    for l in range(1, 4):
        for k in OT.TRAIN_LOSS_KEYS + ('theta', 'pred_cam_t', 'smpl_kp_3d', 'markers'):
            e = _rel(out_list['smpl_out'][l][k].detach().cpu(), outs_ref[l][k].detach())
            assert e < 1e-4, (l, k, e)
    for k, v in stats.items():
        assert _rel(m.state_dict()[k].cpu(), v) < 1e-4, k
    # second oracle pass with the ReLU gates of the three deconv stages replaced by the device's (its maps > 0): what is left of the deconv / backbone
    # gradient differences once no gate is decided differently (the B = 64 test below does the same at the benchmark size)
    gates = [(f.cpu() > 0) for f in vis[1:]]
    p2 = {k: (v.clone().requires_grad_(True) if k in keys else v) for k, v in state_dict.items()}
    dp2, fm_ref = [], []
    with torch.no_grad():
        OT.whmr_forward_train(state_dict, assets, inp['x'], inp['center'], inp['scale'], inp['bbox_height'], inp['orig_shape'], inp['bbox_info'], stage=stage,
                              fmaps_out=fm_ref)
    flips = [int((g_ != (f > 0)).sum()) for g_, f in zip(gates, fm_ref)]
    g2 = []
    outs_g = OT.whmr_forward_train(p2, assets, inp['x'], inp['center'], inp['scale'], inp['bbox_height'], inp['orig_shape'], inp['bbox_info'], stage=stage,
                                   dp_out=dp2, relu_gates=gates, global_out=g2, cam_rotmat=cam_rotmat)
    (OT.cotangent_loss(outs_g) + OT.dp_cotangent_loss(dp2[0]) + OT.global_cotangent_loss(g2[0])).backward()
    worst_gated = 0.0
    named = dict(m.named_parameters())
    bad = {}
    for k in keys:
        g = named[k].grad
        assert g is not None and g.shape == p[k].grad.shape, k
        ref = p[k].grad
        # A constant added in front of a batch-statistics BatchNorm has a true gradient of exactly zero: the deconv BatchNorm biases' convolutions
        # have none, but the Tz head ends in Linear -> Linear -> BatchNorm1d(1) (whmr.py:424-427), so est_Tz.{0,1}.bias and the bias of the last
        # Linear before the token mean (transformer_decoder.mlp.fc2.bias) only shift the BatchNorm input.  Both sides then hold fp32 rounding noise
        # (~1e-7, and the CPU side's noise depends on torch's thread partitioning): gate those on an absolute bound, not on a ratio of two noises.
        if ref.abs().max() < 1e-8 or k in ZERO_GRAD_KEYS:
            assert ref.abs().max().item() < 1e-5, (k, 'the oracle says this gradient is not zero')
            if g.abs().max().item() > 1e-5:
                bad[k] = ('zero-grad', g.abs().max().item())
            continue
        # Upstream of a ReLU that sees a DENSE gradient (the IUV head -- and with TRAIN.STAGE 2 the Tz head -- back-propagate into the whole
        # last feature map; with only the sparse sampler gradients the same parameters agree to ~1e-5)
        # two fp32 evaluations disagree on the few dozen gates whose pre-activation is within rounding of zero, and every flipped gate
        # moves the deconv / ViT gradients by ~1/sqrt(map size) ~ 1.6e-3 RMS -- CPU fp32 vs float64 shows the same figure
        # (tools/probes/deconv_bwd_diag.py).  Those parameters are gated on the RMS error; everything else on the max error.
        dense = k.startswith('deconv_layers') or k.startswith('feature_extractor')
        e = _rms(g.cpu(), ref) if dense else _rel(g.cpu(), ref)
        if not e < (1e-2 if dense else 1e-3):
            bad[k] = e
        if dense:                                  # the same gradient against the oracle pass that took the DEVICE's ReLU gates: fp32 resolution
            eg = _rel(g.cpu(), p2[k].grad)
            worst_gated = max(worst_gated, eg)
            if not eg < GATED_MAXREL_B2:
                bad[(k, 'same gates')] = eg
    print('train view, TRAIN.STAGE %d: %s ReLU gates differ from the CPU oracle; worst deconv / backbone gradient with the device gates injected: max-rel %.2e'
          % (stage, flips, worst_gated))
    assert not bad, 'gradient mismatch: %s' % sorted(bad.items(), key=lambda kv: str(kv[1]))[:8]
    # the global-orientation head is part of the graph (whmr.py:289-305,631): its six parameters were compared above like every other head
    assert sum(k.startswith('global_orient.') for k in keys) == 6 and all(named[k].grad is not None for k in keys if k.startswith('global_orient.'))


KP2DW_DELTA = 1e-5                        # kp_2d_w vs float64, relative to the projection's condition number (measured: see the printout)
GATED_MAXREL_B2 = 3e-5                    # the same at B = 2 over ALL 150-odd deconv / backbone gradients (measured 3.7-4.6e-6, both TRAIN.STAGE layouts)
GATED_MAXREL = 1e-4                       # B = 64 deconv / backbone gradients against the oracle run with the device's ReLU gates (measured 1.4-1.8e-5)


def test_whmr_train_step_batch64_vs_cpu_oracle(dev, assets, state_dict):
    """BASELINE configs[3] at its per-GPU batch of 64 (only B = 2 was tested): at this size other GEMM tiles, TN split-K
    slice counts and the heavy-chain side stream engage.  fp32 numerics, stochastic depth with the SAME injected keep masks on both sides:
    loss, every supervised per-stage output, the BatchNorm running statistics and ten watched gradients (heads exactly, the ReLU-gated
    deconv / backbone ones by RMS) against torch autograd through the CPU oracle -- and the same gradients against a second oracle pass that takes the
    DEVICE's ReLU gates (max-rel 1e-3), plus a float64 forward that shows the kp_2d_w outliers are conditioning.  ~1-1.5 min of CPU oracle."""
    from oracle import synth
    from oracle import train as OT
    B = 64
    inp = synth.make_inputs(B, 3)
    watch = ('regressor.0.fc1.weight', 'regressor.2.deccam.weight', 'regressor.1.decpose.bias', 'maf_extractor.2.conv0.weight',
             'est_Tz.0.weight', 'conv.1.weight', 'dp_head.predict_u.weight', 'deconv_layers.0.weight', 'deconv_layers.7.weight',
             'feature_extractor.backbone.blocks.11.mlp.fc2.weight', 'feature_extractor.backbone.blocks.0.attn.qkv.weight',
             'feature_extractor.backbone.patch_embed.proj.weight')
    assert all(k in state_dict for k in watch)
    p = {k: (v.clone().requires_grad_(True) if k in watch else v) for k, v in state_dict.items()}
    masks = torch.floor((1 - torch.linspace(0, 0.3, 12)).repeat_interleave(2).view(-1, 1) + torch.rand(24, B, generator=torch.Generator().manual_seed(5)))
    torch.set_num_threads(min(16, max(torch.get_num_threads(), 8)))
    stats, dp_ref = {}, []
    fmaps_ref = []
    outs_ref = OT.whmr_forward_train(p, assets, inp['x'], inp['center'], inp['scale'], inp['bbox_height'], inp['orig_shape'], inp['bbox_info'],
                                     stats=stats, dp_out=dp_ref, drop_masks=masks, drop_path_rate=0.3, fmaps_out=fmaps_ref)
    loss_ref = OT.cotangent_loss(outs_ref) + OT.dp_cotangent_loss(dp_ref[0])
    loss_ref.backward()
    m = _train_model(assets, state_dict, 'fp32', dev)
    vit = m.feature_extractor.backbone
    vit.drop_path_rate, vit.dpr, vit.drop_masks = 0.3, [v.item() for v in torch.linspace(0, 0.3, 12)], masks
    d = {k: inp[k].to(dev) for k in ('x', 'center', 'scale', 'bbox_height', 'orig_shape', 'bbox_info')}
    out_list, vis_dev = m(d['x'], None, d['center'], d['scale'], d['bbox_height'], d['orig_shape'], d['bbox_info'], is_train=True)
    loss = OT.cotangent_loss(out_list['smpl_out'], dev=dev) + OT.dp_cotangent_loss(out_list['dp_out'][0], dev=dev)
    loss.backward()
    assert abs(loss.item() - loss_ref.item()) < 1e-4 * max(1.0, abs(loss_ref.item())), (loss.item(), loss_ref.item())
    # ---- two more evaluations of the oracle that turn the two ARGUED gates of this test into shown ones:
    # (i) float64 (forward only): kp_2d_w = (focal (x + tx) / (z + Tz) + c) / c - 1 with the TRAINING-mode Tz head (BatchNorm1d over the 64 crops ->
    #     sigmoid x 10).  Among 64 synthetic crops some put a joint close to the camera plane (z + Tz small) and the division amplifies fp32 rounding.
    #     If that is the whole story, the device is as close to float64 as the CPU fp32 oracle is -- asserted below, per stage.
    # (ii) fp32 again with the ReLU gates of the three deconv stages REPLACED by the ones the device took (its maps > 0): the gradients behind a ReLU
    #     that sees a dense gradient (deconvs, backbone) then agree to the arithmetic's own resolution instead of ~2-4e-3 RMS.
    def dbl(t):
        if torch.is_tensor(t):
            return t.detach().double() if t.is_floating_point() else t
        if isinstance(t, dict):
            return {k_: dbl(v_) for k_, v_ in t.items()}
        return type(t)(dbl(v_) for v_ in t) if isinstance(t, (list, tuple)) else t
    torch.set_default_dtype(torch.float64)
    try:
        p64 = {k: (v.detach().double().requires_grad_(True) if k in watch else dbl(v)) for k, v in state_dict.items()}
        dp64 = []
        outs64 = OT.whmr_forward_train(p64, dbl(assets), *(dbl(inp[k]) for k in ('x', 'center', 'scale', 'bbox_height', 'orig_shape', 'bbox_info')),
                                       dp_out=dp64, drop_masks=masks.double(), drop_path_rate=0.3)
        (OT.cotangent_loss(outs64) + OT.dp_cotangent_loss(dp64[0])).backward()
        outs64 = [{k_: (v_.detach() if torch.is_tensor(v_) else v_) for k_, v_ in o.items()} for o in outs64]
    finally:
        torch.set_default_dtype(torch.float32)
    gates = [(f.cpu() > 0) for f in vis_dev[1:]]
    p2 = {k: (v.clone().requires_grad_(True) if k in watch else v) for k, v in state_dict.items()}
    dp2, fm2 = [], []
    outs_g = OT.whmr_forward_train(p2, assets, inp['x'], inp['center'], inp['scale'], inp['bbox_height'], inp['orig_shape'], inp['bbox_info'],
                                   dp_out=dp2, drop_masks=masks, drop_path_rate=0.3, relu_gates=gates)
    (OT.cotangent_loss(outs_g) + OT.dp_cotangent_loss(dp2[0])).backward()
    flips = [int((g_ != (f.detach() > 0)).sum()) for g_, f in zip(gates, fmaps_ref)]
    print('B=64 train step: ReLU gates of the 3 deconv stages that differ between the device and the CPU fp32 oracle: %s of %s'
          % (flips, [g_.numel() for g_ in gates]))
    bad = {}
    for l in range(1, 4):
        for k in OT.TRAIN_LOSS_KEYS + ('theta', 'pred_cam_t'):
            got, ref32, ref64 = out_list['smpl_out'][l][k].detach().cpu(), outs_ref[l][k].detach(), outs64[l][k]
            e = _rel(got, ref32)
            if k == 'kp_2d_w':
                # q = focal (X + t) / ((Z + Tz) c): both sums cancel when a joint sits near the camera plane / the optical axis.  An upstream relative
                # error delta of the joints / translation shows up as |dq| <= delta (amp_depth |q| + focal (|X| + |t|) / (|Z + Tz| c)), amp_depth =
                # (|Z| + |Tz|) / |Z + Tz| -- the projection's own condition number.  EVERY element of the device result (no joint excluded) sits
                # inside that bound around the float64 value with delta = KP2DW_DELTA, and so does the CPU fp32 oracle.
                o64 = outs64[l]
                J, T = o64['kp_3d'], o64['pred_cam_t'].unsqueeze(1)
                c = (inp['orig_shape'][:, [1, 0]].double() / 2.0).unsqueeze(1)
                depth = (J[..., 2:3] + T[..., 2:3]).abs().clamp_min(1e-30)
                amp = (J[..., 2:3].abs() + T[..., 2:3].abs()) / depth
                bound = amp * ref64.abs() + o64['focal_length'].view(-1, 1, 1) * (J[..., :2].abs() + T[..., :2].abs()) / (depth * c)
                r_dev = ((got.double() - ref64).abs() / bound).max().item()
                r_cpu = ((ref32.double() - ref64).abs() / bound).max().item()
                print('B=64 train step: stage %d kp_2d_w: max-rel device vs CPU fp32 %.2e; error / condition bound vs float64: device %.2e, CPU fp32 %.2e; '
                      'worst amplification %.0f' % (l, e, r_dev, r_cpu, amp.max().item()))
                if not r_dev < KP2DW_DELTA:
                    bad[(l, k, 'vs float64 / condition number')] = (r_dev, r_cpu)
                continue
            if not e < 1e-4:
                bad[(l, k)] = e
    for k, v in dp_ref[0].items():
        e = _rel(out_list['dp_out'][0][k].detach().cpu(), v.detach())
        if not e < 1e-4:
            bad[('dp', k)] = e
    for k, v in stats.items():
        e = _rel(m.state_dict()[k].cpu(), v)
        if not e < 1e-4:
            bad[('stat', k)] = e
    named = dict(m.named_parameters())
    for k in watch:
        g, ref = named[k].grad, p[k].grad
        assert g is not None and g.shape == ref.shape, k
        dense = k.startswith('deconv_layers') or k.startswith('feature_extractor')          # behind ReLU gates that see a dense gradient: see the B = 2 test
        e = _rms(g.cpu(), ref) if dense else _rel(g.cpu(), ref)
        eg_max, eg_rms = _rel(g.cpu(), p2[k].grad), _rms(g.cpu(), p2[k].grad)
        e64_dev, e64_cpu = _rel(g.cpu(), p64[k].grad), _rel(ref, p64[k].grad)
        print('B=64 train step: %-55s %s error %.2e | same ReLU gates on both sides: max-rel %.2e, rms %.2e | max-rel vs float64: device %.2e, CPU fp32 %.2e'
              % (k, 'rms' if dense else 'max-rel', e, eg_max, eg_rms, e64_dev, e64_cpu))
        if dense:
            # behind ReLUs that see a dense gradient.  As is (the reference's own gates) by RMS: each of the ~190 gates (of 264 M) that two fp32
            # evaluations decide differently moves these gradients by ~1/sqrt(map size); with the device's gates injected into the oracle they agree
            # to fp32 resolution (measured 1.4-1.8e-5 max-rel)
            if not e < 1e-2:
                bad[k] = e
            if not (eg_max < GATED_MAXREL and eg_rms < GATED_MAXREL):
                bad[(k, 'same gates')] = (eg_max, eg_rms)
        else:
            # heads, max-rel: two fp32 evaluations of one exact gradient are held to 1e-3 of each other -- or, where fp32 itself is that far from the
            # float64 value (the stage-3 sampler MLP: 1.4e-2 on BOTH sides; crops whose synthetic Tz puts a joint near the camera plane own the
            # largest and worst-conditioned Jacobian entries), to no more than their own distance from it; and the device is never an order of
            # magnitude further from float64 than the reference's fp32 arithmetic
            if not (e < 1e-3 or e <= 1.05 * max(e64_dev, e64_cpu)):
                bad[k] = (e, e64_dev, e64_cpu)
            if not e64_dev <= 10 * e64_cpu + 2e-4:
                bad[(k, 'vs float64')] = (e64_dev, e64_cpu)
    assert not bad, bad


def test_whmr_train_step_batch64_bf16_finite_and_deterministic(dev, assets, state_dict):
    """the same step in the bf16 throughput numerics at batch 64 (the bench.py --workload whmr_train configuration): finite loss and gradients,
    and two runs from the same state with the same masks give the same bits (fixed-order reductions everywhere, no atomics on fp32 sums
    except the sampler's bf16 map scatter, which is order-dependent: its consumers are compared with a tolerance)"""
    from oracle import synth
    from oracle import train as OT
    B = 64
    inp = synth.make_inputs(B, 3)
    masks = torch.floor((1 - torch.linspace(0, 0.3, 12)).repeat_interleave(2).view(-1, 1) + torch.rand(24, B, generator=torch.Generator().manual_seed(5)))
    d = {k: inp[k].to(dev) for k in ('x', 'center', 'scale', 'bbox_height', 'orig_shape', 'bbox_info')}
    runs = []
    for _ in range(2):
        m = _train_model(assets, state_dict, 'bf16', dev)
        vit = m.feature_extractor.backbone
        vit.drop_path_rate, vit.dpr, vit.drop_masks = 0.3, [v.item() for v in torch.linspace(0, 0.3, 12)], masks
        out_list, _ = m(d['x'], None, d['center'], d['scale'], d['bbox_height'], d['orig_shape'], d['bbox_info'], is_train=True)
        loss = OT.cotangent_loss(out_list['smpl_out'], dev=dev) + OT.dp_cotangent_loss(out_list['dp_out'][0], dev=dev)
        loss.backward()
        grads = {k: v.grad.detach().clone() for k, v in m.named_parameters() if v.grad is not None}
        assert torch.isfinite(loss) and all(torch.isfinite(g).all() for g in grads.values())
        runs.append((loss.item(), out_list['smpl_out'][3]['verts'].detach().clone(), grads))
    assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1])          # the forward is bit-reproducible
    exact = [k for k in runs[0][2] if k.startswith('regressor') or k.startswith('est_Tz') or k.startswith('dp_head')]
    assert len(exact) > 30
    for k in exact:                                                                    # heads: no order-dependent accumulation upstream of them
        assert torch.equal(runs[0][2][k], runs[1][2][k]), k
    for k, g in runs[0][2].items():                                                    # behind the sampler's bf16 scatter: equal up to its accumulation order
        assert _rms(runs[1][2][k].cpu(), g.cpu()) < 2e-2, k


@pytest.mark.parametrize('numerics,tol', [('fp32', 1e-4), ('bf16', 3e-2)])
def test_conv_linear_downsample_nodes(dev, assets, numerics, tol):
    """ConvNHWCFn (both Tz-head convolutions, whmr.py:419-420), LinearFn and DownsampleFn against torch autograd on the CPU."""
    import torch.nn.functional as F
    from whmr_amd.train.heads_autograd import ConvNHWCFn, DownsampleFn, LinearFn
    dt = torch.float32 if numerics == 'fp32' else torch.bfloat16
    g = torch.Generator().manual_seed(21)
    x = torch.randn(2, 256, 31, 25, generator=g)
    w0, w1 = torch.randn(64, 256, 7, 7, generator=g) * 0.01, torch.randn(5, 64, 7, 7, generator=g) * 0.02
    rx, r0, r1 = (t.clone().requires_grad_(True) for t in (x, w0, w1))
    y = F.conv2d(F.conv2d(rx, r0, None, stride=3), r1, None, stride=2)
    cot = torch.randn(y.shape, generator=g)
    y.backward(cot)
    dx = x.permute(0, 2, 3, 1).contiguous().to(dev).to(dt).requires_grad_(True)
    d0, d1 = w0.clone().to(dev).requires_grad_(True), w1.clone().to(dev).requires_grad_(True)
    yy = ConvNHWCFn.apply(ConvNHWCFn.apply(dx, d0, 3, dt), d1, 2, dt)
    assert yy.shape == (2, y.shape[2], y.shape[3], 5)
    assert _rel(yy.detach().float().cpu().permute(0, 3, 1, 2), y.detach()) < tol
    yy.backward(cot.permute(0, 2, 3, 1).contiguous().to(dev).to(dt))
    assert _rel(d0.grad.cpu(), r0.grad) < tol and _rel(d1.grad.cpu(), r1.grad) < tol
    assert _rel(dx.grad.float().cpu().permute(0, 3, 1, 2), rx.grad) < tol
    # 3x3 stride-1 'same' convolution with bias and 90 output channels (the IUV head as one GEMM; data gradient = flipped-kernel gather conv
    # in bf16, col2im in fp32)
    x3 = torch.randn(2, 256, 12, 9, generator=g)
    w3, b3 = torch.randn(90, 256, 3, 3, generator=g) * 0.03, torch.randn(90, generator=g)
    rx3, rw3, rb3 = (t.clone().requires_grad_(True) for t in (x3, w3, b3))
    y3 = F.conv2d(rx3, rw3, rb3, stride=1, padding=1)
    c3 = torch.randn(y3.shape, generator=g)
    y3.backward(c3)
    dx3 = x3.permute(0, 2, 3, 1).contiguous().to(dev).to(dt).requires_grad_(True)
    dw3, db3 = w3.clone().to(dev).requires_grad_(True), b3.clone().to(dev).requires_grad_(True)
    yy3 = ConvNHWCFn.apply(dx3, dw3, 1, dt, 1, db3)
    assert _rel(yy3.detach().float().cpu().permute(0, 3, 1, 2), y3.detach()) < tol
    yy3.backward(c3.permute(0, 2, 3, 1).contiguous().to(dev).to(dt))
    assert _rel(dw3.grad.cpu(), rw3.grad) < tol and _rel(db3.grad.cpu(), rb3.grad) < tol
    assert _rel(dx3.grad.float().cpu().permute(0, 3, 1, 2), rx3.grad) < tol
    if numerics == 'bf16':
        return
    a, w, b = torch.randn(7, 300, generator=g), torch.randn(33, 300, generator=g), torch.randn(33, generator=g)
    ra, rw, rb = (t.clone().requires_grad_(True) for t in (a, w, b))
    c2 = torch.randn(7, 33, generator=g)
    F.linear(ra, rw, rb).backward(c2)
    da, dw, db = (t.clone().to(dev).requires_grad_(True) for t in (a, w, b))
    LinearFn.apply(da, dw, db).backward(c2.to(dev))
    assert _rel(da.grad.cpu(), ra.grad) < 1e-5 and _rel(dw.grad.cpu(), rw.grad) < 1e-5 and _rel(db.grad.cpu(), rb.grad) < 1e-5
    v = torch.randn(3, 6890, 3, generator=g)
    rv = v.clone().requires_grad_(True)
    sub = torch.matmul(assets['Dmap0'], rv)
    tmp = torch.matmul(assets['Dmap1'], sub)
    cs, ct = torch.randn(sub.shape, generator=g), torch.randn(tmp.shape, generator=g)
    ((sub * cs).sum() + (tmp * ct).sum()).backward()
    dv = v.clone().to(dev).requires_grad_(True)
    hs, ht = DownsampleFn.apply(dv, assets['Dmap0'].float().to(dev), assets['Dmap1'].float().to(dev), {})
    assert _rel(hs.detach().cpu(), sub.detach()) < 1e-5 and _rel(ht.detach().cpu(), tmp.detach()) < 1e-5
    ((hs * cs.to(dev)).sum() + (ht * ct.to(dev)).sum()).backward()
    assert _rel(dv.grad.cpu(), rv.grad) < 1e-5


def test_whmr_train_step_bf16_error_report(dev, assets, state_dict):
    """bf16 perf numerics of the full training step (B=2): outputs within 2e-2 of the fp32 oracle and every parameter gradient within
    0.35 RMS -- a gross-error gate for the bf16-only code paths (one-launch 4-phase deconv, padded N=5 convolution, bf16 transposed
    im2col / col2im, bf16 BatchNorm maps).  The agreement is bounded by ReLU / leaky-ReLU gates that flip under the 2^-9 rounding of the
    feature maps against a white-noise cotangent at batch 2 (observed: regressor 0.5-1 %, Tz head 5 %, sampler MLP / deconv / ViT
    10-16 % RMS; tools/probes/train_step_diag.py bf16); the fp32 mode is the parity gate."""
    from oracle import synth
    from oracle import train as OT
    inp = synth.make_inputs(2, 0)
    keys = _grad_keys(state_dict)
    p = {k: (v.clone().requires_grad_(True) if k in keys else v) for k, v in state_dict.items()}
    dp_ref = []
    outs_ref = OT.whmr_forward_train(p, assets, inp['x'], inp['center'], inp['scale'], inp['bbox_height'], inp['orig_shape'], inp['bbox_info'],
                                     dp_out=dp_ref)
    (OT.cotangent_loss(outs_ref) + OT.dp_cotangent_loss(dp_ref[0])).backward()
    m = _train_model(assets, state_dict, 'bf16', dev)
    d = {k: inp[k].to(dev) for k in ('x', 'center', 'scale', 'bbox_height', 'orig_shape', 'bbox_info')}
    out_list, _ = m(d['x'], None, d['center'], d['scale'], d['bbox_height'], d['orig_shape'], d['bbox_info'], is_train=True)
    (OT.cotangent_loss(out_list['smpl_out'], dev=dev) + OT.dp_cotangent_loss(out_list['dp_out'][0], dev=dev)).backward()
    for l in range(1, 4):
        for k in ('verts', 'kp_2d', 'kp_3d', 'rotmat', 'pred_shape', 'pred_cam'):
            assert _rel(out_list['smpl_out'][l][k].detach().cpu(), outs_ref[l][k].detach()) < 2e-2, (l, k)
    named = dict(m.named_parameters())
    rep = {}
    for k in keys:
        if p[k].grad.abs().max() < 1e-8 or k in ZERO_GRAD_KEYS:
            continue
        rep[k] = _rms(named[k].grad.cpu(), p[k].grad)
    bad = {k: v for k, v in rep.items() if not v < 0.35}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:8]
    groups = {}
    for k, v in rep.items():
        groups.setdefault(k.split('.')[0], []).append(v)
    print('bf16 train-step gradient RMS error by module: ' + ', '.join('%s %.3f' % (g, max(v)) for g, v in sorted(groups.items())))


def test_regressor_forward_is_train_direct_call(dev, assets, state_dict):
    """Regressor.forward(is_train=True) called on its own (whmr.py:102-209): output dict with a graph, gradients reach its Linear layers."""
    m = _train_model(assets, state_dict, 'fp32', dev)
    reg = m.regressor[1]
    B = 3
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, 67 * 32, generator=g).to(dev).requires_grad_(True)
    bbox = torch.rand(B, 5, generator=g).to(dev)
    Tz = (torch.rand(B, generator=g) * 5 + 2).to(dev)
    orig = torch.tensor([[720., 1280.]] * B, device=dev)
    center = torch.tensor([[640., 360.]] * B, device=dev)
    out, feat = reg(x, bbox, Tz, orig, center, torch.ones(B, device=dev), torch.full((B,), 300., device=dev), is_train=True)
    assert feat.shape == (B, 67 * 32 + 5) and out['verts'].shape == (B, 6890, 3) and out['verts'].requires_grad
    (out['verts'].pow(2).mean() + out['kp_2d_w'].pow(2).mean() + out['kp_2d'].pow(2).mean()).backward()    # kp_2d carries the camera gradient
    assert x.grad is not None and reg.fc1.weight.grad is not None and reg.deccam.weight.grad is not None
    assert torch.isfinite(reg.fc1.weight.grad).all() and reg.fc1.weight.grad.abs().max() > 0


def test_train_step_hip_graph_replay_matches_eager(dev, assets, state_dict):
    """capture_train_step: the replayed whole-step HIP graph (forward + loss + backward, B=2, fp32, dropout off) reproduces the eager step's
    loss and parameter gradients, and follows new data copied into the static input tensors."""
    from oracle import synth
    from oracle import train as OT
    from whmr_amd.train import capture_train_step
    m = _train_model(assets, state_dict, 'fp32', dev)
    for name, p_ in m.named_parameters():
        if name.startswith(('cam_model', 'dp_head', 'global_orient')):
            p_.requires_grad_(False)
    params = [p_ for p_ in m.parameters() if p_.requires_grad]
    inp = synth.make_inputs(2, 0)
    d = {k: inp[k].to(dev).clone() for k in ('x', 'center', 'scale', 'bbox_height', 'orig_shape', 'bbox_info')}

    with torch.no_grad():                                   # device-resident cotangents (no host -> device copy inside the captured step)
        out0, _ = m(d['x'], None, d['center'], d['scale'], d['bbox_height'], d['orig_shape'], d['bbox_info'], is_train=True)
    g = torch.Generator().manual_seed(3)
    cots = [[(torch.randn(out0['smpl_out'][l][k].shape, generator=g) / out0['smpl_out'][l][k][0].numel() ** 0.5).to(dev)
             for k in OT.TRAIN_LOSS_KEYS] for l in range(1, 4)]

    def step():
        for p_ in params:
            p_.grad = None
        out, _ = m(d['x'], None, d['center'], d['scale'], d['bbox_height'], d['orig_shape'], d['bbox_info'], is_train=True)
        loss = sum((out['smpl_out'][l + 1][k] * cots[l][i]).sum() for l in range(3) for i, k in enumerate(OT.TRAIN_LOSS_KEYS))
        loss.backward()
        return loss
    replay, loss = capture_train_step(m, step)
    bn_state = {k: v.clone() for k, v in m.state_dict().items() if 'running' in k}

    def eager_reference():
        m.load_state_dict(bn_state, strict=False)               # the running statistics are updated in place by every step
        le = step()
        return le.item(), {id(p_): p_.grad.clone() for p_ in params}
    for seed in (0, 5):
        new = synth.make_inputs(2, seed)
        for k in d:
            d[k].copy_(new[k].to(dev))
        le, ge = eager_reference()
        m.load_state_dict(bn_state, strict=False)
        replay()
        torch.cuda.synchronize()
        assert abs(loss.item() - le) < 1e-5 * max(1.0, abs(le)), (seed, loss.item(), le)
        worst = max(_rel(p_.grad, ge[id(p_)]) for p_ in params if ge[id(p_)].abs().max() > 1e-8)
        assert worst < 1e-4, (seed, worst)


@pytest.mark.parametrize('switch', ['FORK_SAMPLER3', 'TZ_TAIL_STREAM', 'HEAVY_FIRST', 'OVERLAP_HEAVY'])
def test_training_overlap_switches_do_not_change_the_step(dev, assets, state_dict, switch, monkeypatch):
    """The round-6 scheduling switches of whmr_forward_train (deferred sampler scatter, Tz tail stream, side-stream nodes first in autograd's ready queue, the
    side stream itself) change WHEN kernels run, not what they compute: fp32 numerics, B = 2, the step with the switch flipped gives the same loss and the
    same parameter gradients (bit for bit where no atomic accumulation order is involved; 1e-6 elsewhere)."""
    from oracle import synth
    from oracle import train as OT
    from whmr_amd.train import whmr_train as WT
    inp = synth.make_inputs(2, 0)
    d = {k: inp[k].to(dev) for k in ('x', 'center', 'scale', 'bbox_height', 'orig_shape', 'bbox_info')}
    res = []
    for flip in (False, True):
        if flip:
            monkeypatch.setattr(WT, switch, not getattr(WT, switch))
        m = _train_model(assets, state_dict, 'fp32', dev)
        out_list, _ = m(d['x'], None, d['center'], d['scale'], d['bbox_height'], d['orig_shape'], d['bbox_info'], is_train=True)
        loss = OT.cotangent_loss(out_list['smpl_out'], dev=dev) + OT.dp_cotangent_loss(out_list['dp_out'][0], dev=dev)
        loss.backward()
        torch.cuda.synchronize()
        res.append((loss.item(), {k: v.grad.detach().clone() for k, v in m.named_parameters() if v.grad is not None}))
    assert res[0][0] == res[1][0]
    assert res[0][1].keys() == res[1][1].keys() and len(res[0][1]) > 200
    worst = max(_rel(res[1][1][k], g_) for k, g_ in res[0][1].items() if g_.abs().max() > 1e-8)
    assert worst < 1e-5, worst


def test_bn_and_col2im_edge_shapes(dev):
    """BatchNorm kernels at C = 64 (32 rows per block pass) with a row count that is not a multiple of anything, fp32 and bf16 maps, running
    statistics and accumulate mode; col2im with padding against F.fold."""
    import torch.nn.functional as F
    from whmr_amd import _lib as L
    g = torch.Generator().manual_seed(9)
    M, C = 1237, 64
    z = torch.randn(M, C, generator=g) * 2.0 + 0.7
    gam, bet = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    dy = torch.randn(M, C, generator=g)
    for dt, tol in ((torch.float32, 1e-5), (torch.bfloat16, 2e-2)):
        zr = z.to(dt).float().detach().clone().requires_grad_(True)
        gr, br = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
        rm, rv = torch.zeros(C), torch.ones(C)
        y = F.relu(F.batch_norm(zr, rm, rv, gr, br, True, 0.1, 1e-5))
        y.backward(dy.to(dt).float())
        zd = z.to(dt).to(dev)
        drm, drv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        st = L.bn_stats(zd, gam.to(dev), bet.to(dev), 1e-5, 0.1, drm, drv)
        assert _rel(drm.cpu(), rm) < 1e-5 and _rel(drv.cpu(), rv) < 1e-5
        yd = torch.empty_like(zd)
        L.bn_apply_relu(zd, st, yd)
        assert _rel(yd.float().cpu(), y.detach()) < tol
        dz = torch.empty_like(zd)
        dg, db = torch.ones(C, device=dev), torch.ones(C, device=dev)
        L.bn_relu_bwd(zd, dy.to(dt).to(dev), st, dz, dg, db, accumulate=True)
        assert _rel(dz.float().cpu(), zr.grad) < tol and _rel(dg.cpu() - 1, gr.grad) < 1e-4 and _rel(db.cpu() - 1, br.grad) < 1e-4
    B, Cc, IH, IW, KH, KW, S, P = 2, 64, 9, 7, 4, 4, 2, 1
    OH, OW = (IH + 2 * P - KH) // S + 1, (IW + 2 * P - KW) // S + 1
    dcol = torch.randn(B * OH * OW, KH * KW * Cc, generator=g)
    # F.fold wants [B, C*KH*KW, L] with channel-major patches: our columns are (ky, kx, c)
    cols = dcol.view(B, OH * OW, KH * KW, Cc).permute(0, 3, 2, 1).reshape(B, Cc * KH * KW, OH * OW)
    ref = F.fold(cols, (IH, IW), (KH, KW), padding=P, stride=S)                       # [B, C, IH, IW]
    dx = torch.empty(B, IH, IW, Cc, device=dev)
    L.col2im(dcol.to(dev), dx, OH, OW, KH, KW, S, P)
    assert _rel(dx.cpu().permute(0, 3, 1, 2), ref) < 1e-5


@pytest.mark.parametrize('stage', [1, 2])
def test_regressor_post_node_matches_tensor_arithmetic(dev, stage):
    """RegressorPostFn (fused forward / backward of whmr.py:142-173) against the same arithmetic written as differentiable tensor expressions
    (whmr_train._projection / _perspective_norm), with cotangents on all four outputs including cam_t and focal."""
    from whmr_amd.train.heads_autograd import RegressorPostFn
    from whmr_amd.train.whmr_train import _perspective_norm, _projection
    g = torch.Generator().manual_seed(stage)
    B, J = 5, 49
    joints = (torch.randn(B, J, 3, generator=g) * 0.4).to(dev)
    cam = torch.cat([torch.rand(B, 1, generator=g) * 0.6 + 0.6, torch.randn(B, 2, generator=g) * 0.1], 1).to(dev)
    Tz = (torch.rand(B, generator=g) * 6 + 2).to(dev)
    bh = (torch.rand(B, generator=g) * 300 + 150).to(dev)
    orig = torch.tensor([[720., 1280.]] * B, device=dev)
    center = (torch.rand(B, 2, generator=g) * torch.tensor([1280., 720.])).to(dev)
    cots = [torch.randn(B, J, 2, generator=g).to(dev), torch.randn(B, J, 2, generator=g).to(dev), torch.randn(B, 3, generator=g).to(dev),
            torch.randn(B, generator=g).to(dev) * 1e-3]

    def ref(j, c, t):
        kp = _projection(j if stage == 1 else j.detach(), c)
        focal = c[:, 0].detach() * bh * t / 2.0
        cd = c.detach()
        cam_t = torch.stack([cd[:, 1] + 2.0 * (center[:, 0] - orig[:, 1] / 2.0) / (cd[:, 0] * bh),
                             cd[:, 2] + 2.0 * (center[:, 1] - orig[:, 0] / 2.0) / (cd[:, 0] * bh), t], dim=-1)
        kw = _perspective_norm(j.detach() if stage == 1 else j, cam_t, focal, orig.flip(1) / 2.0)
        return kp, kw, cam_t, focal
    leaves_r = [t.clone().requires_grad_(True) for t in (joints, cam, Tz)]
    leaves_h = [t.clone().requires_grad_(True) for t in (joints, cam, Tz)]
    out_r = ref(*leaves_r)
    out_h = RegressorPostFn.apply(*leaves_h, bh, center, orig, stage, (1000.0, 256.0, 256.0))
    for a, b in zip(out_h, out_r):
        assert _rel(a.detach(), b.detach()) < 1e-5
    sum((o * c).sum() for o, c in zip(out_r, cots)).backward()
    sum((o * c).sum() for o, c in zip(out_h, cots)).backward()
    for a, b, name in zip(leaves_h, leaves_r, ('joints', 'cam', 'Tz')):
        assert _rel(a.grad, b.grad) < 1e-4, (name, _rel(a.grad, b.grad))


@pytest.mark.parametrize('numerics,tol', [('fp32', 2e-4), ('bf16', 4e-2)])
def test_vit_backward_stochastic_depth(dev, numerics, tol):
    """Stochastic depth of the training ViT (vit.py:132-139,233: x + drop_path(branch), per-sample keep mask / keep_prob, dpr = linspace(0, rate,
    depth)): output and every parameter gradient of a depth-3 ViT-B with injected masks against torch autograd through the oracle on the
    same masks; a dropped sample's branch contributes nothing (its tokens equal the skip path), and fresh draws keep ~keep_prob of them."""
    from oracle import synth
    from oracle.vit import vit_forward
    from whmr_amd.models.pose_vit import ViT
    size, depth, rate, B = (64, 48), 3, 0.5, 4
    sd = synth.make_vit_state(3, size, depth=depth)
    x = synth.make_inputs(B, 9, size)['x']
    G = torch.randn(B, 768, 4, 3, generator=torch.Generator().manual_seed(4))
    masks = torch.tensor([[1, 1, 1, 1], [1, 1, 1, 1], [1, 0, 1, 1], [0, 1, 1, 0], [0, 0, 1, 1], [1, 1, 0, 0]], dtype=torch.float32)   # [2*depth, B]
    ref_sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref_out = vit_forward(ref_sd, x, depth=depth, drop_masks=masks, drop_path_rate=rate)
    (ref_out * G).sum().backward()
    m = ViT(img_size=size, depth=depth, qkv_bias=True, numerics=numerics, drop_path_rate=rate)
    assert [round(v, 6) for v in m.dpr] == [0.0, 0.25, 0.5]
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).train()
    m.drop_masks = masks
    out = m(x.to(dev))
    assert _rel(out.detach().cpu(), ref_out.detach()) < (1e-4 if numerics == 'fp32' else 3e-2)
    (out * G.to(dev)).sum().backward()
    bad = {}
    for name, p in m.named_parameters():
        e = _rel(p.grad.cpu(), ref_sd[name].grad)
        if not e < tol:
            bad[name] = e
    assert not bad, 'gradient mismatch: %s' % sorted(bad.items(), key=lambda kv: -kv[1])[:6]
    # identity masks at rate 0 == eval path; at rate > 0 the masked run must differ from it
    m.eval()
    with torch.no_grad():
        assert _rel(m(x.to(dev)), out.detach()) > 1e-2
    # fresh draws: Bernoulli(keep_prob) per (branch, sample); block 0 (dpr 0) never drops
    m.train()
    m.drop_masks = None
    from whmr_amd.train.vit_autograd import vit_forward_train
    torch.manual_seed(0)
    kept = []
    for _ in range(20):
        _, saved = vit_forward_train(m, x.to(dev))
        assert saved.layers[0].rs_attn is None
        kept.append(torch.stack([(saved.layers[2].rs_attn > 0).float().mean(), (saved.layers[2].rs_mlp > 0).float().mean()]))
        assert set(saved.layers[2].rs_attn.unique().tolist()) <= {0.0, 2.0}              # 0 or 1 / keep_prob
    assert 0.3 < torch.stack(kept).mean().item() < 0.7


def test_whmr_train_step_stochastic_depth_matches_oracle(dev, assets, state_dict):
    """the whole training step with the reference's stochastic depth (ViTPose-B drop_path_rate 0.3) on the keep masks of the fixture: loss
    against the imported reference's value (tests/golden/whmr_train_b2.npz) and backbone / head gradients against the oracle's autograd"""
    import numpy as np
    import os
    from conftest import GOLDEN
    from oracle import synth
    from oracle import train as OT
    fx = np.load(os.path.join(GOLDEN, 'whmr_train_b2.npz'))
    masks = torch.from_numpy(fx['drop_masks'])
    inp = synth.make_inputs(2, 0)
    keys = _grad_keys(state_dict)
    p = {k: (v.clone().requires_grad_(True) if k in keys else v) for k, v in state_dict.items()}
    dp_ref = []
    outs_ref = OT.whmr_forward_train(p, assets, inp['x'], inp['center'], inp['scale'], inp['bbox_height'], inp['orig_shape'], inp['bbox_info'],
                                     stage=2, dp_out=dp_ref, drop_masks=masks, drop_path_rate=0.3)
    (OT.cotangent_loss(outs_ref) + OT.dp_cotangent_loss(dp_ref[0])).backward()
    m = _train_model(assets, state_dict, 'fp32', dev)
    vit = m.feature_extractor.backbone
    vit.drop_path_rate = 0.3
    vit.drop_masks = masks
    assert abs(vit.dpr[-1] - 0.3) < 1e-6 and vit.dpr[0] == 0.0
    d = {k: inp[k].to(dev) for k in ('x', 'center', 'scale', 'bbox_height', 'orig_shape', 'bbox_info')}
    out_list, _ = m(d['x'], None, d['center'], d['scale'], d['bbox_height'], d['orig_shape'], d['bbox_info'], is_train=True)
    loss = OT.cotangent_loss(out_list['smpl_out'], dev=dev)
    (loss + OT.dp_cotangent_loss(out_list['dp_out'][0], dev=dev)).backward()
    assert abs(loss.item() - float(fx['loss_stage2_droppath'])) < 1e-4 * max(1.0, abs(float(fx['loss_stage2_droppath'])))
    assert abs(loss.item() - float(fx['loss_stage2'])) > 1e-4                            # not the function without stochastic depth
    named = dict(m.named_parameters())
    for k in ('feature_extractor.backbone.blocks.5.attn.proj.weight', 'feature_extractor.backbone.blocks.11.mlp.fc2.bias',
              'feature_extractor.backbone.blocks.0.attn.qkv.weight', 'feature_extractor.backbone.pos_embed', 'regressor.2.deccam.weight'):
        assert _rms(named[k].grad.cpu(), p[k].grad) < 1e-2, k


@pytest.mark.parametrize('B,N,H,d', [(2, 196, 12, 64), (3, 5, 2, 108), (2, 192, 3, 64), (1, 70, 2, 32)])
def test_attention_backward_f32_kernel(dev, B, N, H, d):
    """whmr_attention_bwd_f32 (fp32 parity mode of the ViT, the Tz head's 5-token timm Block) against torch autograd of softmax(scale q k^T) v"""
    from whmr_amd import _lib as L
    g = torch.Generator().manual_seed(N + d)
    qkv = (torch.randn(B, N, 3, H, d, generator=g) * 1.2).requires_grad_(True)
    do = torch.randn(B, N, H * d, generator=g)
    q, k, v = qkv.permute(2, 0, 3, 1, 4)
    out = (((q * d ** -0.5) @ k.transpose(-2, -1)).softmax(-1) @ v).transpose(1, 2).reshape(B, N, H * d)
    (out * do).sum().backward()
    got = L.attention_bwd_f32(qkv.detach().reshape(B * N, 3 * H * d).to(dev), do.reshape(B * N, H * d).to(dev), B, N, H, d, d ** -0.5)
    assert _rel(got.cpu().view_as(qkv), qkv.grad) < 2e-5
    again = L.attention_bwd_f32(qkv.detach().reshape(B * N, 3 * H * d).to(dev), do.reshape(B * N, H * d).to(dev), B, N, H, d, d ** -0.5)
    assert torch.equal(got, again)                                                     # deterministic


def test_tz_block_autograd_nodes(dev):
    """LayerNormFn / GeluFn / AttentionF32Fn (HIP forward + backward) of the Tz head's timm Block against torch autograd of the same ops"""
    import torch.nn.functional as F
    from whmr_amd.train.heads_autograd import AttentionF32Fn, GeluFn, LayerNormFn
    g = torch.Generator().manual_seed(0)
    B, N, H, d = 4, 5, 2, 108
    D = H * d
    x = torch.randn(B * N, D, generator=g)
    w, b = torch.randn(D, generator=g) * 0.2 + 1, torch.randn(D, generator=g) * 0.1
    wq = torch.randn(3 * D, D, generator=g) / D ** 0.5
    G = torch.randn(B * N, D, generator=g)

    def run(xx, ww, bb, wqq, hip):
        h = LayerNormFn.apply(xx, ww, bb, 1e-5) if hip else F.layer_norm(xx, (D,), ww, bb, 1e-5)
        qkv = h @ wqq.t()
        if hip:
            a = AttentionF32Fn.apply(qkv, B, N, H, d, d ** -0.5)
            y = GeluFn.apply(a)
        else:
            q, k, v = qkv.view(B, N, 3, H, d).permute(2, 0, 3, 1, 4)
            a = (((q * d ** -0.5) @ k.transpose(-2, -1)).softmax(-1) @ v).transpose(1, 2).reshape(B * N, D)
            y = F.gelu(a)
        return y
    ref_in = [t.clone().requires_grad_(True) for t in (x, w, b, wq)]
    (run(*ref_in, hip=False) * G).sum().backward()
    dev_in = [t.clone().to(dev).requires_grad_(True) for t in (x, w, b, wq)]
    out = run(*dev_in, hip=True)
    (out * G.to(dev)).sum().backward()
    assert _rel(out.detach().cpu(), run(x, w, b, wq, hip=False)) < 1e-5
    for a, r, name in zip(dev_in, ref_in, ('x', 'ln.weight', 'ln.bias', 'qkv.weight')):
        assert _rel(a.grad.cpu(), r.grad) < 5e-5, name


def test_vit_node_publishes_gradients_to_the_reducer(dev):
    """the ViT backward hands every block's gradients to GradReducer.publish as soon as they are enqueued (buckets exchange under the backward
    of the earlier blocks at world size > 1); here always_bucket=True exercises the same path on one GPU: identical gradients, every
    backbone parameter delivered early, none through the accumulate hooks"""
    from oracle import synth
    from whmr_amd.models.pose_vit import ViT
    from whmr_amd.parallel import GradReducer
    sd = synth.make_vit_state(2, (256, 192), depth=3)
    x = synth.make_inputs(2, 3, (256, 192))['x'].to(dev)
    cot = torch.randn(2, 768, 16, 12, generator=torch.Generator().manual_seed(4)).to(dev)

    def run(with_reducer):
        m = ViT(img_size=(256, 192), depth=3, qkv_bias=True, numerics='bf16', drop_path_rate=0.0)
        m.load_state_dict(sd, strict=True)
        m = m.to(dev).train()
        red = GradReducer(m.parameters(), bucket_bytes=24 << 20, always_bucket=True).attach(m) if with_reducer else None
        for _ in range(2):                                                     # twice: buckets re-arm after finish()
            for p in m.parameters():
                p.grad = None
            (m(x) * cot).sum().backward()
            if red is not None:
                assert len(red._published) == sum(1 for _ in m.parameters())
                red.finish()
        return [p.grad.clone() for p in m.parameters()], red

    ref, _ = run(False)
    got, red = run(True)
    assert len(red.buckets) >= 3
    for a, b in zip(got, ref):
        assert torch.equal(a, b)


@pytest.mark.parametrize('bucket_bytes', [1 << 20, 64 << 20])
def test_whmr_train_step_reducer_small_buckets_with_side_stream_gradients(dev, assets, state_dict, bucket_bytes):
    """GradReducer on the full training step with the heavy chain on its side stream (whmr_train.OVERLAP_HEAVY, the default): head buckets
    then mix gradients produced on two streams.  With 1-MiB buckets every bucket closes long before the streams are joined, so the pack must
    wait on the events of EVERY producing stream (round-2 advisor finding); always_bucket runs the multi-GPU code path at world size 1
    (pack, exchange stream, unpack).  The reduced gradients must equal the plain backward's bit for bit (the sampler's order-dependent
    bf16 scatter aside: compared with a tolerance, like the determinism test)."""
    from oracle import synth
    from oracle import train as OT
    from whmr_amd.parallel import GradReducer
    from whmr_amd.train import whmr_train
    assert whmr_train.OVERLAP_HEAVY
    B = 4
    inp = synth.make_inputs(B, 3)
    d = {k: inp[k].to(dev) for k in ('x', 'center', 'scale', 'bbox_height', 'orig_shape', 'bbox_info')}
    runs = []
    for with_reducer in (False, True):
        m = _train_model(assets, state_dict, 'bf16', dev)
        m.feature_extractor.backbone.drop_path_rate = 0.0
        named = [(n, p) for n, p in m.named_parameters() if p.requires_grad and not n.startswith('cam_model')]
        red = None
        if with_reducer:
            red = GradReducer([p for _, p in named], groups=[n.startswith('feature_extractor') for n, _ in named], bucket_bytes=bucket_bytes,
                              always_bucket=True)
            red.attach(m.feature_extractor.backbone)
        for _ in range(2):                                                     # twice: buckets re-arm after finish()
            for _, p in named:
                p.grad = None
            out_list, _ = m(d['x'], None, d['center'], d['scale'], d['bbox_height'], d['orig_shape'], d['bbox_info'], is_train=True)
            loss = OT.cotangent_loss(out_list['smpl_out'], dev=dev) + OT.dp_cotangent_loss(out_list['dp_out'][0], dev=dev)
            loss.backward()
            if red is not None:
                red.finish()
        torch.cuda.synchronize()
        runs.append({n: p.grad.detach().clone() for n, p in named if p.grad is not None})
    if bucket_bytes < (8 << 20):
        assert len(red.buckets) > 20
    assert runs[0].keys() == runs[1].keys() and len(runs[0]) > 200
    exact = [k for k in runs[0] if k.startswith('regressor') or k.startswith('est_Tz') or k.startswith('dp_head')]
    assert len(exact) > 30
    for k in exact:
        assert torch.equal(runs[0][k], runs[1][k]), k
    for k, g in runs[0].items():
        assert _rms(runs[1][k].cpu(), g.cpu()) < 2e-2, k


def test_tz_compose_kernels_match_the_fp64_composition(dev):
    """whmr_tz_compose (the per-step device composition of the training graph: T = w1p . w0, 9-term gather-sum into the space-to-depth layout) against
    models.whmr.compose_tz_weights (fp64, the inference path's composition), and whmr_tz_compose_bwd as its exact transpose (<dG, compose(T)> = <compose_bwd(dG), T>)."""
    from whmr_amd import _lib as L
    from whmr_amd.models.whmr import compose_tz_weights
    g = torch.Generator().manual_seed(2)
    C = 256
    w0 = (torch.randn(64, C, 7, 7, generator=g) * 0.02).to(dev)
    w1 = (torch.randn(5, 64, 7, 7, generator=g) * 0.05).to(dev)
    w1p = w1.permute(2, 3, 0, 1).reshape(245, 64).contiguous()
    T = torch.empty(245, C * 49, device=dev)
    L.gemm(w1p, w0.reshape(64, C * 49), T, trans_w=True)
    G = L.tz_compose(T, C, torch.empty(128, 36 * C, device=dev))
    ref = compose_tz_weights(w0, w1)
    assert _rel(G.cpu(), ref.cpu()) < 1e-5, _rel(G.cpu(), ref.cpu())
    Gb = L.tz_compose(T, C, torch.empty(128, 36 * C, dtype=torch.bfloat16, device=dev))
    assert torch.equal(Gb, L.cast_bf16(G))
    dG = torch.randn(128, 36 * C, generator=g).to(dev)
    dT = L.tz_compose_bwd(dG, C, torch.empty(245, C * 49, device=dev))
    lhs, rhs = (dG.double() * G.double()).sum().item(), (dT.double() * T.double()).sum().item()
    assert abs(lhs - rhs) < 1e-6 * max(abs(lhs), 1.0), (lhs, rhs)


@pytest.mark.parametrize('geom', [(2, 128, 96), (4, 64, 48)])
def test_tz_composed_convolution_node_matches_autograd_through_the_two_convolutions(dev, geom):
    """TzComposedFn (bf16 training numerics): tokens, data gradient (with and without a gradient already on the map) and both weight gradients of the
    composed Conv2d(256, 5, k25, s6) against torch autograd through conv2d(conv2d(x, w0, s3), w1, s2) in fp64 on the same bf16-rounded map."""
    import torch.nn.functional as F
    from whmr_amd.train.heads_autograd import TzComposedFn
    B, H, W = geom
    g = torch.Generator().manual_seed(7)
    assert not TzComposedFn.fits(torch.empty(3, 64, 48, 256, dtype=torch.bfloat16, device=dev))        # 264 rows: not a multiple of the TN kernel's K step
    x = (torch.randn(B, 256, H, W, generator=g) * 0.5).bfloat16()
    w0 = torch.randn(64, 256, 7, 7, generator=g) * 0.02
    w1 = torch.randn(5, 64, 7, 7, generator=g) * 0.05
    xr, w0r, w1r = x.double().requires_grad_(True), w0.double().requires_grad_(True), w1.double().requires_grad_(True)
    yr = F.conv2d(F.conv2d(xr, w0r, stride=3), w1r, stride=2)                           # [B, 5, H2, W2]
    cot = torch.randn(yr.shape, generator=g).double()
    gx_extra = torch.randn(B, H, W, 256, generator=g) * 0.1
    yr.backward(cot)
    for pt in (False, True):
        xd = x.permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
        w0d, w1d = w0.to(dev).requires_grad_(True), w1.to(dev).requires_grad_(True)
        out = TzComposedFn.apply(xd, w0d, w1d, pt)
        t, x2 = out if pt else (out, None)
        assert t.shape == (B * 5, yr.shape[2] * yr.shape[3]) and _rel(t.view(yr.shape).cpu(), yr.detach()) < 1e-2
        loss = (t.view(yr.shape) * cot.float().to(dev)).sum() + ((x2.float() * gx_extra.to(dev)).sum() if pt else 0.0)
        loss.backward()
        want_dx = xr.grad.permute(0, 2, 3, 1) + (gx_extra.double() if pt else 0.0)
        assert _rel(xd.grad.float().cpu(), want_dx) < 2e-2, (pt, _rel(xd.grad.float().cpu(), want_dx))
        assert _rms(xd.grad.float().cpu(), want_dx) < 1e-2
        assert _rel(w0d.grad.cpu(), w0r.grad) < 1e-2 and _rms(w0d.grad.cpu(), w0r.grad) < 5e-3, (_rel(w0d.grad.cpu(), w0r.grad), _rms(w0d.grad.cpu(), w0r.grad))
        assert _rel(w1d.grad.cpu(), w1r.grad) < 1e-2 and _rms(w1d.grad.cpu(), w1r.grad) < 5e-3, (_rel(w1d.grad.cpu(), w1r.grad), _rms(w1d.grad.cpu(), w1r.grad))


def test_smpl_skin_backward_is_not_disturbed_by_the_side_streams_kernels(dev, assets):
    """Round 6 regression guard (profiles/r06_coresidency_probe.txt): ``smpl_skin_bwd_kernel`` gave different bits when the 64-row tile of the gathering TN
    kernel (then the Tz head's weight gradient) ran beside it on another stream.  That tile is gone; here the SMPL skinning backward runs on the current
    stream while the Tz head's and the IUV head's convolution nodes (forward + backward, batch 64 geometry) run on a side stream, and every launch must
    give the bits of the launch that ran alone."""
    from whmr_amd import _lib as L
    from whmr_amd.models import whmr_net
    from whmr_amd.train.heads_autograd import ConvNHWCFn
    smpl = whmr_net(None, assets=assets, numerics='bf16').to(dev).regressor[0].smpl
    m = smpl._model()
    g = torch.Generator().manual_seed(0)
    B = 64
    f32 = dict(dtype=torch.float32, device=dev)
    betas = (torch.randn(B, 10, generator=g) * 0.5).to(dev)
    A = torch.randn(B, 24, 12, generator=g).to(dev)
    pose_off = (torch.randn(B, 6890 * 3, generator=g) * 0.01).to(dev)
    dv = torch.randn(B, 6890, 3, generator=g).to(dev)
    dregd = torch.randn(B, 33, 3, generator=g).to(dev)

    def skin():
        dvp, dA = torch.empty(B, 6890 * 3, **f32), torch.empty(B, 54, 288, **f32)
        L.smpl_skin_bwd(m, betas, A, pose_off, dv, dregd, dvp, dA)
        return dvp, dA
    ref = skin()
    torch.cuda.synchronize()
    x = (torch.randn(B, 128, 96, 256, generator=g) * 0.5).to(dev).bfloat16()
    w_tz = (torch.randn(64, 256, 7, 7, generator=g) * 0.02).to(dev).requires_grad_(True)
    w_iuv = (torch.randn(90, 256, 3, 3, generator=g) * 0.02).to(dev).requires_grad_(True)
    b_iuv = torch.zeros(90, device=dev, requires_grad=True)
    side = L.side_stream(dev, 0)
    bad = 0
    for r in range(6):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            xx = x.clone().requires_grad_(True)
            y = ConvNHWCFn.apply(xx, w_tz, 3, torch.bfloat16) if r % 2 == 0 else ConvNHWCFn.apply(xx, w_iuv, 1, torch.bfloat16, 1, b_iuv)
            y.backward(torch.ones_like(y))
        outs = [skin() for _ in range(12)]
        torch.cuda.synchronize()
        bad += sum(int(not (torch.equal(o[0], ref[0]) and torch.equal(o[1], ref[1]))) for o in outs)
    assert bad == 0, '%d of 72 launches beside the side stream differ from the launch that ran alone' % bad


def test_side_stream_pool_is_shared(dev):
    """_lib.side_stream: four streams per device, created once; the forward's camera / Tz streams and the training graph's heavy / tail streams are those"""
    from whmr_amd import _lib as L
    from whmr_amd.models.whmr import WHMR
    from whmr_amd.train import whmr_train as WT
    pool = [L.side_stream(dev, i) for i in range(4)]
    assert len({s_.cuda_stream for s_ in pool}) == 4 and all(L.side_stream(dev, i) is pool[i] for i in range(4))
    assert WHMR._camera_stream(dev) is pool[1] and WHMR._camera_stream(dev, 'tz') is pool[0] and WT._heavy_stream(dev) is pool[0]
    assert L.ClockProbe(dev).side is pool[3]


@pytest.mark.parametrize('case', ['tz conv0 7x7 s3', 'tz conv1 7x7 s2'])
def test_strided_convolution_data_gradient_grouped_launch_matches_single_launches(dev, case, monkeypatch):
    """ConvNHWCFn.backward, stride S > 1: the S*S residue-class GEMMs as ONE grouped launch (whmr_gemm_bf16_group, tiles 192 x 256 / 128 x 64) give the
    bits of the S*S single launches -- same tiles' K order, disjoint output pixels -- with and without a gradient already on the map."""
    from whmr_amd.train import heads_autograd as HA
    g = torch.Generator().manual_seed(5)
    Cout, Cin, S, H, W = (64, 256, 3, 40, 31) if 'conv0' in case else (5, 64, 2, 42, 31)
    w = (torch.randn(Cout, Cin, 7, 7, generator=g) * 0.02).to(dev)
    x0 = (torch.randn(3, H, W, Cin, generator=g) * 0.5).to(dev).bfloat16()
    gx = torch.randn(3, H, W, Cin, generator=g).to(dev)
    res = {}
    for grouped in (True, False):
        monkeypatch.setattr(HA, 'GROUP_DX', grouped)
        for pt in (False, True):
            x = x0.clone().requires_grad_(True)
            out = HA.ConvNHWCFn.apply(x, w, S, torch.bfloat16, 0, None, pt)
            y, x2 = out if pt else (out, None)
            gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(11)).to(dev)
            loss = (y.float() * gy).sum() + ((x2.float() * gx).sum() if pt else 0.0)
            loss.backward()
            res[(grouped, pt)] = x.grad.clone()
    for pt in (False, True):
        assert res[(True, pt)].abs().max() > 0
        assert torch.equal(res[(True, pt)], res[(False, pt)]), (case, pt, (res[(True, pt)].float() - res[(False, pt)].float()).abs().max().item())


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float32])
def test_passthrough_nodes_accumulate_the_data_gradient_in_place(dev, dt):
    """ConvNHWCFn / DeconvBNReLUFn with ``passthrough``: the input map is handed on as a second output, and the gradient that comes back on it
    receives the node's own data gradient in the GEMM epilogue (row-major: residual == C; strided convolution: the scattered in-place
    accumulate, epi_flags bit 2; fp32 strided: the separate-tensor fallback).  Same input gradient as the two-consumer graph autograd sums."""
    from whmr_amd.train.deconv_autograd import DeconvBNReLUFn
    from whmr_amd.train.heads_autograd import ConvNHWCFn
    g = torch.Generator().manual_seed(3)
    B, Cin = 2, 256
    tol = 2.0 ** -6 if dt == torch.bfloat16 else 1e-5
    bn = torch.nn.BatchNorm2d(256).to(dev).train()
    ctw = (torch.randn(Cin, 256, 4, 4, generator=g) * 0.03).to(dev)
    cases = [('tz 7x7 s3', lambda x, pt: ConvNHWCFn.apply(x, cw7, 3, dt, 0, None, pt), (31, 25)),
             ('iuv 3x3 same', lambda x, pt: ConvNHWCFn.apply(x, cw3, 1, dt, 1, cb3, pt), (16, 12)),
             ('deconv', lambda x, pt: DeconvBNReLUFn.apply(x, ctw, bn.weight, bn.bias, bn, dt, pt), (8, 6))]
    cw7 = (torch.randn(64, Cin, 7, 7, generator=g) * 0.02).to(dev)
    cw3 = (torch.randn(90, Cin, 3, 3, generator=g) * 0.02).to(dev)
    cb3 = torch.zeros(90, device=dev)
    for name, fn, (H, W) in cases:
        x0 = (torch.randn(B, H, W, Cin, generator=g) * 0.5).to(dev).to(dt)
        gx = torch.randn(B, H, W, Cin, generator=g).to(dev)
        grads = []
        for pt in (True, False):
            x = x0.clone().requires_grad_(True)
            out = fn(x, pt)
            y, x2 = out if pt else (out, x)
            assert not pt or (x2.data_ptr() == x.data_ptr() and x2.shape == x.shape)       # the same map, no copy
            gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(11)).to(dev)
            ((y.float() * gy).sum() + (x2.float() * gx).sum()).backward()
            grads.append(x.grad.float())
        a, b = grads
        assert (a - b).abs().max() <= tol * b.abs().max(), (name, ((a - b).abs().max() / b.abs().max()).item())
        # the handed-on map alone (its first output unused): the gradient passes through untouched
        x = x0.clone().requires_grad_(True)
        (fn(x, True)[1].float() * gx).sum().backward()
        assert torch.equal(x.grad.float(), gx.to(dt).float()), name


def test_weight_operands_one_launch_matches_the_per_weight_casts(dev):
    """whmr_weights_prepare (L.WeightOperands): W and W^T bf16 copies of a list of fp32 matrices -- ragged sizes, a 4-D conv weight viewed as
    [N, K] -- are the bits of whmr_cast_bf16 / whmr_transpose_cast; a second refresh without a version change launches nothing, an in-place
    update re-makes the copies in the SAME buffers."""
    from whmr_amd import _lib as L
    g = torch.Generator().manual_seed(2)
    ps = [torch.randn(*sh, generator=g).to(dev) for sh in ((768, 3, 16, 16), (2304, 768), (70, 130), (64, 64), (1, 5), (3072, 768))]
    wo = L.WeightOperands(ps, {id(ps[0]): (768, 768)})
    ops = wo.refresh()
    for p in ps:
        w2 = p.reshape(p.shape[0], -1)
        w, wt = ops[id(p)]
        assert torch.equal(w, L.cast_bf16(w2)) and torch.equal(wt, L.transpose_cast(w2, torch.bfloat16, pad_to=1))
    ptrs = [(a.data_ptr(), b.data_ptr()) for a, b in ops.values()]
    ops[id(ps[2])][0].zero_()
    assert wo.refresh() is ops and not ops[id(ps[2])][0].any()                          # unchanged versions: nothing re-made
    with torch.no_grad():
        ps[2].mul_(2.0)
    ops = wo.refresh()
    assert torch.equal(ops[id(ps[2])][0], L.cast_bf16(ps[2])) and ptrs == [(a.data_ptr(), b.data_ptr()) for a, b in ops.values()]
    assert not wo.stale()
    ps[1].data = ps[1].data.clone()
    assert wo.stale()


def test_grad_reducer_exchange_waits_for_the_pack_on_its_own_stream(dev, monkeypatch):
    """a bucket packed on stream A may be launched later from stream B (it waited for the bucket before it).  The exchange
    stream must order itself behind the PACK (an event at the tail of stream A), not behind stream B.  Fake collective = flat *= 2 on the
    exchange stream; stream A is held back by a long sleep kernel, so an exchange that only waits for stream B would double garbage and be
    overwritten by the late copy (gradients x 1 instead of x 2)."""
    import torch.distributed as dist
    from whmr_amd.parallel import GradReducer

    class _Work:
        def wait(self):
            return True

    def fake_all_reduce(t, op=None, group=None, async_op=False):
        t.mul_(2.0)
        return _Work()
    monkeypatch.setattr(dist, 'is_initialized', lambda: True)
    monkeypatch.setattr(dist, 'get_world_size', lambda group=None: 1)
    monkeypatch.setattr(dist, 'all_reduce', fake_all_reduce)
    a, b = torch.nn.Linear(256, 256).to(dev), torch.nn.Linear(256, 256).to(dev)
    params = list(a.parameters()) + list(b.parameters())
    red = GradReducer(params, bucket_bytes=1 << 30, always_bucket=True, groups=[0, 0, 1, 1])
    assert len(red.buckets) == 2
    sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    grads = [torch.randn_like(p) for p in params]
    torch.cuda.synchronize(dev)
    with torch.cuda.stream(sa):
        torch.cuda._sleep(200_000_000)                     # ~0.1 s: stream A is far behind the host
        ga = [g.clone() for g in grads[:2]]                # the gradients of bucket 1 are PRODUCED on stream A, behind the sleep
        for p, g in zip(params[:2], ga):
            red.publish(p, g)                              # packs bucket 1 on stream A; its launch waits for bucket 0
    assert red.buckets[1]['flat'] is not None and red.buckets[1]['work'] is None
    with torch.cuda.stream(sb):
        for p, g in zip(params[2:], grads[2:]):
            red.publish(p, g)                              # bucket 0 packs and launches; bucket 1 launches from HERE (stream B)
    assert red.buckets[1]['work'] is not None
    red.finish()
    torch.cuda.synchronize(dev)
    for p, g in zip(params, grads):
        assert torch.equal(p.grad, 2.0 * g)
    red.remove()


@pytest.mark.parametrize('dtname', ['fp32', 'bf16'])
def test_sync_batchnorm_split_kernels(dev, dtname):
    """The SyncBatchNorm halves of the BatchNorm kernels (include/whmr_hip.h: whmr_bn_sums / _stats_from_sums / _bwd_sums / _bwd_apply):
    (a) run back to back with nothing exchanged they equal the unsplit entries (whmr_bn_stats / whmr_bn_relu_bwd) to fp32 rounding;
    (b) the sums of two HALF batches, added like the all-reduce adds them, give the statistics, running statistics and data gradients of the
        whole batch, and the halves' local dgamma / dbeta add up to the whole batch's (two ranks of B == one process of 2B)."""
    from whmr_amd import _lib as L
    dt = torch.float32 if dtname == 'fp32' else torch.bfloat16
    g = torch.Generator().manual_seed(5)
    M, Cc = 2 * 37 * 24, 256
    z = (torch.randn(M, Cc, generator=g) * 1.7 + torch.linspace(-4, 4, Cc)).to(dt).to(dev)
    dy = torch.randn(M, Cc, generator=g).to(dt).to(dev)
    gamma, beta = (torch.rand(Cc, generator=g) + 0.5).to(dev), torch.randn(Cc, generator=g).to(dev)
    rm0, rv0 = torch.randn(Cc, generator=g).to(dev), (torch.rand(Cc, generator=g) + 0.5).to(dev)
    rel = lambda a, b: ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()
    # unsplit
    rm_u, rv_u = rm0.clone(), rv0.clone()
    st_u = L.bn_stats(z, gamma, beta, 1e-5, 0.1, rm_u, rv_u)
    dz_u, dg_u, db_u = torch.empty_like(z), torch.empty(Cc, device=dev), torch.empty(Cc, device=dev)
    L.bn_relu_bwd(z, dy, st_u, dz_u, dg_u, db_u)
    # (a) split, one "rank"
    rm_s, rv_s = rm0.clone(), rv0.clone()
    sums = L.bn_sums(z)
    assert sums.dtype == torch.float64 and sums[-1].item() == M
    zd = z.double()
    assert rel(sums[:Cc], zd.sum(0)) < 1e-6 and rel(sums[Cc:2 * Cc], (zd * zd).sum(0)) < 1e-6          # fp32 row-chunk partials, double from there
    st_s = L.bn_stats_from_sums(sums, gamma, beta, 1e-5, 0.1, rm_s, rv_s)
    assert rel(st_s, st_u) < 2e-6 and rel(rm_s, rm_u) < 1e-6 and rel(rv_s, rv_u) < 2e-6
    dz_s, dg_s, db_s = torch.empty_like(z), torch.empty(Cc, device=dev), torch.empty(Cc, device=dev)
    bs = L.bn_bwd_sums(z, dy, st_u, dg_s, db_s)
    L.bn_bwd_apply(z, dy, st_u, bs, sums[-1:], dz_s)
    assert torch.equal(dg_s, dg_u) and torch.equal(db_s, db_u)
    assert rel(dz_s.float(), dz_u.float()) < (1e-6 if dt == torch.float32 else 1e-2)
    # (b) two half batches
    h = M // 2
    parts = [(z[:h].contiguous(), dy[:h].contiguous()), (z[h:].contiguous(), dy[h:].contiguous())]
    tot = sum(L.bn_sums(zp) for zp, _ in parts)                    # the all-reduce
    assert tot[-1].item() == M and rel(tot, sums) < 1e-6
    rm_h, rv_h = rm0.clone(), rv0.clone()
    st_h = L.bn_stats_from_sums(tot, gamma, beta, 1e-5, 0.1, rm_h, rv_h)
    assert rel(st_h, st_u) < 2e-6 and rel(rm_h, rm_u) < 1e-6 and rel(rv_h, rv_u) < 2e-6
    loc = []
    for zp, dyp in parts:
        dgp, dbp = torch.empty(Cc, device=dev), torch.empty(Cc, device=dev)
        loc.append((L.bn_bwd_sums(zp, dyp, st_h, dgp, dbp), dgp, dbp))
    btot = loc[0][0] + loc[1][0]                                   # the all-reduce
    assert rel(loc[0][1] + loc[1][1], dg_u) < 1e-5 and rel(loc[0][2] + loc[1][2], db_u) < 1e-5
    dz_h = torch.cat([L.bn_bwd_apply(zp, dyp, st_h, btot, tot[-1:], torch.empty_like(zp)) for zp, dyp in parts])
    assert rel(dz_h.float(), dz_u.float()) < (2e-5 if dt == torch.float32 else 1e-2)
    # and the whole thing against float64 BatchNorm autograd on the CPU (fp32 maps only: the bf16 maps round z itself)
    if dt == torch.float32:
        zr = z.cpu().double().requires_grad_(True)
        y = torch.relu(torch.nn.functional.batch_norm(zr, None, None, gamma.cpu().double(), beta.cpu().double(), True, 0.1, 1e-5))
        y.backward(dy.cpu().double())
        assert rel(dz_h.cpu(), zr.grad) < 1e-4


def test_whmr_train_step_with_converted_sync_batchnorm_world1(dev, assets, state_dict):
    """convert_sync_batchnorm on the W-HMR model (core/trainer.py:83) at world size 1 with the split kernels FORCED (always=True: the exchange is
    a no-op): the training step -- per-stage outputs, BatchNorm running statistics, loss and every gradient -- equals the local-BatchNorm step
    (the four trained BatchNorm layers take the SyncBatchNorm path: 3 x BatchNorm2d of the deconv pyramid, BatchNorm1d of the Tz head)."""
    from oracle import synth
    from oracle import train as OT
    from whmr_amd.parallel import convert_sync_batchnorm, revert_sync_batchnorm
    inp = synth.make_inputs(2, 0)
    d = {k: inp[k].to(dev) for k in ('x', 'center', 'scale', 'bbox_height', 'orig_shape', 'bbox_info')}
    res = {}
    for mode in ('local', 'sync'):
        m = _train_model(assets, state_dict, 'fp32', dev)
        if mode == 'sync':
            convert_sync_batchnorm(m, always=True)
            assert m.whmr_sync_layers >= 4
        out_list, _ = m(d['x'], None, d['center'], d['scale'], d['bbox_height'], d['orig_shape'], d['bbox_info'], is_train=True)
        loss = OT.cotangent_loss(out_list['smpl_out'], dev=dev) + OT.dp_cotangent_loss(out_list['dp_out'][0], dev=dev)
        loss.backward()
        res[mode] = (loss.item(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None},
                     {k: v.clone() for k, v in m.state_dict().items() if 'running_' in k and 'cam_model' not in k},
                     out_list['smpl_out'][3]['verts'].detach().clone())
        if mode == 'sync':
            revert_sync_batchnorm(m)
            assert not any(hasattr(q, 'whmr_sync') for q in m.modules())
    assert abs(res['local'][0] - res['sync'][0]) < 1e-5 * max(1.0, abs(res['local'][0]))
    assert _rel(res['sync'][3], res['local'][3]) < 1e-5
    for k, v in res['local'][2].items():
        assert _rel(res['sync'][2][k], v) < 1e-5, k
    assert res['local'][1].keys() == res['sync'][1].keys()
    bad = {}
    for k, gl in res['local'][1].items():
        if gl.abs().max() < 1e-7 or k in ZERO_GRAD_KEYS:
            continue
        e = _rel(res['sync'][1][k], gl)
        if not e < 2e-4:
            bad[k] = e
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:6]


_DDP_WRAP_CHILD = r'''
import json, os, sys, warnings
sys.path.insert(0, %(root)r)
import torch
import torch.distributed as dist
import torch.nn as nn
from torch.nn.parallel import DistributedDataParallel
from oracle import synth
from oracle import train as OT
from whmr_amd.models import whmr_net
from whmr_amd.parallel import sync_bn

torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
dist.init_process_group('nccl', device_id=dev)            # "nccl" IS RCCL on ROCm; one rank is all a 1-GPU box has
assets = synth.make_assets(0)
sd = synth.make_state_dict(0, assets)
inp = synth.make_inputs(2, 0)
d = {k: inp[k].to(dev) for k in ('x', 'center', 'scale', 'bbox_height', 'orig_shape', 'bbox_info')}


def model():
    m = whmr_net(None, assets=assets, numerics='fp32')
    m.load_state_dict(sd, strict=False)
    m = m.to(dev).train()
    for mod in m.modules():
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
    m.feature_extractor.backbone.drop_path_rate = 0.0
    for n, p in m.named_parameters():
        if n.startswith('cam_model'):
            p.requires_grad_(False)
    return m


def step(call, m):
    out, _ = call(d['x'], None, d['center'], d['scale'], d['bbox_height'], d['orig_shape'], d['bbox_info'], is_train=True)
    loss = OT.cotangent_loss(out['smpl_out'], dev=dev) + OT.dp_cotangent_loss(out['dp_out'][0], dev=dev)
    loss.backward()
    torch.cuda.synchronize()
    return (loss.item(), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None},
            {k: v.detach().clone() for k, v in m.state_dict().items() if 'running_' in k and 'cam_model' not in k})


plain = model()
ref = step(plain, plain)
# ---- the reference's own two lines, core/trainer.py:83-86
sync_bn.ALWAYS_SPLIT = True       # one rank: run the split kernels and the RCCL all-reduces anyway (they are what a multi-rank job executes)
m = model()
with warnings.catch_warnings(record=True) as caught:
    warnings.simplefilter('always')
    m = nn.SyncBatchNorm.convert_sync_batchnorm(m)
    wrapped = DistributedDataParallel(m, device_ids=[0], find_unused_parameters=True)
    got = step(wrapped, m)
    for p in m.parameters():
        p.grad = None
    got2 = step(wrapped, m)          # a second step: DDP has rebuilt its buckets, the unused-parameter set is known
groups = sync_bn.auto_sync_groups()
n_sync = sum(isinstance(q, nn.SyncBatchNorm) for q in m.modules())
rel = lambda a, b: ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()
res = {'loss': [ref[0], got[0], got2[0]], 'n_sync_modules': n_sync, 'groups': len(groups), 'collectives': sum(g.collectives for g in groups),
       'only_plain': sorted(set(ref[1]) - set(got[1])), 'only_wrapped': sorted(set(got[1]) - set(ref[1])),
       'only_wrapped_max': max([got[1][k].abs().max().item() for k in set(got[1]) - set(ref[1])] or [0.0]),
       'grad_err': {k: rel(got[1][k], v) for k, v in ref[1].items() if k in got[1] and v.abs().max() > 1e-7},
       'stat_err': {k: rel(got[2][k], v) for k, v in ref[2].items()},
       'stream_warnings': [str(w.message)[:200] for w in caught if 'stream' in str(w.message).lower()],
       'rccl': '.'.join(str(v) for v in torch.cuda.nccl.version())}
print('RESULT ' + json.dumps(res), flush=True)
dist.destroy_process_group()
'''


def test_reference_syncbn_ddp_wrap_world1_rccl(dev, tmp_path):
    """The reference's OWN two lines on this module (core/trainer.py:83-86, VERDICT r5 missing #2): ``model = nn.SyncBatchNorm.convert_sync_batchnorm(model)``
    then ``DistributedDataParallel(model, device_ids=[gpu], find_unused_parameters=True)``, one ``is_train=True`` step under a 1-rank RCCL group (fresh
    child under torch.distributed.run: this process has initialised the GPU and must not exec).  The torch nn.SyncBatchNorm modules are honoured by
    parallel/sync_bn.py::sync_of (the four trained layers take the split kernels + a packed fp64 all-reduce each way -- ``ALWAYS_SPLIT`` makes that
    run at world 1), DDP's reducer receives every gradient the HIP autograd nodes return: loss, gradients and BatchNorm running statistics equal the
    unwrapped step's, twice in a row."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'ddp_wrap_child.py'
    script.write_text(_DDP_WRAP_CHILD % {'root': root})
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
                          '--master-port', '29533', str(script)], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith('RESULT ')][-1][7:])
    print('reference wrap at world 1 (RCCL %s): %d nn.SyncBatchNorm modules, %d BatchNorm all-reduces over 2 steps, worst gradient max-rel %.2e'
          % (res['rccl'], res['n_sync_modules'], res['collectives'], max(res['grad_err'].values())))
    assert res['n_sync_modules'] >= 4 and res['groups'] == 1
    assert res['collectives'] == 2 * 8, res['collectives']          # 4 trained layers x (forward + backward) per step, two steps
    # every gradient of the plain step exists under the wrap; what DDP adds are the parameters no loss reaches (global_orient.* without a loss on
    # global_output -- the reason for find_unused_parameters, core/trainer.py:86): DDP hands them all-zero gradients where plain autograd leaves None
    assert not res['only_plain'] and all(k.startswith('global_orient.') for k in res['only_wrapped']) and res['only_wrapped_max'] == 0.0, \
        (res['only_plain'], res['only_wrapped'], res['only_wrapped_max'])
    assert abs(res['loss'][0] - res['loss'][1]) < 1e-5 * max(1.0, abs(res['loss'][0])), res['loss']
    bad = {k: e for k, e in res['grad_err'].items() if not e < 2e-4 and k not in ZERO_GRAD_KEYS}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:6]
    assert all(e < 1e-5 for e in res['stat_err'].values()), res['stat_err']
    assert not res['stream_warnings'], res['stream_warnings']
    assert 'stream does not match' not in out.stderr, out.stderr[-1500:]
