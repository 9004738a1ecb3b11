"""Backward pass of the ViT backbone (HIP GEMM / LayerNorm / GELU kernels) against the CPU oracle's autograd."""
import math

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
