"""Backward pass of the ViT backbone (HIP GEMM / LayerNorm / GELU kernels) against the CPU oracle's autograd."""
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
