"""Per-kernel parity of the HIP library against plain fp32 torch on the same inputs (runs on the MI355X box)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize('M,N,K', [(392, 768, 768), (128, 128, 64), (12544, 2304, 768), (300, 3072, 768), (392, 768, 3072)])
@pytest.mark.parametrize('glds', [True])
def test_gemm_bf16(dev, M, N, K, glds):
    from whmr_amd import _lib as L
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).bfloat16()
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    ref = F.gelu(a.float() @ w.float().t() + bias) + res
    out = torch.empty(M, N, device=dev)
    L.gemm(a.to(dev), w.to(dev), out, bias=bias.to(dev), residual=res.to(dev), act=L.ACT_GELU, glds=glds)
    # the bf16-path GELU is a clamped degree-13 polynomial (|err| <= 1.9e-4 abs vs erf, below bf16 output resolution; common.h)
    assert _rel(out.cpu(), ref) < 2e-4
    # bf16 output, no epilogue extras; asymmetric operands catch transposed C writes
    out2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    L.gemm(a.to(dev), w.to(dev), out2, glds=glds)
    assert _rel(out2.float().cpu(), a.float() @ w.float().t()) < 1e-2


@pytest.mark.parametrize('M,N,K', [(392, 768, 768), (12544, 2304, 768), (300, 3072, 768), (392, 768, 3072), (1000, 130, 64), (1001, 136, 128)])
@pytest.mark.parametrize('tile', [64, 65, 128, 256, 192, 257, 259, 320, 160, 224])
def test_gemm_bf16_big_tile(dev, M, N, K, tile):
    from whmr_amd import _lib as L
    g = torch.Generator().manual_seed(M + N + K + tile)
    a = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).bfloat16()
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    ref = F.gelu(a.float() @ w.float().t() + bias) + res
    out = torch.empty(M, N, device=dev)
    L.gemm(a.to(dev), w.to(dev), out, bias=bias.to(dev), residual=res.to(dev), act=L.ACT_GELU, tile=tile)
    assert _rel(out.cpu(), ref) < 2e-4
    out1 = torch.empty(M, N, device=dev)
    L.gemm(a.to(dev), w.to(dev), out1, bias=bias.to(dev), residual=res.to(dev), tile=tile)       # exact epilogue ops only
    assert _rel(out1.cpu(), a.float() @ w.float().t() + bias + res) < 2e-5 * math.sqrt(K) / 8 + 1e-5
    out2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    L.gemm(a.to(dev), w.to(dev), out2, tile=tile)
    assert _rel(out2.float().cpu(), a.float() @ w.float().t()) < 1e-2


def test_gemm_bf16_rowmod_residual(dev):
    from whmr_amd import _lib as L
    g = torch.Generator().manual_seed(3)
    M, N, K, R = 392, 768, 768, 196
    a = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).bfloat16()
    pos = torch.randn(R, N, generator=g)
    out = torch.empty(M, N, device=dev)
    L.gemm(a.to(dev), w.to(dev), out, residual=pos.to(dev), res_row_mod=R)
    ref = a.float() @ w.float().t() + pos.repeat(M // R, 1)
    assert _rel(out.cpu(), ref) < 1e-4


@pytest.mark.parametrize('M,N,K', [(64, 1024, 2250), (2, 216, 1024), (64, 3, 1024), (392, 768, 768), (320, 648, 216), (77, 130, 33),
                                   (64, 1024, 2149), (64, 20670, 207), (5, 144, 1024), (1, 1024, 2149), (64, 2048, 2164), (33, 70, 31)])
def test_gemm_f32(dev, M, N, K):
    from whmr_amd import _lib as L
    g = torch.Generator().manual_seed(M * 7 + N + K)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    out = torch.empty(M, N, device=dev)
    L.gemm(a.to(dev), w.to(dev), out, bias=bias.to(dev), residual=res.to(dev))
    ref = (a.double() @ w.double().t() + bias + res).float()
    assert _rel(out.cpu(), ref) < 2e-6


@pytest.mark.parametrize('M,N,K', [(12544, 768, 768), (2049, 2304, 772), (1100, 3072, 40), (4100, 200, 3076), (1024, 1030, 16)])
def test_gemm_f32_big_kernel(dev, M, N, K):
    """large-M fp32 GEMM (128x128 double-buffered kernel, gemm_f32.hip): ragged M / N, K not a multiple of the 16-deep step, every epilogue option;
    against float64 and BITWISE against the 64x64 kernel (same k order of the exact-f32 MFMA chain)"""
    from whmr_amd import _lib as L
    g = torch.Generator().manual_seed(M + 3 * N + K)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    ad, wd, bd, rd = a.to(dev), w.to(dev), bias.to(dev), res.to(dev)
    out = torch.empty(M, N, device=dev)
    L.gemm(ad, wd, out, bias=bd, residual=rd, act=L.ACT_GELU)
    ref = (torch.nn.functional.gelu(a.double() @ w.double().t() + bias) + res).float()
    assert _rel(out.cpu(), ref) < 3e-6
    try:
        L.gemm_f32_set_big(0)
        small = torch.empty(M, N, device=dev)
        L.gemm(ad, wd, small, bias=bd, residual=rd, act=L.ACT_GELU)
    finally:
        L.gemm_f32_set_big(1)
    assert torch.equal(out, small)
    # positional residual (row = m % R) into a bf16 output, ReLU before the skip
    R = 7
    pos = torch.randn(R, N, generator=g).to(dev)
    o16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    L.gemm(ad, wd, o16, residual=pos, res_row_mod=R, act=L.ACT_RELU)
    ref2 = torch.relu(a.double() @ w.double().t()).float() + pos.cpu().repeat((M + R - 1) // R, 1)[:M]
    assert _rel(o16.float().cpu(), ref2) < 1e-2


def test_gemm_f32_big_kernel_conv_gather_and_scatter(dev):
    """the same kernel as an implicit GEMM: NHWC gather (7x7 s3 conv, 3x3 s1 p1 conv) and the sub-pixel deconv phase with scattered rows"""
    from whmr_amd import _lib as L
    g = torch.Generator().manual_seed(5)
    for (B, Cin, IH, IW, Cout, KH, S, P) in ((3, 64, 70, 60, 128, 7, 3, 0), (2, 32, 40, 36, 72, 3, 1, 1)):
        x = torch.randn(B, Cin, IH, IW, generator=g)
        w = torch.randn(Cout, Cin, KH, KH, generator=g) / math.sqrt(Cin * KH * KH)
        ref = F.conv2d(x.double(), w.double(), stride=S, padding=P).float()
        OH, OW = ref.shape[2:]
        assert B * OH * OW >= 1024
        xn = x.permute(0, 2, 3, 1).contiguous().to(dev)
        w2 = w.permute(0, 2, 3, 1).reshape(Cout, KH * KH * Cin).contiguous().to(dev)
        out = torch.empty(B, OH, OW, Cout, device=dev)
        L.gemm(xn, w2, out.view(-1, Cout), conv=dict(IH=IH, IW=IW, Cin=Cin, OH=OH, OW=OW, KW=KH, SH=S, SW=S, PH=P, PW=P))
        assert _rel(out.permute(0, 3, 1, 2).cpu(), ref) < 3e-6
        try:
            L.gemm_f32_set_big(0)
            small = torch.empty_like(out)
            L.gemm(xn, w2, small.view(-1, Cout), conv=dict(IH=IH, IW=IW, Cin=Cin, OH=OH, OW=OW, KW=KH, SH=S, SW=S, PH=P, PW=P))
        finally:
            L.gemm_f32_set_big(1)
        assert torch.equal(out, small)
    # scatter: rows (b, oy, ox) land at every second pixel of a [B, 2H, 2W, C] map (deconv phase, whmr.py:488-495)
    B, H, W, Cin, Cout = 2, 24, 24, 32, 64
    x = torch.randn(B, H, W, Cin, generator=g).to(dev)
    w = (torch.randn(Cout, 4 * Cin, generator=g) / math.sqrt(4 * Cin)).to(dev)
    big = torch.zeros(B, 2 * H, 2 * W, Cout, device=dev)
    conv = dict(IH=H, IW=W, Cin=Cin, OH=H, OW=W, KW=2, SH=1, SW=1, PH=1, PW=1)
    sc = dict(c_off=(2 * W + 1) * Cout, osb=4 * H * W * Cout, osy=4 * W * Cout, osx=2 * Cout)
    L.gemm(x, w, big, conv=conv, scatter=sc)
    dense = torch.empty(B * H * W, Cout, device=dev)
    L.gemm(x, w, dense, conv=conv)
    assert torch.equal(big[:, 1::2, 1::2], dense.view(B, H, W, Cout)) and not big[:, 0::2].any() and not big[:, :, 0::2].any()


@pytest.mark.parametrize('K,Mo,No,splits', [(12544, 768, 768, 0), (4096, 2304, 768, 0), (96, 128, 256, 1), (32, 256, 256, 0), (1056, 384, 512, 3),
                                            (12288, 3072, 768, 0), (2048, 768, 3072, 1), (2080, 64, 512, 0), (640, 192, 256, 2)])
def test_gemm_tn(dev, K, Mo, No, splits):
    """weight-gradient product straight from reduction-major operands (gemm_tn.hip: transposing LDS reads): C = A^T . B against float64,
    bitwise repeatable, strided rows (operands that are column slices of wider buffers)"""
    from whmr_amd import _lib as L
    g = torch.Generator().manual_seed(K + Mo + No)
    a = (torch.randn(K, Mo + 64, generator=g) * 0.5).bfloat16()
    b = (torch.randn(K, No, generator=g) * 0.5).bfloat16()
    ad, bd = a.to(dev)[:, 32:32 + Mo], b.to(dev)                          # a: a 64-B-offset column slice (row stride Mo + 64)
    assert L.gemm_tn_ok(ad, bd)
    out = torch.full((Mo, No), float('nan'), device=dev)
    db = torch.full((Mo,), float('nan'), device=dev)
    L.gemm_tn(ad, bd, out, splits=splits, db=db)
    ref = (a[:, 32:32 + Mo].double().t() @ b.double()).float()
    assert _rel(out.cpu(), ref) < 2e-5
    assert _rel(db.cpu(), a[:, 32:32 + Mo].double().sum(0).float()) < 2e-5             # column sums of A in the same pass (bias gradient)
    out2 = torch.empty_like(out)
    L.gemm_tn(ad, bd, out2, splits=splits)
    assert torch.equal(out, out2)
    assert not L.gemm_tn_ok(ad[:, :40], bd) and not L.gemm_tn_ok(ad[:-1], bd[:-1])


@pytest.mark.parametrize('K,shapes', [(12288, [(768, 3072), (3072, 768), (768, 768), (2304, 768)]), (4096, [(1024, 1024), (3072, 1024)]),
                                      (96, [(256, 256), (512, 256), (256, 512)]), (2048, [(4096, 1024), (1024, 4096), (1024, 1024), (3072, 1024)])])
def test_gemm_tn_group(dev, K, shapes):
    """the weight gradients of one transformer layer in ONE launch (whmr_gemm_tn_bf16_group): every product against float64 and against the single
    launch (same kernel body, another slice count: last bits only), bias sums included, bitwise repeatable; ViT-B (two slices), a 2-item group,
    a K too short to slice, ViT-L (208 tiles: unsliced, direct stores)"""
    from whmr_amd import _lib as L
    g = torch.Generator().manual_seed(K + len(shapes))
    jobs, refs = [], []
    for i, (Mo, No) in enumerate(shapes):
        a = (torch.randn(K, Mo + 64, generator=g) * 0.5).bfloat16()
        b = (torch.randn(K, No, generator=g) * 0.5).bfloat16()
        ad, bd = a.to(dev)[:, 32:32 + Mo], b.to(dev)
        out = torch.full((Mo, No), float('nan'), device=dev)
        db = torch.full((Mo,), float('nan'), device=dev) if i != 1 else None
        jobs.append((ad, bd, out, db))
        refs.append(((a[:, 32:32 + Mo].double().t() @ b.double()).float(), a[:, 32:32 + Mo].double().sum(0).float()))
    assert L.gemm_tn_group_ok(jobs)
    L.gemm_tn_group(jobs)
    first = [(o.clone(), None if d is None else d.clone()) for _, _, o, d in jobs]
    for (ad, bd, out, db), (ref, dref) in zip(jobs, refs):
        assert _rel(out.cpu(), ref) < 2e-5
        assert db is None or _rel(db.cpu(), dref) < 2e-5
        single = torch.empty_like(out)
        L.gemm_tn(ad, bd, single)
        assert _rel(out.cpu(), single.cpu()) < 2e-6
    for _, _, o, d in jobs:
        o.fill_(float('nan'))
    L.gemm_tn_group(jobs)
    assert all(torch.equal(o, f[0]) and (d is None or torch.equal(d, f[1])) for (_, _, o, d), f in zip(jobs, first))
    assert not L.gemm_tn_group_ok(jobs + jobs + jobs) and not L.gemm_tn_group_ok([(jobs[0][0][:, :128], jobs[0][1], None, None)])


def test_conv_dw_tn(dev):
    """convolution weight gradients from the gathering TN kernel (no column matrix) against torch autograd on the CPU: 3x3 s1 p1 (IUV head),
    7x7 s3 p0 (Tz head) and ConvTranspose2d k4 s2 p1 (deconv stages), incl. forced split-K"""
    from whmr_amd import _lib as L
    g = torch.Generator().manual_seed(3)
    for (B, Cin, IH, IW, Cout, KH, S, P, splits) in ((2, 256, 16, 12, 128, 3, 1, 1, 0), (2, 256, 34, 25, 128, 7, 3, 0, 5), (4, 512, 8, 8, 256, 3, 1, 1, 1)):
        x = (torch.randn(B, Cin, IH, IW, generator=g) * 0.5).bfloat16().float()
        w = torch.zeros(Cout, Cin, KH, KH, requires_grad=True)
        y = F.conv2d(x, w, stride=S, padding=P)
        OH, OW = y.shape[2:]
        dy = (torch.randn(y.shape, generator=g) * 0.5).bfloat16().float()
        if (B * OH * OW) % 32:
            continue
        y.backward(dy)
        a = dy.permute(0, 2, 3, 1).reshape(-1, Cout).bfloat16().contiguous().to(dev)
        img = x.permute(0, 2, 3, 1).bfloat16().contiguous().to(dev)
        assert L.conv_dw_tn_ok(a, img)
        out = torch.empty(Cout, KH * KH * Cin, device=dev)
        db = torch.empty(Cout, device=dev)
        L.conv_dw_tn(a, img, out, OH, OW, KH, KH, S, P, splits=splits, db=db)
        got = out.view(Cout, KH, KH, Cin).permute(0, 3, 1, 2).cpu()
        assert _rel(got, w.grad) < 2e-5, (KH, S, P)
        assert _rel(db.cpu(), dy.sum((0, 2, 3))) < 2e-5
    # 64 output channels are outside the envelope (the 64-row tile of the gathering kernel is not built, gemm_tn.hip): callers widen dY to 128 columns
    assert not L.conv_dw_tn_ok(torch.empty(64, 64, dtype=torch.bfloat16, device=dev), img) and L.conv_dw_tn_ok(torch.empty(64, 128, dtype=torch.bfloat16, device=dev), img)
    # ConvTranspose2d(k4, s2, p1): dW[ci, co, ky, kx] = sum x[b, iy, ix, ci] dz[b, 2 iy - 1 + ky, 2 ix - 1 + kx, co]
    B, Cin, H, W, Cout = 2, 256, 8, 6, 256
    x = (torch.randn(B, Cin, H, W, generator=g) * 0.5).bfloat16().float()
    wt = torch.zeros(Cin, Cout, 4, 4, requires_grad=True)
    z = F.conv_transpose2d(x, wt, stride=2, padding=1)
    dz = (torch.randn(z.shape, generator=g) * 0.5).bfloat16().float()
    z.backward(dz)
    a = x.permute(0, 2, 3, 1).reshape(-1, Cin).bfloat16().contiguous().to(dev)
    img = dz.permute(0, 2, 3, 1).bfloat16().contiguous().to(dev)
    out = torch.empty(Cin, 16 * Cout, device=dev)
    L.conv_dw_tn(a, img, out, H, W, 4, 4, 2, 1)
    assert _rel(out.view(Cin, 4, 4, Cout).permute(0, 3, 1, 2).cpu(), wt.grad) < 2e-5


def _conv_case(dev, dtype, tol):
    """implicit GEMM: Conv2d k7 s3 (Tz head conv, whmr.py:419) on an NHWC image"""
    from whmr_amd import _lib as L
    g = torch.Generator().manual_seed(11)
    B, Cin, IH, IW, Cout, KH, KW, S = 2, 64, 23, 20, 128, 7, 7, 3
    x = torch.randn(B, Cin, IH, IW, generator=g)
    w = torch.randn(Cout, Cin, KH, KW, generator=g) / math.sqrt(Cin * KH * KW)
    if dtype == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    ref = F.conv2d(x, w, stride=S)                                       # [B, Cout, OH, OW]
    OH, OW = ref.shape[2:]
    xn = x.permute(0, 2, 3, 1).contiguous().to(dtype).to(dev)             # NHWC
    w2 = w.permute(0, 2, 3, 1).reshape(Cout, KH * KW * Cin).contiguous().to(dtype).to(dev)
    out = torch.empty(B, OH, OW, Cout, device=dev)
    L.gemm(xn, w2, out.view(-1, Cout), conv=dict(IH=IH, IW=IW, Cin=Cin, OH=OH, OW=OW, KW=KW, SH=S, SW=S, PH=0, PW=0))
    assert _rel(out.permute(0, 3, 1, 2).cpu(), ref) < tol


def test_conv_gather_bf16(dev):
    _conv_case(dev, torch.bfloat16, 2e-5)


def test_conv_gather_f32(dev):
    _conv_case(dev, torch.float32, 5e-6)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_deconv_phases(dev, dtype):
    """ConvTranspose2d(k4,s2,p1) as 4 sub-pixel 2x2 convs with scatter stores (whmr.py:488-498)"""
    from whmr_amd import _lib as L
    g = torch.Generator().manual_seed(5)
    B, Cin, H, W, Cout = 2, 64, 6, 5, 128
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cin, Cout, 4, 4, generator=g) / math.sqrt(Cin * 4)
    if dtype == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    ref = F.relu(F.conv_transpose2d(x, w, stride=2, padding=1))
    xn = x.permute(0, 2, 3, 1).contiguous().to(dtype).to(dev)
    out = torch.zeros(B, 2 * H, 2 * W, Cout, device=dev)
    for py in range(2):
        for px in range(2):
            # taps a,b in {0,1}: iy = y + py - 1 + a, ky = 3 - py - 2a
            wp = torch.stack([torch.stack([w[:, :, 3 - py - 2 * a, 3 - px - 2 * b] for b in range(2)], 0)
                              for a in range(2)], 0)                     # [a, b, Cin, Cout]
            w2 = wp.permute(3, 0, 1, 2).reshape(Cout, 4 * Cin).contiguous().to(dtype).to(dev)
            L.gemm(xn, w2, out, act=L.ACT_RELU,
                   conv=dict(IH=H, IW=W, Cin=Cin, OH=H, OW=W, KW=2, SH=1, SW=1, PH=1 - py, PW=1 - px),
                   scatter=dict(c_off=(py * 2 * W + px) * Cout, osb=4 * H * W * Cout, osy=2 * 2 * W * Cout, osx=2 * Cout))
    assert _rel(out.permute(0, 3, 1, 2).cpu(), ref) < (2e-5 if dtype == torch.bfloat16 else 2e-6)


def test_deconv_phases_batched_launch(dev):
    """the 4 sub-pixel phases in ONE launch (n_phase = 4) == ConvTranspose2d(k4,s2,p1) + ReLU"""
    from whmr_amd import _lib as L
    g = torch.Generator().manual_seed(6)
    B, Cin, H, W, Cout = 3, 128, 7, 5, 256
    x = (torch.randn(B, Cin, H, W, generator=g)).bfloat16().float()
    w = (torch.randn(Cin, Cout, 4, 4, generator=g) / math.sqrt(Cin * 4)).bfloat16().float()
    shift = torch.randn(Cout, generator=g)
    ref = F.relu(F.conv_transpose2d(x, w, stride=2, padding=1) + shift[None, :, None, None])
    xn = x.permute(0, 2, 3, 1).contiguous().bfloat16().to(dev)
    ph = []
    for py in range(2):
        for px in range(2):
            taps = [w[:, :, 3 - py - 2 * a, 3 - px - 2 * b] for a in range(2) for b in range(2)]
            ph.append(torch.stack(taps, 0).permute(2, 0, 1).reshape(Cout, -1))
    wst = torch.stack(ph, 0).contiguous().bfloat16().to(dev)
    out = torch.zeros(B, 2 * H, 2 * W, Cout, device=dev)
    L.gemm(xn, wst, out, bias=shift.to(dev), act=L.ACT_RELU,
           conv=dict(IH=H, IW=W, Cin=Cin, OH=H, OW=W, KW=2, SH=1, SW=1, PH=1, PW=1),
           scatter=dict(c_off=0, osb=4 * H * W * Cout, osy=4 * W * Cout, osx=2 * Cout), phases=dict(cy=2 * W * Cout, cx=Cout))
    assert _rel(out.permute(0, 3, 1, 2).cpu(), ref) < 2e-5


@pytest.mark.parametrize('C', [768, 216, 1024])
@pytest.mark.parametrize('bf16', [False, True])
def test_layernorm(dev, C, bf16):
    from whmr_amd import _lib as L
    g = torch.Generator().manual_seed(C)
    x = torch.randn(391, C, generator=g) * 3 + 0.5
    w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
    out = torch.empty(391, C, device=dev, dtype=torch.bfloat16 if bf16 else torch.float32)
    L.layernorm(x.to(dev), w.to(dev), b.to(dev), out, 1e-6)
    ref = F.layer_norm(x, (C,), w, b, 1e-6)
    assert _rel(out.float().cpu(), ref) < (1e-2 if bf16 else 2e-6)


@pytest.mark.parametrize('N', [196, 192, 5, 37])
def test_attention_bf16(dev, N):
    from whmr_amd import _lib as L
    g = torch.Generator().manual_seed(N)
    B, H, d = 3, 12, 64
    qkv = (torch.randn(B, N, 3, H, d, generator=g) * 1.5).bfloat16()
    q, k, v = qkv.float().permute(2, 0, 3, 1, 4)
    ref = ((q * d ** -0.5) @ k.transpose(-2, -1)).softmax(-1) @ v
    ref = ref.transpose(1, 2).reshape(B, N, H * d)
    out = torch.empty(B, N, H * d, device=dev, dtype=torch.bfloat16)
    L.attention(qkv.to(dev).view(B, N, 3 * H * d), out, B, N, H, d, d ** -0.5)
    assert _rel(out.float().cpu(), ref) < 2e-2


@pytest.mark.parametrize('N,H,d', [(196, 12, 64), (5, 2, 108), (192, 12, 64), (70, 3, 32), (33, 2, 64), (256, 3, 64), (100, 16, 64)])
def test_attention_f32(dev, N, H, d):
    """fp32 attention core: the matrix-pipe kernel (d = 64, 32 < N <= 256: exact-f32 MFMA, online softmax) and the VALU kernel (other shapes,
    and as the A/B partner: whmr_attention_set_variant bit 3) against float64"""
    from whmr_amd import _lib as L
    g = torch.Generator().manual_seed(N + d)
    B = 2
    qkv = torch.randn(B, N, 3, H, d, generator=g) * 1.5
    q, k, v = qkv.double().permute(2, 0, 3, 1, 4)
    ref = ((q * d ** -0.5) @ k.transpose(-2, -1)).softmax(-1) @ v
    ref = ref.transpose(1, 2).reshape(B, N, H * d).float()
    out = torch.empty(B, N, H * d, device=dev)
    L.attention(qkv.to(dev).view(B, N, 3 * H * d), out, B, N, H, d, d ** -0.5)
    assert _rel(out.cpu(), ref) < 5e-6
    try:
        L.attention_set_variant(1 | 8)                                  # VALU kernel
        valu = torch.empty_like(out)
        L.attention(qkv.to(dev).view(B, N, 3 * H * d), valu, B, N, H, d, d ** -0.5)
    finally:
        L.attention_set_variant(1)
    assert _rel(valu.cpu(), ref) < 5e-6 and _rel(out, valu) < 5e-6


def test_patch_im2col(dev):
    from whmr_amd import _lib as L
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 3, 64, 80, generator=g)[:, :, :, 8:-8]             # sliced view like demo/tester.py:152
    out = torch.empty(2 * 4 * 4, 768, device=dev)
    L.patch_im2col(x.to(dev)[:, :, :, :], out, 16, 2)
    ref = F.unfold(x, 16, padding=2, stride=16).transpose(1, 2).reshape(-1, 768)
    assert torch.equal(out.cpu(), ref)


@pytest.mark.parametrize('tile', [None, 64, 65, 128, 257, 160, 224])
@pytest.mark.parametrize('out_bf16', [True, False])
def test_gemm_bf16_skip_before_relu(dev, tile, out_bf16):
    """ResNet bottleneck epilogue: relu(a.w^T + bias + skip) with a bf16 skip tensor (epi_flags bits 0|1)."""
    from whmr_amd import _lib as L
    g = torch.Generator().manual_seed(11)
    M, N, K = 1000, 256, 64
    a = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).bfloat16()
    bias = torch.randn(N, generator=g)
    skip = torch.randn(M, N, generator=g).bfloat16()
    ref = F.relu(a.float() @ w.float().t() + bias + skip.float())
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    L.gemm(a.to(dev), w.to(dev), out, bias=bias.to(dev), residual=skip.to(dev), act=L.ACT_RELU, res_first=True, tile=tile)
    assert _rel(out.float().cpu(), ref) < (1e-2 if out_bf16 else 1e-5)
    # fp32 skip, added before the activation
    out2 = torch.empty(M, N, device=dev)
    L.gemm(a.to(dev), w.to(dev), out2, bias=bias.to(dev), residual=skip.float().to(dev), act=L.ACT_RELU, res_first=True, tile=tile)
    assert _rel(out2.cpu(), ref) < 1e-5


def test_gemm_f32_skip_before_relu(dev):
    from whmr_amd import _lib as L
    g = torch.Generator().manual_seed(12)
    M, N, K = 333, 200, 96
    a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K)
    bias, skip = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    out = torch.empty(M, N, device=dev)
    L.gemm(a.to(dev), w.to(dev), out, bias=bias.to(dev), residual=skip.to(dev), act=L.ACT_RELU, res_first=True)
    assert _rel(out.cpu(), F.relu(a @ w.t() + bias + skip)) < 5e-6


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float32])
def test_nhwc_pools_and_stem_im2col(dev, dt):
    from whmr_amd import _lib as L
    g = torch.Generator().manual_seed(13)
    x = torch.randn(2, 64, 37, 45, generator=g).to(dt)
    xn = x.permute(0, 2, 3, 1).contiguous().to(dev)
    ref = F.max_pool2d(x.float(), 3, 2, 1).permute(0, 2, 3, 1)
    assert torch.equal(L.maxpool_nhwc(xn, 3, 2, 1).float().cpu(), ref)
    assert _rel(L.avgpool_nhwc(xn).cpu(), x.float().mean(dim=(2, 3))) < 2e-6
    if dt == torch.float32:
        img = torch.randn(2, 3, 50, 71, generator=g)
        cols, OH, OW = L.conv_im2col(img.to(dev)[:, :, 1:, 2:], 7, 7, 2, 3, 192)          # strided view in, like a cropped frame
        ref = F.unfold(img[:, :, 1:, 2:], 7, padding=3, stride=2).transpose(1, 2).reshape(-1, 21, 7)      # (ci*7 + ky, kx)
        assert (OH, OW) == (25, 35)
        got = cols.float().cpu().view(-1, 24, 8)
        assert torch.equal(got[:, :21, :7], ref.bfloat16().float()) and not got[:, 21:].any() and not got[:, :, 7].any()


@pytest.mark.parametrize('numerics,tol', [('fp32', 1e-4), ('bf16x3', 1e-4), ('bf16', 4e-2)])
def test_cam_model_resnet50(dev, numerics, tol):
    """the HIP NHWC ResNet-50 of cam_model against the CPU fp32 oracle (oracle/whmr.py::cam_model_forward: pooled
    features, the three 256-bin logit vectors, the soft-argmax angles and the Rx(pitch).Rz(roll) rotation)."""
    from oracle import synth
    from oracle import whmr as OW
    from whmr_amd.models.cam_model import CameraRegressorNetwork, batch_euler2matrix, convert_preds_to_angles
    sd = synth.make_state_dict(0, synth.make_assets(0))
    m = CameraRegressorNetwork()
    m.load_state_dict({k[len('cam_model.'):]: v for k, v in sd.items() if k.startswith('cam_model.')}, strict=True)
    m.numerics = numerics
    m = m.to(dev).eval()
    x = torch.randn(2, 3, 160, 224, generator=torch.Generator().manual_seed(5))
    taps = {}
    with torch.no_grad():
        R_ref, _ = OW.cam_model_forward(sd, x, taps=taps)
    out, feat = m(x.to(dev))
    assert _rel(feat.cpu(), taps['feat']) < tol
    for o, name in zip(out, ('vfov', 'pitch', 'roll')):
        assert o.shape == taps['logits_' + name].shape and _rel(o.cpu(), taps['logits_' + name]) < tol, name
    vfov, pitch, roll = convert_preds_to_angles(*out)
    assert _rel(pitch.cpu(), taps['angle_pitch']) < tol and _rel(roll.cpu(), taps['angle_roll']) < tol and _rel(vfov.cpu(), taps['angle_vfov']) < tol
    assert pitch.abs().min() > 0.2 and roll.abs().min() > 0.2             # the synthetic head is far from the mid-range (0, 0) case
    R = batch_euler2matrix(torch.stack([pitch, torch.zeros_like(pitch), roll], 1).float())
    assert _rel(R.cpu(), R_ref) < tol
    m.train()
    with pytest.raises(RuntimeError):
        m(x.to(dev))
    with pytest.raises(RuntimeError):                                      # the module tree holds parameters only
        m.backbone(x.to(dev))


@pytest.mark.parametrize('out_bf16', [True, False])
def test_gemm_bf16_split_k(dev, out_bf16):
    """Few-tile deep-K shapes take the split-K route (gemm_bf16.hip); same epilogue semantics, deterministic."""
    from whmr_amd import _lib as L
    g = torch.Generator().manual_seed(21)
    M, N, K = 475, 512, 4608
    a = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).bfloat16()
    bias = torch.randn(N, generator=g)
    skip = torch.randn(M, N, generator=g).bfloat16()
    ref = F.relu(a.float() @ w.float().t() + bias + skip.float())
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    L.gemm(a.to(dev), w.to(dev), out, bias=bias.to(dev), residual=skip.to(dev), act=L.ACT_RELU, res_first=True)
    assert _rel(out.float().cpu(), ref) < (1e-2 if out_bf16 else 2e-5)
    out_b = torch.empty_like(out)
    L.gemm(a.to(dev), w.to(dev), out_b, bias=bias.to(dev), residual=skip.to(dev), act=L.ACT_RELU, res_first=True)
    assert torch.equal(out, out_b)
    # the unsplit kernel (explicit tile) agrees to fp32 summation-order noise
    out_t = torch.empty(M, N, device=dev)
    out_s = torch.empty(M, N, device=dev)
    L.gemm(a.to(dev), w.to(dev), out_t, bias=bias.to(dev), tile=65)
    L.gemm(a.to(dev), w.to(dev), out_s, bias=bias.to(dev))
    assert _rel(out_s, out_t) < 1e-5


def test_conv3x3_bf16_split_k(dev):
    """ResNet layer4 3x3 conv on one 600x800 frame: implicit GEMM M = 475, K = 4608 through the split-K route."""
    from whmr_amd import _lib as L
    g = torch.Generator().manual_seed(22)
    B, H, W, Cin, Cout = 1, 19, 25, 512, 512
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)).bfloat16()
    bias = torch.randn(Cout, generator=g)
    ref = F.relu(F.conv2d(x.float(), w.float(), bias, stride=1, padding=1)).permute(0, 2, 3, 1)
    out = torch.empty(B, H, W, Cout, device=dev)
    L.gemm(x.permute(0, 2, 3, 1).contiguous().to(dev), w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to(dev), out,
           bias=bias.to(dev), act=L.ACT_RELU, conv=dict(IH=H, IW=W, Cin=Cin, OH=H, OW=W, KW=3, SH=1, SW=1, PH=1, PW=1))
    assert _rel(out.cpu(), ref) < 2e-5


def test_crop_normalize_matches_oracle(dev):
    """all person crops of a frame in one launch == the per-detection cv2-style CPU restatement, bit for bit."""
    import numpy as np
    from oracle import crop as OC
    from whmr_amd.datasets.img_utils import crop_persons, get_single_image_crop_demo
    rng = np.random.default_rng(5)
    frame = rng.integers(0, 256, size=(360, 640, 3), dtype=np.uint8)
    boxes = [(320.0, 180.0, 200.0, 200.0), (20.5, 30.25, 150.0, 150.0), (600.0, 340.0, 333.3, 333.3), (100.0, 100.0, 37.0, 37.0),
             (319.7, 12.0, 512.0, 512.0)]                                  # centred, off the top-left / bottom-right edges, up- and down-scaled
    out, raw = crop_persons(torch.from_numpy(frame).to(dev), boxes, crop_size=256, scale=1.0, want_raw=True)
    sl = crop_persons(torch.from_numpy(frame).to(dev), boxes, crop_size=256, scale=1.0, x_slice=(32, 224))
    for i, b in enumerate(boxes):
        ref, ref_raw, _ = OC.get_single_image_crop_demo(frame, b, None, scale=1.0, crop_size=256)
        assert np.array_equal(raw[i].cpu().numpy(), ref_raw)
        assert np.array_equal(out[i].cpu().numpy(), ref)
        assert np.array_equal(sl[i].cpu().numpy(), ref[:, :, 32:224])          # demo/tester.py:151
    one, one_raw, _ = get_single_image_crop_demo(torch.from_numpy(frame).to(dev), boxes[0], None, scale=1.2, crop_size=224)
    ref, ref_raw, _ = OC.get_single_image_crop_demo(frame, boxes[0], None, scale=1.2, crop_size=224)
    assert np.array_equal(one.cpu().numpy(), ref) and np.array_equal(one_raw.cpu().numpy(), ref_raw)


def test_conv_gather_chunk_major_k_order(dev):
    """epi_flags bit 3: weight columns ordered (ci chunk of 64, ky, kx, ci in chunk); same conv result as the (ky, kx, ci) order."""
    from whmr_amd import _lib as L
    g = torch.Generator().manual_seed(31)
    B, Cin, IH, IW, Cout, KH, KW, S = 2, 128, 23, 20, 64, 7, 7, 3
    x = torch.randn(B, Cin, IH, IW, generator=g).bfloat16()
    w = (torch.randn(Cout, Cin, KH, KW, generator=g) / math.sqrt(Cin * KH * KW)).bfloat16()
    ref = F.conv2d(x.float(), w.float(), stride=S).permute(0, 2, 3, 1)
    OH, OW = ref.shape[1:3]
    xn = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wk = w.permute(0, 2, 3, 1)                                                            # [Cout, ky, kx, ci]
    w_cm = wk.reshape(Cout, KH * KW, Cin // 64, 64).permute(0, 2, 1, 3).reshape(Cout, -1).contiguous().to(dev)
    out = torch.empty(B, OH, OW, Cout, device=dev)
    for tile in (None, 65, 64):
        L.gemm(xn, w_cm, out.view(-1, Cout), conv=dict(IH=IH, IW=IW, Cin=Cin, OH=OH, OW=OW, KW=KW, SH=S, SW=S, PH=0, PW=0, chunk_major=True), tile=tile)
        assert _rel(out.cpu(), ref) < 2e-5


@pytest.mark.parametrize('tile', [None, 64, 65, 128, 192, 257, 320, 160, 224])
@pytest.mark.parametrize('act', [0, 1, 2])
def test_gemm_bf16_staged_epilogue(dev, tile, act):
    """bf16 outputs take the whole-tile bf16 staging epilogue (bias from LDS, activation in registers): ragged M, N % 8 == 0 only."""
    from whmr_amd import _lib as L
    g = torch.Generator().manual_seed(41 + act)
    M, N, K = 777, 328, 192
    a = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).bfloat16()
    bias = torch.randn(N, generator=g)
    pre = a.float() @ w.float().t() + bias
    ref = pre if act == 0 else (F.gelu(pre) if act == 1 else F.relu(pre))
    out = torch.full((M + 3, N), 7.0, device=dev, dtype=torch.bfloat16)
    L.gemm(a.to(dev), w.to(dev), out[:M], bias=bias.to(dev), act=act, tile=tile)
    assert _rel(out[:M].float().cpu(), ref) < 1e-2
    assert (out[M:] == 7.0).all()                                      # rows past M untouched


@pytest.mark.parametrize('M,N,K', [(64, 1024, 2149), (229, 1024, 64), (70, 229, 207), (320, 216, 864), (1024, 2149, 64), (5, 3, 1)])
def test_gemm_f32_reduction_major_operands(dev, M, N, K):
    """whmr_gemm_f32 with epi_flags bits 4 / 5 (L.gemm trans_a / trans_w): the operand is given as [K, M] / [K, N] -- the forms dY, X and W
    have in the backward products of nn.Linear -- against the float64 product, every combination, incl. rows that are only 4-byte aligned,
    ragged tile edges in every dimension, the split-K route (K >= 256) and a bias"""
    from whmr_amd import _lib as L
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g)
    b = torch.randn(N, generator=g)
    ref = (a.double() @ w.double().t() + b.double()).float()
    ad, wd, bd = a.to(dev), w.to(dev), b.to(dev)
    at, wt = ad.t().contiguous(), wd.t().contiguous()
    base = L.gemm(ad, wd, torch.empty(M, N, device=dev), bias=bd)
    for ta, tw in ((True, False), (False, True), (True, True)):
        out = L.gemm(at if ta else ad, wt if tw else wd, torch.empty(M, N, device=dev), bias=bd, trans_a=ta, trans_w=tw)
        err = (out.cpu() - ref).abs().max() / ref.abs().max()
        assert err < 2e-6, (ta, tw, err.item())
        assert torch.equal(out, base), (ta, tw)                        # same k order of the MFMA chain: the same bits as the row-major call
    # a strided reduction-major A (a column block of a wider matrix, as dY of a fused multi-head Linear would be)
    if M >= 8:
        wide = torch.randn(K, M + 8, generator=g).to(dev)
        out = L.gemm(wide[:, 4:4 + M], wd, torch.empty(M, N, device=dev), trans_a=True)
        ref2 = (wide[:, 4:4 + M].t().double().cpu() @ w.double().t()).float()
        assert (out.cpu() - ref2).abs().max() / ref2.abs().max() < 2e-6
    with pytest.raises(RuntimeError):                                   # more than 1024 output rows: no reduction-major route
        L.gemm(torch.randn(K, 1100, device=dev), wd, torch.empty(1100, N, device=dev), trans_a=True)


@pytest.mark.parametrize('M,N,K', [(12288, 3072, 768), (392, 3072, 768), (1000, 256, 128)])
def test_gemm_bf16_gelu_epilogues_of_the_training_step(dev, M, N, K):
    """fc1 of the training forward: C = gelu(z) and C2 = z = a . w^T + b from one launch (whmr_gemm.C2); fc2's data gradient times the GELU
    derivative of the saved pre-activation in the epilogue (epi_flags bit 7).  z must be the bits of the plain launch; gelu / gelu' are the
    polynomial forms (|error| <= 1.9e-4 / 2.6e-4) of the exact erf expressions, applied to the bf16-rounded values."""
    from whmr_amd import _lib as L
    g = torch.Generator().manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g) * 0.5).to(dev).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.08).to(dev).bfloat16()
    b = torch.randn(N, generator=g).to(dev)
    plain = L.gemm(a, w, torch.empty(M, N, dtype=torch.bfloat16, device=dev), bias=b)
    hid, pre = torch.empty_like(plain), torch.empty_like(plain)
    L.gemm(a, w, hid, bias=b, act=L.ACT_GELU, pre_out=pre)
    assert torch.equal(pre, plain)
    z = pre.float()
    want = torch.nn.functional.gelu(z)
    err = (hid.float() - want).abs()
    assert (err <= 2.0e-4 + 2.0 ** -8 * want.abs()).all(), err.max().item()
    assert (z.abs() > 1.0).float().mean() > 0.1                                        # the test data reaches the curved part and the tails
    # backward: (a2 . w2^T) * gelu'(z)
    a2 = (torch.randn(M, K, generator=g) * 0.5).to(dev).bfloat16()
    lin = L.gemm(a2, w, torch.empty(M, N, dtype=torch.bfloat16, device=dev))
    out = L.gemm(a2, w, torch.empty(M, N, dtype=torch.bfloat16, device=dev), gelu_bwd_of=pre)
    zd = z.double()
    dgelu = 0.5 * (1 + torch.erf(zd / 2 ** 0.5)) + zd * torch.exp(-0.5 * zd * zd) / (2 * torch.pi) ** 0.5
    want = lin.double() * dgelu
    err = (out.double() - want).abs()
    assert (err <= 2.0 ** -8 * want.abs() + 2.7e-4 * lin.double().abs() + 1e-30).all(), (err / (want.abs() + 1e-3)).max().item()
    # outside the envelope: fp32 output / a residual next to the second output
    with pytest.raises((RuntimeError, AssertionError)):
        L.gemm(a, w, torch.empty(M, N, device=dev), bias=b, act=L.ACT_GELU, pre_out=torch.empty(M, N, device=dev))


@pytest.mark.parametrize('Bf,B', [(1, 1), (1, 64), (5, 5)])
def test_forward_glue_kernels_match_the_tensor_expressions(dev, Bf, B):
    """whmr_cam_head / whmr_orient_state / whmr_orient_tail (one launch each) against the tensor expressions they replace -- the restatements of
    utils/cam_utils.py:121-145 + pare softargmax1d / batch_euler2matrix in whmr_amd.models.cam_model (themselves checked against the CPU oracle
    by test_cam_model_*), rotmat_to_rot6d / unbiased_gram_schmidt / rotation_matrix_to_angle_axis of whmr_amd.utils.geometry (fixture-pinned)"""
    from whmr_amd import _lib as L
    from whmr_amd.models.cam_model import PITCH_RANGE, ROLL_RANGE, batch_euler2matrix, convert_preds_to_angles
    from whmr_amd.utils import geometry as G
    g = torch.Generator().manual_seed(Bf * 100 + B)
    logits = (torch.randn(Bf, 768, generator=g) * 3).to(dev)
    _, pitch, roll = convert_preds_to_angles(logits[:, :256], logits[:, 256:512], logits[:, 512:])
    if Bf == 1:
        pitch, roll = pitch.expand(B), roll.expand(B)
    z = torch.zeros(B, 1, device=dev)
    ref_cam = batch_euler2matrix(torch.cat([pitch[:, None], z, roll[:, None]], 1))
    ref_ren = batch_euler2matrix(torch.cat([-pitch[:, None], z, roll[:, None]], 1))
    cam, ren = L.cam_head(logits, 256, PITCH_RANGE, ROLL_RANGE, B)
    assert cam.shape == (B, 3, 3) and (cam - ref_cam).abs().max() < 2e-6 and (ren - ref_ren).abs().max() < 2e-6
    # orientation head state + tail
    rot = G.batch_rodrigues(torch.randn(B * 24, 3, generator=g).to(dev)).reshape(B, 24, 3, 3)
    xc = torch.full((B, 2200), float('nan'), device=dev)
    L.orient_state(cam, rot, xc, 2149)
    assert torch.equal(xc[:, 2149:2155], cam[:, :, :2].reshape(B, 6)) and torch.equal(xc[:, 2155:2164], rot[:, 0].reshape(B, 9))
    assert torch.isnan(xc[:, :2149]).all() and torch.isnan(xc[:, 2164:]).all()
    r = (rot[:, 0] + 0.1 * torch.randn(B, 3, 3, generator=g).to(dev)).reshape(B, 9).contiguous()
    aa = torch.randn(B, 72, generator=g).to(dev)
    g_pose, g_rot = L.orient_tail(r, aa, rot.contiguous())
    gs = G.unbiased_gram_schmidt(r.reshape(-1, 1, 3, 3))
    assert torch.equal(g_rot[:, 1:], rot[:, 1:]) and torch.equal(g_pose[:, 3:], aa[:, 3:])
    assert (g_rot[:, :1] - gs).abs().max() < 1e-6
    assert (g_pose[:, :3] - G.rotation_matrix_to_angle_axis(gs.reshape(-1, 3, 3))).abs().max() < 1e-5


@pytest.mark.parametrize('IH,IW,dt', [(41, 30, torch.bfloat16), (41, 30, torch.float32), (21, 17, torch.float32), (9, 7, torch.bfloat16)])
def test_tz_conv1_matches_conv2d(dev, IH, IW, dt):
    """second convolution of the Tz head (whmr.py:420, Conv2d(64, 5, k7, s2), no bias) on the NHWC map, tokens [B, 5, OH*OW] like the reference's
    reshape (whmr.py:571): four output pixels per wave; widths whose pixel count is not a multiple of four exercise the ragged last group"""
    from whmr_amd import _lib as L
    g = torch.Generator().manual_seed(IH * 100 + IW)
    B = 3
    x = torch.randn(B, IH, IW, 64, generator=g).to(dt)
    w = torch.randn(5, 64, 7, 7, generator=g) * 0.05
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w, stride=2).reshape(B, 5, -1)
    OH, OW = (IH - 7) // 2 + 1, (IW - 7) // 2 + 1
    tok = torch.full((B, 5, OH * OW), float('nan'), device=dev)
    L.tz_conv1(x.to(dev), w.permute(0, 2, 3, 1).reshape(5, 49, 64).contiguous().to(dev), tok)
    assert ref.shape == tok.shape
    assert _rel(tok.cpu(), ref) < 1e-5


@pytest.mark.parametrize('mode', ['fp32', 'bf16x3', 'bf16'])
def test_tz_composed_convolution_matches_the_two_convolutions(dev, mode):
    """whmr.py:418-421 / :567-571: Conv2d(256, 64, k7, s3) -> Conv2d(64, 5, k7, s2) (no bias, nothing in between) as ONE Conv2d(256, 5, k25, s6),
    evaluated as the space-to-depth implicit GEMM + ``whmr_tz_fold`` (models/whmr.py::_tz_tokens_composed) against torch's two conv2d calls in fp64.
    B = 3 (ragged last row tile); the bottom rows Y = 21 read past the map (zero source)."""
    from whmr_amd import _lib as L
    from whmr_amd.models.whmr import compose_tz_weights
    g = torch.Generator().manual_seed(11)
    B, H, W, C = 3, 128, 96, 256
    x = torch.relu(torch.randn(B, H, W, C, generator=g))
    w0 = torch.randn(64, C, 7, 7, generator=g) / math.sqrt(49 * C)
    w1 = torch.randn(5, 64, 7, 7, generator=g) / math.sqrt(49 * 64)
    ref = F.conv2d(F.conv2d(x.double().permute(0, 3, 1, 2), w0.double(), stride=3), w1.double(), stride=2).reshape(B, 5, -1)
    G = compose_tz_weights(w0, w1).to(dev)
    xd = x.to(dev)
    if mode == 'bf16':
        a, gw, Cp, halves, tol = xd.bfloat16(), G.bfloat16(), C, 1, 1e-2
    elif mode == 'bf16x3':
        a = torch.cat(L.split_bf16(xd), -1).contiguous()
        hi, lo = L.split_bf16(G)
        hi, lo = hi.view(128, 36, C), lo.view(128, 36, C)
        gw = torch.cat([torch.cat([hi, hi], -1), torch.cat([lo, torch.zeros_like(lo)], -1)], 0).reshape(256, -1).contiguous()
        Cp, halves, tol = 2 * C, 2, 2e-5
    else:
        a, gw, Cp, halves, tol = xd, G, C, 1, 1e-5
    P = torch.full((B * 22 * 16, gw.shape[0]), float('nan'), device=dev)
    L.gemm(a.view(B, H, 16, 6 * Cp), gw, P, conv=dict(IH=H, IW=16, Cin=6 * Cp, OH=22, OW=16, KW=1, SH=6, SW=1, PH=0, PW=0))
    tok = torch.full((B * 5, 216), float('nan'), device=dev)
    L.tz_fold(P, tok, B, 22, 16, 18, 12, halves=halves)
    assert _rel(tok.view(B, 5, -1).cpu(), ref) < tol
    if mode != 'fp32':          # two raw split-K planes (whmr_gemm_bf16_split_raw) added by the fold: what the model launches at batch 64
        P2 = torch.full((2, B * 22 * 16, gw.shape[0]), float('nan'), device=dev)
        L.gemm(a.view(B, H, 16, 6 * Cp), gw, P2, conv=dict(IH=H, IW=16, Cin=6 * Cp, OH=22, OW=16, KW=1, SH=6, SW=1, PH=0, PW=0),
               tile=192 if mode == 'bf16x3' else 64, raw_splits=2)
        tok2 = torch.full((B * 5, 216), float('nan'), device=dev)
        L.tz_fold(P2, tok2, B, 22, 16, 18, 12, halves=halves, nsplit=2, split_stride=P2[0].numel())
        assert _rel(tok2.view(B, 5, -1).cpu(), ref) < tol
        assert _rel(tok2.cpu(), tok.cpu()) < 1e-5


def test_tz_fold_split_planes(dev):
    """whmr_tz_fold over nsplit partial planes and two column halves == the same sums done by torch"""
    from whmr_amd import _lib as L
    g = torch.Generator().manual_seed(5)
    B, ns = 2, 3
    P = torch.randn(ns, B * 22 * 16, 256, generator=g)
    full = (P[:, :, :128] + P[:, :, 128:]).sum(0).view(B, 22, 16, 128)
    ref = torch.zeros(B, 5, 18, 12)
    for jA in range(5):
        for jB in range(5):
            for o in range(5):
                ref[:, o] += full[:, jA:jA + 18, jB:jB + 12, (jA * 5 + jB) * 5 + o]
    tok = torch.empty(B * 5, 216, device=dev)
    L.tz_fold(P.to(dev), tok, B, 22, 16, 18, 12, halves=2, nsplit=ns, split_stride=B * 22 * 16 * 256)
    assert _rel(tok.view(B, 5, 18, 12).cpu(), ref) < 1e-5


def test_mat_to_aa_backward_matches_autograd(dev):
    """whmr_mat_to_aa_bwd against torch autograd through the oracle's rotation_matrix_to_angle_axis (bit-identical to the reference's
    utils/geometry.py:54-83 on the fixture): rotations that take each of the four quaternion branches (angles up to pi), and the NON-orthonormal
    matrices the training graph feeds it (no Gram-Schmidt in training, whmr.py:129,174)."""
    from oracle import geometry as G
    from whmr_amd import _lib as L
    g = torch.Generator().manual_seed(2)
    axis = torch.nn.functional.normalize(torch.randn(600, 3, generator=g), dim=1)
    ang = torch.cat([torch.rand(300, generator=g) * 3.1, 3.0 + torch.rand(300, generator=g) * 0.14])          # well inside (0, pi)
    R = G.batch_rodrigues(axis * ang[:, None])
    R = torch.cat([R, R[:200] + 0.05 * torch.randn(200, 3, 3, generator=g)])                                    # + perturbed (not rotations)
    d2, d0, d1 = R[:, 2, 2], R[:, 0, 0], R[:, 1, 1]
    branch = torch.where(d2 < 1e-6, torch.where(d0 > d1, 0, 1), torch.where(d0 < -d1, 2, 3))
    assert all(int((branch == b).sum()) >= 10 for b in range(4)), branch.bincount()
    cot = torch.randn(R.shape[0], 3, generator=g)
    Rr = R.clone().requires_grad_(True)
    aa = G.rotation_matrix_to_angle_axis(Rr)
    (aa * cot).sum().backward()
    got = L.mat_to_aa_bwd(R.to(dev), cot.to(dev)).cpu().view(-1, 3, 3)
    fwd = L.mat_to_aa(R.reshape(-1, 9).to(dev)).cpu()
    assert _rel(fwd, aa.detach()) < 1e-5
    ref = Rr.grad
    scale = ref.abs().amax(dim=(1, 2), keepdim=True).clamp_min(1e-6)
    worst = ((got - ref).abs() / scale).max().item()
    assert worst < 2e-4, worst          # per matrix, relative to its largest gradient entry (near pi the map is ill-conditioned: fp32 both sides)


def test_measurement_aids_ceilings_and_clock_probe(dev):
    """csrc/ceilings.hip (bench.py's roofline.attainable / hbm_attainable_GBps / sclk_mhz_observed): the copy kernel copies, the MFMA stream and the
    clock probe return physically plausible figures, and a probe that is never released ends by itself (bounded wait)."""
    from whmr_amd import _lib as L
    g = torch.Generator().manual_seed(1)
    src = torch.randint(0, 255, (1 << 20,), generator=g, dtype=torch.uint8).to(dev)
    dst = torch.zeros_like(src)
    L._check(L.lib().whmr_hbm_copy(src.data_ptr(), dst.data_ptr(), src.numel(), L._stream()), 'whmr_hbm_copy')
    assert torch.equal(src, dst)
    assert L.lib().whmr_hbm_copy(src.data_ptr(), dst.data_ptr(), 100, L._stream()) != 0          # not a multiple of 16 bytes: rejected
    gbps = L.hbm_copy_ceiling(dev, mbytes=256, reps=3)
    assert 500.0 < gbps < 16000.0, gbps                       # read + write bytes per second; the datasheet peak is 8 TB/s each way
    mf = L.mfma_ceiling(dev, seconds=0.03)
    assert 300.0 < mf['tflops'] < 2600.0 and 500.0 < mf['sclk_mhz'] < 3000.0, mf      # never above the 2.5 PF dense bf16 peak
    with L.ClockProbe(dev) as cp:
        a = torch.randn(2048, 2048, device=dev)
        for _ in range(5):
            a = a @ a * 1e-3
    assert not cp.timed_out and 500.0 < cp.mhz < 3000.0 and cp.seconds > 0, (cp.mhz, cp.seconds)
    state = torch.zeros(6, dtype=torch.int64, device=dev)      # never released: leaves through its time limit and says so
    L._check(L.lib().whmr_clock_probe_begin(state.data_ptr(), 0.01, L._stream()), 'whmr_clock_probe_begin')
    torch.cuda.synchronize()
    st = state.tolist()
    assert st[5] == 1 and 0.009 < (st[3] - st[1]) / 1e8 < 0.2, st
