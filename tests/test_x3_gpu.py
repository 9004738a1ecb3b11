"""The bf16x3 (split-bf16) numerics of the ViT inference path -- every GEMM / attention operand a hi + lo bf16 pair, three bf16 MFMAs per product,
fp32 accumulate (csrc/gemm_blk_x3.hip, attention_x3.hip, the *_x3 forms in vit_ops.hip) -- against float64 torch on the same inputs, the
reference fixtures and the CPU oracle.  The point of the mode: the reference's fp32 arithmetic (vit.py:61-165) reproduced to ~1e-6 on the bf16
matrix pipes, i.e. the north-star 1e-4 tolerance AND MFMA-rate throughput in one mode."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN, ew_err

pytestmark = pytest.mark.gpu

TILES = [0x44, 0x55, 0x43, 0x33, 0x32, 0x22, 0x54, 0x21]


def _rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


def _pair(L, t, dev):
    hi, lo = L.split_bf16(t)
    return L.to_blocked(hi.to(dev)), L.to_blocked(lo.to(dev))


def _join(L, hi, lo, R):
    return L.from_blocked(hi, R).double().cpu() + L.from_blocked(lo, R).double().cpu()


def test_split_bf16_pair_carries_16_bits():
    from whmr_amd import _lib as L
    x = torch.randn(4096, generator=torch.Generator().manual_seed(0)) * 3
    hi, lo = L.split_bf16(x)
    assert hi.dtype == lo.dtype == torch.bfloat16
    assert ((hi.double() + lo.double() - x.double()).abs() / x.double().abs()).max() < 2.0 ** -16


@pytest.mark.parametrize('M,N,K', [(392, 768, 768), (1000, 256, 32), (1001, 512, 64), (777, 256, 96), (12544, 2304, 768), (3000, 768, 3072)])
@pytest.mark.parametrize('tile', [0] + TILES)
def test_gemm_blk_x3(dev, M, N, K, tile):
    """every epilogue x every tile on fp32 random operands given as hi / lo pairs, against the float64 product of the SAME fp32 operands:
    the split drops 2^-17 per operand and the lo.lo term, fp32 accumulation does the rest -- gated at 2e-5 of the result's scale (a plain
    bf16 GEMM sits at ~4e-3 on these inputs), asymmetric shapes and M tails as in test_gemm_blk"""
    from whmr_amd import _lib as L
    if M > 4000 and tile not in (0, 0x44, 0x43, 0x55):
        pytest.skip('full-size shape: chooser + the tiles the ViT uses')
    g = torch.Generator().manual_seed(M + N + K + tile)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    pos = torch.randn(196, N, generator=g)
    (ah, al), (wh, wl) = _pair(L, a, dev), _pair(L, w, dev)
    nb = ah.shape[0]
    lin = a.double() @ w.double().t() + bias.double()
    tol = 2e-5
    oh = torch.full((nb, N // 8, 32, 8), float('nan'), device=dev, dtype=torch.bfloat16)
    ol = torch.full_like(oh, float('nan'))
    L.gemm_blk(ah, wh, oh, M, bias=bias.to(dev), epi=L.EPI_BF16, tile=tile, a_lo=al, w_lo=wl, out_lo=ol)
    assert _rel(_join(L, oh, ol, M), lin) < tol
    assert _rel(L.from_blocked(oh, M).double().cpu(), lin) < 1e-2 and L.from_blocked(ol, M).float().abs().max() < 2e-2 * lin.abs().max()     # hi ~ the bf16 result, lo a correction
    L.gemm_blk(ah, wh, oh, M, epi=L.EPI_BF16, tile=tile, a_lo=al, w_lo=wl, out_lo=ol)                                              # no bias
    assert _rel(_join(L, oh, ol, M), a.double() @ w.double().t()) < tol
    L.gemm_blk(ah, wh, oh, M, bias=bias.to(dev), epi=L.EPI_BF16_GELU, tile=tile, a_lo=al, w_lo=wl, out_lo=ol)                      # exact erf GELU
    assert _rel(_join(L, oh, ol, M), F.gelu(lin)) < tol
    t = L.to_blocked(res.to(dev))
    L.gemm_blk(ah, wh, t, M, bias=bias.to(dev), epi=L.EPI_F32_RES, res=t, tile=tile, a_lo=al, w_lo=wl)
    assert _rel(L.from_blocked(t, M).cpu(), lin + res.double()) < tol
    t2 = torch.full((nb, N // 4, 32, 4), float('nan'), device=dev)
    L.gemm_blk(ah, wh, t2, M, bias=bias.to(dev), epi=L.EPI_F32_POS, res=pos.to(dev), res_rows=196, tile=tile, a_lo=al, w_lo=wl)
    assert _rel(L.from_blocked(t2, M).cpu(), lin + pos.double()[torch.arange(M) % 196]) < tol


def test_gemm_blk_x3_rejects_half_specified_operands(dev):
    from whmr_amd import _lib as L
    a = torch.randn(64, 64).bfloat16().to(dev)
    ab, wb = L.to_blocked(a), L.to_blocked(torch.randn(256, 64).bfloat16().to(dev))
    out = torch.empty(2, 32, 32, 8, device=dev, dtype=torch.bfloat16)
    p = L.WhmrGemmBlk()
    p.A, p.W, p.C, p.M, p.N, p.K, p.epi = ab.data_ptr(), wb.data_ptr(), out.data_ptr(), 64, 256, 64, 0
    p.A_lo = ab.data_ptr()                                     # W_lo / C_lo missing
    import ctypes
    assert L.lib().whmr_gemm_blk(ctypes.byref(p), None) != 0


@pytest.mark.parametrize('C', [768, 1024])
def test_layernorm_blk_x3(dev, C):
    from whmr_amd import _lib as L
    g = torch.Generator().manual_seed(C)
    rows = 391
    x = torch.randn(rows, C, generator=g) * 3 + 0.5
    w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
    ref = F.layer_norm(x.double(), (C,), w.double(), b.double(), 1e-6)
    xb = L.to_blocked(x.to(dev))
    hi = torch.empty(xb.shape[0], C // 8, 32, 8, device=dev, dtype=torch.bfloat16)
    lo = torch.empty_like(hi)
    L.layernorm_blk_x3(xb, w.to(dev), b.to(dev), hi, lo, rows, 1e-6)
    assert _rel(_join(L, hi, lo, rows), ref) < 1e-5
    # hi is exactly what the bf16 kernel writes
    h16 = torch.empty_like(hi)
    L.layernorm_blk(xb, w.to(dev), b.to(dev), h16, rows, 1e-6)
    assert torch.equal(hi[:rows // 32], h16[:rows // 32])


def test_patch_im2col_blk_x3(dev):
    from whmr_amd import _lib as L
    x = torch.randn(3, 3, 64, 80, generator=torch.Generator().manual_seed(0))[:, :, :, 8:-8]
    M = 3 * 4 * 4
    hi = torch.full(((M + 31) // 32, 768 // 8, 32, 8), float('nan'), device=dev, dtype=torch.bfloat16)
    lo = torch.full_like(hi, float('nan'))
    L.patch_im2col_blk(x.to(dev), hi, 16, 2, out_lo=lo)
    ref = F.unfold(x, 16, padding=2, stride=16).transpose(1, 2).reshape(-1, 768)
    rh, rl = L.split_bf16(ref)
    assert torch.equal(L.from_blocked(hi, M).cpu(), rh) and torch.equal(L.from_blocked(lo, M).cpu(), rl)
    assert not L.from_blocked(lo, 64)[M:].any() and not L.from_blocked(hi, 64)[M:].any()


@pytest.mark.parametrize('N,B', [(196, 3), (192, 3), (100, 3), (256, 3), (65, 2), (208, 2), (196, 64), (192, 27)])
def test_attention_blk_x3(dev, N, B):
    """split-bf16 attention (round 5: the persistent 16-row-tile kernel for N <= 208, csrc/attention_blk16.hip; the round-3 kernel above that)
    against float64 on the same fp32 inputs -- fp32-grade -- and against the round-3 kernel; B = 64 / 27: three / one-or-two items per workgroup
    (the staged prefetch of the next item's K / Q and V)."""
    from whmr_amd import _lib as L
    g = torch.Generator().manual_seed(N)
    H, d = 12, 64
    qkv = torch.randn(B, N, 3, H, d, generator=g) * 1.5
    q, k, v = qkv.double().permute(2, 0, 3, 1, 4)
    ref = ((q * d ** -0.5) @ k.transpose(-2, -1)).softmax(-1) @ v
    ref = ref.transpose(1, 2).reshape(B * N, H * d)
    qh, ql = _pair(L, qkv.view(B * N, 3 * H * d), dev)
    oh = torch.full((qh.shape[0], H * d // 8, 32, 8), float('nan'), device=dev, dtype=torch.bfloat16)
    ol = torch.full_like(oh, float('nan'))
    L.attention_blk(qh, oh, B, N, H, d ** -0.5, qkv_lo=ql, out_lo=ol)
    got = _join(L, oh, ol, B * N)
    err = _rel(got, ref)
    print('x3 attention N=%d B=%d max-rel %.2e' % (N, B, err))
    assert err < 2e-5
    oh2, ol2 = torch.full_like(oh, float('nan')), torch.full_like(oh, float('nan'))
    L.attention_x3_set_variant(1)
    try:
        L.attention_blk(qh, oh2, B, N, H, d ** -0.5, qkv_lo=ql, out_lo=ol2)
    finally:
        L.attention_x3_set_variant(0)
    assert _rel(got, _join(L, oh2, ol2, B * N)) < 2e-5
    oh3, ol3 = torch.full_like(oh, float('nan')), torch.full_like(oh, float('nan'))
    L.attention_blk(qh, oh3, B, N, H, d ** -0.5, qkv_lo=ql, out_lo=ol3)
    assert torch.equal(L.from_blocked(oh3, B * N), L.from_blocked(oh, B * N)) and torch.equal(L.from_blocked(ol3, B * N), L.from_blocked(ol, B * N))     # deterministic


def test_gemm_epilogue_writes_the_split3_operand_form(dev):
    """epi_flags bit 8 of whmr_gemm_bf16 (round 4): an fp32 output also leaves its split-bf16 operand form [hi | lo | hi] -- bit for bit what
    whmr_split3_bf16 makes of that output in a pass of its own -- on every epilogue route the bf16x3 path uses: plain rows with bias + ReLU, the
    residual-before-ReLU route of a ResNet block, the split-K route (few tiles, deep K), a strided 3x3 convolution gather, and the four
    scattered sub-pixel phases of a deconvolution."""
    from whmr_amd import _lib as L
    g = torch.Generator().manual_seed(3)

    def check(out, s3):
        ref = L.split3(out.contiguous())
        assert torch.equal(s3.view(ref.shape), ref)
    # plain rows, M tail inside a tile, bias + ReLU
    M, N, K = 1000, 256, 192
    a = torch.randn(M, K, generator=g).bfloat16().to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).bfloat16().to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    out, s3 = torch.empty(M, N, device=dev), torch.full((M, 3 * N), float('nan'), device=dev, dtype=torch.bfloat16)
    L.gemm(a, w, out, bias=bias, act=L.ACT_RELU, split3_out=s3)
    assert _rel(out.cpu(), F.relu(a.float().cpu() @ w.float().cpu().t() + bias.cpu())) < 1e-5
    check(out, s3)
    # two-part form [hi | lo] (epi_flags bit 9: the operand of the Tz head's N-concatenated first convolution)
    s2 = torch.full((M, 2 * N), float('nan'), device=dev, dtype=torch.bfloat16)
    L.gemm(a, w, out, bias=bias, act=L.ACT_RELU, split3_out=s2, split_parts=2)
    assert torch.equal(s2, s3[:, :2 * N])
    # fp32 skip added before the ReLU (conv3 of a bottleneck block)
    skip = torch.randn(M, N, generator=g).to(dev)
    L.gemm(a, w, out, bias=bias, act=L.ACT_RELU, residual=skip, res_first=True, split3_out=s3)
    assert _rel(out.cpu(), F.relu(a.float().cpu() @ w.float().cpu().t() + bias.cpu() + skip.cpu())) < 1e-5
    check(out, s3)
    # split-K route: 475 x 512 x 4608 (the chooser slices K: tests/test_kernels_gpu.py::test_gemm_bf16_split_k)
    M2, N2, K2 = 475, 512, 4608
    a2 = torch.randn(M2, K2, generator=g).bfloat16().to(dev)
    w2 = (torch.randn(N2, K2, generator=g) / math.sqrt(K2)).bfloat16().to(dev)
    out2, s32 = torch.empty(M2, N2, device=dev), torch.full((M2, 3 * N2), float('nan'), device=dev, dtype=torch.bfloat16)
    L.gemm(a2, w2, out2, act=L.ACT_RELU, split3_out=s32)
    assert _rel(out2.cpu(), F.relu(a2.float().cpu() @ w2.float().cpu().t())) < 2e-5
    check(out2, s32)
    # strided 3x3 convolution gather (NHWC, 64 channels)
    B, H, W, Ci, Co = 2, 11, 9, 64, 128
    x = torch.randn(B, H, W, Ci, generator=g).bfloat16().to(dev)
    wc = (torch.randn(Co, 3, 3, Ci, generator=g) / 24).bfloat16().to(dev)
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y, ys = torch.empty(B, OH, OW, Co, device=dev), torch.full((B, OH, OW, 3 * Co), float('nan'), device=dev, dtype=torch.bfloat16)
    L.gemm(x, wc.view(Co, -1), y, split3_out=ys, conv=dict(IH=H, IW=W, Cin=Ci, OH=OH, OW=OW, KW=3, SH=2, SW=2, PH=1, PW=1))
    refc = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), wc.float().cpu().permute(0, 3, 1, 2), stride=2, padding=1).permute(0, 2, 3, 1)
    assert _rel(y.cpu(), refc) < 1e-5
    check(y, ys)
    # deconvolution as four scattered sub-pixel phases in one launch (whmr.py:488-498; the product's own call, models/whmr.py::_deconv)
    Hd, Wd, Cd = 5, 4, 64
    xd = torch.randn(B, Hd, Wd, Cd, generator=g).bfloat16().to(dev)
    ph = (torch.randn(4, 256, 4 * Cd, generator=g) / 16).bfloat16().to(dev)
    od = torch.full((B, 2 * Hd, 2 * Wd, 256), float('nan'), device=dev)
    ods = torch.full((B, 2 * Hd, 2 * Wd, 768), float('nan'), device=dev, dtype=torch.bfloat16)
    L.gemm(xd, ph, od, act=L.ACT_RELU, split3_out=ods, conv=dict(IH=Hd, IW=Wd, Cin=Cd, OH=Hd, OW=Wd, KW=2, SH=1, SW=1, PH=1, PW=1),
           scatter=dict(c_off=0, osb=4 * Hd * Wd * 256, osy=4 * Wd * 256, osx=2 * 256), phases=dict(cy=2 * Wd * 256, cx=256))
    assert torch.isfinite(od).all()                                     # every output pixel of every phase was written ...
    check(od, ods)                                                      # ... and so was its split form


def _vit(sd, size, dev, numerics, dim=768, depth=12, heads=12):
    from whmr_amd.models.pose_vit import ViT
    m = ViT(img_size=size, patch_size=16, embed_dim=dim, depth=depth, num_heads=heads, ratio=1, mlp_ratio=4, qkv_bias=True, numerics=numerics)
    m.load_state_dict(sd, strict=True)
    return m.to(dev).eval()


def test_vit_x3_matches_reference_fixtures(dev, state_dict):
    """ViT-B/16 in bf16x3 against the fp32 CPU forward of the imported reference: 224^2 (BASELINE configs[1], vit224_b2.npz) and the ref-native
    256x192 (whmr_b2.npz s_feat) within 1e-4 -- max-rel and element-wise; the bf16 mode sits at ~5e-3 on the same fixtures"""
    from oracle import synth
    g = np.load(os.path.join(GOLDEN, 'vit224_b2.npz'))
    sd = synth.make_vit_state(1, (224, 224))
    out = _vit(sd, (224, 224), dev, 'bf16x3')(torch.from_numpy(g['x']).to(dev))
    ref = torch.from_numpy(g['s_feat'])
    e, ew = _rel(out.cpu(), ref), ew_err(out, ref)
    print('bf16x3 ViT-B 224^2 vs reference fixture: max-rel %.2e element-wise %.2e' % (e, ew))
    assert out.shape == (2, 768, 14, 14) and e < 1e-4 and ew < 1e-4
    g2 = np.load(os.path.join(GOLDEN, 'whmr_b2.npz'))
    pre = 'feature_extractor.backbone.'
    sd2 = {k[len(pre):]: v for k, v in state_dict.items() if k.startswith(pre)}
    out2 = _vit(sd2, (256, 192), dev, 'bf16x3')(torch.from_numpy(g2['in_x']).to(dev))
    e2, ew2 = _rel(out2.cpu(), torch.from_numpy(g2['s_feat'])), ew_err(out2, g2['s_feat'])
    print('bf16x3 ViT-B 256x192 vs reference fixture: max-rel %.2e element-wise %.2e' % (e2, ew2))
    assert e2 < 1e-4 and ew2 < 1e-4


def test_vit_x3_full_size_properties_and_fp32_agreement(dev):
    """BASELINE batch-64 size: per-image independence bit for bit (64 == 32 + 32: other tile heights), finite, and the first crops agree with
    the exact-f32 MFMA mode of the same module to 1e-4 (two independent parity-grade implementations of the same arithmetic)"""
    from oracle import synth
    sd = synth.make_vit_state(1, (224, 224))
    m = _vit(sd, (224, 224), dev, 'bf16x3')
    x = synth.make_inputs(64, 3, (224, 224))['x'].to(dev)
    full = m(x)
    assert torch.equal(full, torch.cat([m(x[:32]), m(x[32:])])) and torch.isfinite(full).all()
    ref = _vit(sd, (224, 224), dev, 'fp32')(x[:8])
    assert _rel(full[:8], ref) < 1e-4 and ew_err(full[:8], ref) < 1e-4


def test_vit_large_x3_full_depth_matches_oracle(dev):
    """BASELINE configs[4] backbone at FULL depth (ViT-L/16: dim 1024, depth 24, 16 heads, 256x192) in bf16x3 against the CPU oracle: the
    parity-grade mode on the blocked kernels at ViT-L's shapes (K = 1024 / 4096, N = 3072 / 4096, 16 heads)"""
    from oracle import synth
    from oracle.vit import vit_forward
    sd = synth.make_vit_state(5, (256, 192), embed_dim=1024, depth=24)
    x = synth.make_inputs(2, 13, (256, 192))['x']
    with torch.no_grad():
        ref = vit_forward(sd, x, num_heads=16)
    out = _vit(sd, (256, 192), dev, 'bf16x3', 1024, 24, 16)(x.to(dev))
    e, ew = _rel(out.cpu(), ref), ew_err(out, ref)
    print('bf16x3 ViT-L depth 24 vs oracle: max-rel %.2e element-wise %.2e' % (e, ew))
    assert out.shape == (2, 1024, 16, 12) and e < 1e-4 and ew < 1e-4


def test_vit_x3_layernorm_fold_matches_explicit_passes(dev):
    """bf16x3 with the LayerNorm folded into the GEMM pairs (the default: producers emit the centred row as a hi / lo pair + partial sums, consumers
    normalise in their epilogue) against bf16x3 with explicit fp32 LayerNorm passes and against the exact-f32 mode -- on the fixture weights and on a
    ViT whose tokens sit 30 std off zero (the stream the per-row shift exists for): all within 1e-4, element-wise"""
    from oracle import synth
    g = np.load(os.path.join(GOLDEN, 'vit224_b2.npz'))
    x = torch.from_numpy(g['x']).to(dev)
    base = synth.make_vit_state(1, (224, 224))
    off = {k: v.clone() for k, v in base.items()}
    off['pos_embed'] = off['pos_embed'] + torch.randn(1, off['pos_embed'].shape[1], 1, generator=torch.Generator().manual_seed(1)).sign() * 30.0
    for name, sd in (('fixture weights', base), ('tokens 30 std off zero', off)):
        ref = _vit(sd, (224, 224), dev, 'fp32')(x)
        m = _vit(sd, (224, 224), dev, 'bf16x3')
        assert m.ln_fold
        folded = m(x)
        m.ln_fold = False
        explicit = m(x)
        ef, ee = ew_err(folded, ref), ew_err(explicit, ref)
        print('bf16x3 %s: LayerNorm folded %.2e, explicit %.2e (element-wise vs the exact-f32 mode)' % (name, ef, ee))
        assert ef < 1e-4 and ee < 1e-4 and _rel(folded, ref) < 1e-4
        assert not torch.equal(folded, explicit)                  # two different pipelines did run


def test_vit_large_x3_32_crops_properties(dev):
    """the per-GPU share of BASELINE configs[4] in the parity-grade numerics: ViT-L/16 256x192 at 32 crops (6144 tokens: the 96-row tile, K = 1024 / 4096
    -> 64 / 256 sixteen-deep ring slots) in bf16x3 -- per-image independence bit for bit (32 == 16 + 16) and agreement of the first crops with the
    exact-f32 mode of the same module to 1e-4, element-wise"""
    from oracle import synth
    sd = synth.make_vit_state(5, (256, 192), embed_dim=1024, depth=24)
    m = _vit(sd, (256, 192), dev, 'bf16x3', 1024, 24, 16)
    x = synth.make_inputs(32, 13, (256, 192))['x'].to(dev)
    full = m(x)
    assert torch.equal(full, torch.cat([m(x[:16]), m(x[16:])])) and torch.isfinite(full).all()
    ref = _vit(sd, (256, 192), dev, 'fp32', 1024, 24, 16)(x[:4])
    e, ew = _rel(full[:4], ref), ew_err(full[:4], ref)
    print('bf16x3 ViT-L depth 24, 32 crops vs the exact-f32 mode: max-rel %.2e element-wise %.2e' % (e, ew))
    assert e < 1e-4 and ew < 1e-4


def test_whmr_x3_graph_replay_and_side_streams(dev, assets, state_dict):
    """the full forward in bf16x3 (ViT on the split-operand blocked kernels, deconvs / Tz conv as K-concatenated split operands, fp32 heads): the HIP-graph
    replay equals the eager call, and the side-stream arrangement (cam_model beside the backbone, deconv 2 / 3 + Tz head beside the regressor loop) gives
    the same bits as the in-line order"""
    from whmr_amd.graph import GraphedForward
    from whmr_amd.models import whmr_net
    from oracle import synth
    m = whmr_net(None, assets=assets, numerics='bf16x3')
    m.load_state_dict(state_dict, strict=False)
    m = m.to(dev).eval()
    inp = {k: v.to(dev) for k, v in synth.make_inputs(6, 31).items()}
    full = torch.randn(1, 3, 160, 224, generator=torch.Generator().manual_seed(3)).to(dev)
    args = (inp['x'], None, inp['center'], inp['scale'], inp['bbox_height'], inp['orig_shape'], inp['bbox_info'])
    m.overlap_camera = m.overlap_tz = False
    ref = {k: v.clone() for k, v in m(*args, full_x=full).items()}
    m.overlap_camera = m.overlap_tz = True
    out = m(*args, full_x=full)
    for k in ref:
        assert torch.equal(out[k], ref[k]), k
    g = GraphedForward(m, *args, full_x=full)
    rep = g(*args, full_x=full)
    for k in ref:
        assert torch.allclose(rep[k], ref[k], rtol=1e-5, atol=1e-6), k
