"""bf16x3 diagnostics: GEMM error vs float64 per epilogue, then the ViT x3 forward launch by launch (sync after each call)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from whmr_amd import _lib as L

dev = torch.device('cuda:0')


def rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max()).item()


def pair(t):
    hi, lo = L.split_bf16(t)
    return L.to_blocked(hi.to(dev)), L.to_blocked(lo.to(dev))


def join(hi, lo, R):
    return L.from_blocked(hi, R).double().cpu() + L.from_blocked(lo, R).double().cpu()


for (M, N, K) in [(392, 768, 768), (1000, 256, 32), (12544, 2304, 768), (3000, 768, 3072)]:
    g = torch.Generator().manual_seed(M)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g)
    (ah, al), (wh, wl) = pair(a), pair(w)
    lin = a.double() @ w.double().t() + bias.double()
    nb = ah.shape[0]
    oh = torch.full((nb, N // 8, 32, 8), float('nan'), device=dev, dtype=torch.bfloat16)
    ol = torch.full_like(oh, float('nan'))
    for tile in (0, 0x44, 0x22):
        L.gemm_blk(ah, wh, oh, M, bias=bias.to(dev), epi=L.EPI_BF16, tile=tile, a_lo=al, w_lo=wl, out_lo=ol)
        torch.cuda.synchronize()
        got = join(oh, ol, M)
        # split-only reference: what an exact product of the split operands would give
        a2 = (L.split_bf16(a)[0].double() + L.split_bf16(a)[1].double())
        w2 = (L.split_bf16(w)[0].double() + L.split_bf16(w)[1].double())
        print('M %d N %d K %d tile %x: vs f64 %.2e, vs f64 of split operands %.2e, hi alone %.2e' % (
            M, N, K, tile, rel(got, lin), rel(got, a2 @ w2.t() + bias.double()), rel(L.from_blocked(oh, M).double().cpu(), lin)), flush=True)
        t = L.to_blocked(torch.zeros(M, N).to(dev))
        L.gemm_blk(ah, wh, t, M, bias=bias.to(dev), epi=L.EPI_F32_RES, res=t, tile=tile, a_lo=al, w_lo=wl)
        torch.cuda.synchronize()
        print('    fp32 epilogue: %.2e' % rel(L.from_blocked(t, M).cpu(), lin), flush=True)

from oracle import synth
from whmr_amd.models.pose_vit import ViT
sd = synth.make_vit_state(1, (224, 224))
m = ViT(img_size=(224, 224), patch_size=16, embed_dim=768, depth=12, num_heads=12, ratio=1, mlp_ratio=4, qkv_bias=True, numerics='bf16x3')
m.load_state_dict(sd, strict=True)
m = m.to(dev).eval()
for name in ('gemm_blk', 'attention_blk', 'layernorm_blk_x3', 'layernorm_blk', 'patch_im2col_blk'):
    fn = getattr(L, name)
    def wrap(*a, _fn=fn, _n=name, **k):
        r = _fn(*a, **k)
        torch.cuda.synchronize()
        print('ok', _n, flush=True)
        return r
    setattr(L, name, wrap)
for B in (2, 64):
    x = synth.make_inputs(B, 7, (224, 224))['x'].to(dev)
    out = m(x)
    torch.cuda.synchronize()
    print('B', B, 'finite', torch.isfinite(out).all().item(), flush=True)
m32 = ViT(img_size=(224, 224), patch_size=16, embed_dim=768, depth=12, num_heads=12, ratio=1, mlp_ratio=4, qkv_bias=True, numerics='fp32')
m32.load_state_dict(sd, strict=True)
x = synth.make_inputs(2, 7, (224, 224))['x'].to(dev)
print('x3 vs fp32 mode:', rel(m(x), m32.to(dev).eval()(x)))
