"""Diagnostic: fp32 deconv train chain -- HIP and the CPU fp32 autograd, both measured against a float64 CPU evaluation."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle.train import deconv_bn_relu_train
from whmr_amd.train.deconv_autograd import DeconvBNReLUFn
dev = torch.device('cuda:0')
def rms(a, b): return ((a.double() - b.double()).pow(2).mean().sqrt() / b.double().pow(2).mean().sqrt()).item()
for (B, H, W) in [(4, 8, 6), (8, 16, 12), (2, 32, 24)]:
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, 768, H, W, generator=g)
    ws = [torch.randn(768, 256, 4, 4, generator=g) * 0.02, torch.randn(256, 256, 4, 4, generator=g) * 0.03]
    gam = [torch.rand(256, generator=g) + 0.5 for _ in range(2)]
    bet = [torch.randn(256, generator=g) * 0.2 for _ in range(2)]
    dy = torch.randn(B, 256, 4 * H, 4 * W, generator=g)
    res = {}
    for name, cast in (('f64', torch.float64), ('f32', torch.float32)):
        rx = x.detach().clone().to(cast).requires_grad_(True)
        rp = [[t.detach().clone().to(cast).requires_grad_(True) for t in (ws[i], gam[i], bet[i])] for i in range(2)]
        h = rx
        for i in range(2):
            h = deconv_bn_relu_train(h, rp[i][0], rp[i][1], rp[i][2])
        h.backward(dy.to(cast))
        res[name] = [rx.grad] + [t.grad for r in rp for t in r]
    bns = [torch.nn.BatchNorm2d(256).to(dev) for _ in range(2)]
    dp = [[t.clone().to(dev).requires_grad_(True) for t in (ws[i], gam[i], bet[i])] for i in range(2)]
    xi = x.permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
    hh = xi
    for i in range(2):
        hh = DeconvBNReLUFn.apply(hh, dp[i][0], dp[i][1], dp[i][2], bns[i], torch.float32)
    hh.backward(dy.permute(0, 2, 3, 1).contiguous().to(dev))
    hip = [xi.grad.cpu().permute(0, 3, 1, 2)] + [t.grad.cpu() for r in dp for t in r]
    names = ['dx', 'w0', 'g0', 'b0', 'w1', 'g1', 'b1']
    print((B, H, W), 'rms-rel vs float64:  ' + '  '.join('%s cpu32 %.1e hip %.1e' % (n, rms(a, t), rms(b, t)) for n, a, b, t in zip(names, res['f32'], hip, res['f64'])))
