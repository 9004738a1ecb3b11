"""Diagnostic: error of the two-stage deconv train chain vs CPU autograd at several sizes, both numerics."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle.train import deconv_bn_relu_train
from whmr_amd.train.deconv_autograd import DeconvBNReLUFn
dev = torch.device('cuda:0')
def rel(a, b): return ((a.double() - b.double()).abs().max() / b.double().abs().max()).item()
def rms(a, b): return ((a.double() - b.double()).pow(2).mean().sqrt() / b.double().pow(2).mean().sqrt()).item()
for (B, H, W) in [(3, 4, 3), (4, 8, 6), (8, 16, 12)]:
    for dt in (torch.float32, torch.bfloat16):
        g = torch.Generator().manual_seed(11)
        x = torch.randn(B, 768, H, W, generator=g)
        ws = [torch.randn(768, 256, 4, 4, generator=g) * 0.02, torch.randn(256, 256, 4, 4, generator=g) * 0.03]
        gam = [torch.rand(256, generator=g) + 0.5 for _ in range(2)]
        bet = [torch.randn(256, generator=g) * 0.2 for _ in range(2)]
        dy = torch.randn(B, 256, 4 * H, 4 * W, generator=g)
        rx = x.clone().requires_grad_(True)
        rp = [[t.clone().requires_grad_(True) for t in (ws[i], gam[i], bet[i])] for i in range(2)]
        h = rx
        for i in range(2):
            h = deconv_bn_relu_train(h, rp[i][0], rp[i][1], rp[i][2])
        h.backward(dy)
        bns = [torch.nn.BatchNorm2d(256).to(dev) for _ in range(2)]
        dp = [[t.clone().to(dev).requires_grad_(True) for t in (ws[i], gam[i], bet[i])] for i in range(2)]
        xi = x.permute(0, 2, 3, 1).contiguous().to(dev).to(dt).requires_grad_(True)
        hh = xi
        for i in range(2):
            hh = DeconvBNReLUFn.apply(hh, dp[i][0], dp[i][1], dp[i][2], bns[i], dt)
        hh.backward(dy.permute(0, 2, 3, 1).contiguous().to(dev).to(dt))
        print((B, H, W), dt, 'y %.2e' % rel(hh.detach().float().cpu().permute(0, 3, 1, 2), h.detach()),
              ' '.join('%s%d %.2e/%.2e' % (n, i, rel(a.grad.float().cpu(), b.grad), rms(a.grad.float().cpu(), b.grad))
                       for i in range(2) for a, b, n in zip(dp[i], rp[i], 'wgb')),
              'dx %.2e/%.2e' % (rel(xi.grad.float().cpu().permute(0, 3, 1, 2), rx.grad), rms(xi.grad.float().cpu().permute(0, 3, 1, 2), rx.grad)))
