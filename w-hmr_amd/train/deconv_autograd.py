"""Training mode of one deconv stage: ConvTranspose2d(k4, s2, p1, no bias) -> BatchNorm2d(batch statistics) -> ReLU.

Reference semantics: the ``nn.Sequential`` built at models/whmr.py:459-501 and run at :560-564 in ``model.train()``, and its
autograd as driven by ``core/trainer.py:410-470``.  Maps are channels-last (NHWC) like the inference path:

  forward   z = sub-pixel GEMM (4 phases, one launch in bf16 mode; no BN folding, the statistics come from z itself)
            stats = whmr_bn_stats(z) (+ running-stat update, momentum 0.1, unbiased variance)     y = relu(z*a + b)
  backward  dz, dgamma, dbeta = whmr_bn_relu_bwd(z, dy)               (mask and xhat recomputed from the saved z)
            dx = Conv2d(k4, s2, p1) of dz with W[ci, (ky,kx,co)]      (implicit GEMM, NHWC gather, K = 16*Cout)
            dW = X^T [Cin, M] . col(dz)^T [(ky,kx,co), M]             (whmr_transpose_cast + whmr_im2col_t, K = M = B*H*W)

``dt`` = torch.bfloat16 (perf numerics: bf16 maps and GEMM operands, fp32 statistics / parameter gradients) or torch.float32
(parity numerics against the CPU reference's autograd).
"""
import torch

import os

from .. import _lib as L
from ..parallel import sync_bn

USE_TN = os.environ.get('WHMR_TN_GEMM', '1') != '0'      # weight gradients on the gathering TN kernel (A/B switch)


_PHASE_INDEX = {}


def _phase_weights(w, dt):
    """ConvTranspose2d weight [Cin, Cout, 4, 4] -> 4 sub-pixel phase matrices [4, Cout, 4*Cin], k = (a, b, ci) (whmr.py:488-495):
    wp[2 py + px, co, (a, b, ci)] = w[ci, co, 3 - py - 2a, 3 - px - 2b].  ONE gather (cached index arrays) + the cast -- the weights change every
    step, and the stack-of-slices form was ~20 launches per deconv."""
    Cout = w.shape[1]
    key = (w.device, Cout)
    idx = _PHASE_INDEX.get(key)
    if idx is None:
        two = torch.arange(2, device=w.device)
        ky = (3 - two.view(2, 1, 1, 1, 1) - 2 * two.view(1, 1, 1, 2, 1)).expand(2, 2, Cout, 2, 2)      # [py, px, co, a, b]
        kx = (3 - two.view(1, 2, 1, 1, 1) - 2 * two.view(1, 1, 1, 1, 2)).expand(2, 2, Cout, 2, 2)
        co = torch.arange(Cout, device=w.device).view(1, 1, Cout, 1, 1).expand(2, 2, Cout, 2, 2)
        idx = _PHASE_INDEX[key] = (ky.contiguous(), kx.contiguous(), co.contiguous())
    p = w.permute(2, 3, 1, 0)[idx].reshape(4, Cout, -1)             # [py, px, co, a, b, ci] -> [4, Cout, 4 * Cin]
    return L.cast_bf16(p) if dt == torch.bfloat16 else p


@torch.no_grad()
def deconv_forward_train(x_nhwc, weight, gamma, beta, bn, dt, update_running=True):
    """x [B,H,W,Cin] (dt) -> (y [B,2H,2W,Cout] (dt), saved).  ``bn``: the nn.BatchNorm2d whose eps / momentum / running stats are used."""
    if not x_nhwc.is_cuda:
        raise RuntimeError('whmr_amd deconv stages run on a HIP device only (no CPU fallback)')
    B, H, W, Cin = x_nhwc.shape
    Cout = weight.shape[1]
    assert weight.shape == (Cin, Cout, 4, 4) and x_nhwc.dtype == dt and x_nhwc.is_contiguous()
    dev = x_nhwc.device
    wp = _phase_weights(weight.detach().float(), dt)
    z = torch.empty(B, 2 * H, 2 * W, Cout, dtype=dt, device=dev)
    if dt == torch.bfloat16:
        L.gemm(x_nhwc, wp, z, conv=dict(IH=H, IW=W, Cin=Cin, OH=H, OW=W, KW=2, SH=1, SW=1, PH=1, PW=1),
               scatter=dict(c_off=0, osb=4 * H * W * Cout, osy=4 * W * Cout, osx=2 * Cout), phases=dict(cy=2 * W * Cout, cx=Cout))
    else:
        for py in range(2):
            for px in range(2):
                L.gemm(x_nhwc, wp[py * 2 + px], z, conv=dict(IH=H, IW=W, Cin=Cin, OH=H, OW=W, KW=2, SH=1, SW=1, PH=1 - py, PW=1 - px),
                       scatter=dict(c_off=(py * 2 * W + px) * Cout, osb=4 * H * W * Cout, osy=4 * W * Cout, osx=2 * Cout))
    z2 = z.view(-1, Cout)
    track = update_running and bn.track_running_stats and bn.running_mean is not None
    assert not track or bn.momentum is not None, 'cumulative-average BatchNorm (momentum=None) is not used by W-HMR (BN_MOMENTUM = 0.1)'
    sg = sync_bn.sync_of(bn)
    if sg is not None:
        # SyncBatchNorm (core/trainer.py:83): the statistics of ALL ranks' rows -- one packed fp64 all-reduce between the two halves of bn_stats
        y2, stats, count = sync_bn.bn_relu_forward(z2, gamma.detach(), beta.detach(), bn, track, sg)
        if track and bn.num_batches_tracked is not None:
            bn.num_batches_tracked += 1
        return y2.view_as(z), (x_nhwc, z, stats, sg, count)
    stats = L.bn_stats(z2, gamma.detach(), beta.detach(), bn.eps, bn.momentum if track else 0.0,
                       bn.running_mean if track else None, bn.running_var if track else None)
    if track and bn.num_batches_tracked is not None:
        bn.num_batches_tracked += 1
    y = torch.empty_like(z)
    L.bn_apply_relu(z2, stats, y.view(-1, Cout))
    return y, (x_nhwc, z, stats, None, None)


@torch.no_grad()
def deconv_backward(saved, weight, dy, dt, need_dx=True, dx_dtype=None, dx_into=None):
    """dy [B,2H,2W,Cout] (dt or fp32) -> (dx [B,H,W,Cin] or None, dW [Cin,Cout,4,4] fp32, dgamma [Cout], dbeta [Cout]).
    ``dx_into`` [B,H,W,Cin] (contiguous, the dx dtype): the data gradient is ADDED to it in the GEMM epilogue and it is returned as dx."""
    x, z, stats, sg, count = saved
    B, H, W, Cin = x.shape
    Cout = z.shape[-1]
    dev = x.device
    dy = dy.contiguous()
    if dy.dtype != dt and not (dt == torch.bfloat16 and dy.dtype == torch.float32):
        dy = dy.to(dt)
    dz = torch.empty_like(z)
    dg, db = torch.empty(Cout, dtype=torch.float32, device=dev), torch.empty(Cout, dtype=torch.float32, device=dev)
    if sg is not None:
        sync_bn.bn_relu_backward(z.view(-1, Cout), dy.view(-1, Cout), stats, count, dz.view(-1, Cout), dg, db, sg)
    else:
        L.bn_relu_bwd(z.view(-1, Cout), dy.view(-1, Cout), stats, dz.view(-1, Cout), dg, db)
    M = B * H * W
    # dW[ci, (ky,kx,co)] = sum_m x[m, ci] * dz[b, 2iy-1+ky, 2ix-1+kx, co]
    pad = 64 if dt == torch.bfloat16 else 8          # the bf16 GEMM needs K % 64 == 0; rows are zero-padded up to it
    dwm = torch.empty(Cin, 16 * Cout, dtype=torch.float32, device=dev)
    x2 = x.view(M, Cin)
    if USE_TN and L.conv_dw_tn_ok(x2, dz):
        # the gathering TN kernel reads X and dZ as they are (no transposed X, no 16-tap transposed column matrix of dZ: 1.6 GB at stage 3)
        L.conv_dw_tn(x2, dz, dwm, H, W, 4, 4, 2, 1)
    else:
        xt = L.transpose_cast(x2, dt, pad_to=pad)                                          # [Cin, Mpad]
        colt = L.im2col_t(dz, H, W, 4, 4, 2, 1, pad_to=pad)                                # [16*Cout, Mpad]
        L.gemm(xt, colt, dwm)
        del colt, xt
    dW = dwm.view(Cin, 4, 4, Cout).permute(0, 3, 1, 2)
    dx = None
    if need_dx:
        wd = weight.detach().float().permute(0, 2, 3, 1).reshape(Cin, 16 * Cout).contiguous()       # [ci, (ky,kx,co)]
        if dt == torch.bfloat16:
            wd = L.cast_bf16(wd)
        acc = dx_into is not None and dx_into.dtype == dt and dx_into.is_contiguous() and tuple(dx_into.shape) == (B, H, W, Cin)
        dx = dx_into if acc else torch.empty(B, H, W, Cin, dtype=dx_dtype or dt, device=dev)
        L.gemm(dz, wd, dx.view(M, Cin), conv=dict(IH=2 * H, IW=2 * W, Cin=Cout, OH=H, OW=W, KW=4, SH=2, SW=2, PH=1, PW=1), accumulate=acc)
        if dx_into is not None and not acc:
            dx = dx + dx_into.to(dx.dtype)
    elif dx_into is not None:
        dx = dx_into
    return dx, dW, dg, db


class DeconvBNReLUFn(torch.autograd.Function):
    """y = DeconvBNReLUFn.apply(x_nhwc, ct.weight, bn.weight, bn.bias, bn, dt): autograd node of one deconv stage.
    ``passthrough=True`` -> (y, x): the input map is handed on to its other consumer (the MAF sampler of that stage), whose gradient then arrives
    here and receives this stage's data gradient in the GEMM epilogue -- no zero-filled map per consumer, no full-size add by autograd."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, bn, dt, passthrough=False):
        ctx.set_materialize_grads(False)
        y, saved = deconv_forward_train(x, weight, gamma, beta, bn, dt)
        ctx.saved, ctx.weight, ctx.dt = saved, weight, dt
        ctx.need_dx = ctx.needs_input_grad[0]
        return (y, x.detach().view_as(x)) if passthrough else y

    @staticmethod
    def backward(ctx, dy, dx_in=None):
        if dy is None:
            return dx_in, None, None, None, None, None, None
        dx, dW, dg, db = deconv_backward(ctx.saved, ctx.weight, dy, ctx.dt, need_dx=ctx.need_dx, dx_into=dx_in if ctx.need_dx else None)
        ctx.saved = None
        return dx, dW, dg, db, None, None, None
