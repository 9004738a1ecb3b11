"""Timing of the deconv pyramid in training mode (3 x ConvT k4s2p1 -> BN(batch stats) -> ReLU; whmr.py:459-501) at batch 64:
forward and backward per stage, HIP events on the current stream.  `python tools/deconv_train_timing.py [B] [numerics]`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from whmr_amd.train.deconv_autograd import deconv_forward_train, deconv_backward


def timed(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        r = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, r


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    dt = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == 'fp32') else torch.bfloat16
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(0)
    H, W, Cin = 16, 12, 768
    x = torch.randn(B, H, W, Cin, generator=g).to(dev).to(dt)
    tot_f = tot_b = 0.0
    for i in range(3):
        w = (torch.randn(Cin, 256, 4, 4, generator=g) * 0.02).to(dev)
        bn = torch.nn.BatchNorm2d(256, momentum=0.1).to(dev)
        tf, (y, saved) = timed(lambda: deconv_forward_train(x, w, bn.weight, bn.bias, bn, dt))
        dy = torch.randn(y.shape, generator=g).to(dev).to(dt)
        tb, _ = timed(lambda: deconv_backward(saved, w, dy, dt, need_dx=True, dx_dtype=torch.float32 if i == 0 else dt))
        flops = 2.0 * B * H * W * Cin * 16 * 256
        print('stage %d  x [%d,%d,%d,%d] -> [%d,%d,%d,256]: forward %.3f ms (%.0f TFLOP/s incl. BN passes), backward %.3f ms (%.0f TFLOP/s)'
              % (i, B, H, W, Cin, B, 2 * H, 2 * W, tf, flops / tf / 1e9, tb, 2 * flops / tb / 1e9))
        tot_f += tf
        tot_b += tb
        x, H, W, Cin = y, 2 * H, 2 * W, 256
    print('deconv pyramid train: forward %.3f ms + backward %.3f ms = %.3f ms at batch %d (%s)' % (tot_f, tot_b, tot_f + tot_b, B, dt))


if __name__ == '__main__':
    main()
