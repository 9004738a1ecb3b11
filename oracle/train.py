"""CPU oracle for the training-mode stages (TEST INFRASTRUCTURE ONLY: imported by tests/, never by w-hmr_amd/).

Plain PyTorch fp32 restatements whose autograd is the reference gradient.  Parity: these are the torch modules the reference
itself instantiates (nn.ConvTranspose2d / nn.BatchNorm2d / nn.ReLU at models/whmr.py:488-498), called functionally.
"""
import torch
import torch.nn.functional as F


def deconv_bn_relu_train(x, weight, gamma, beta, running_mean=None, running_var=None, eps=1e-5, momentum=0.1):
    """models/whmr.py:488-498 in train mode: ConvTranspose2d(k4, s2, p1, output_padding 0, no bias) -> BatchNorm2d -> ReLU.
    x NCHW; running stats (if given) are updated in place like nn.BatchNorm2d does."""
    z = F.conv_transpose2d(x, weight, None, stride=2, padding=1, output_padding=0)
    z = F.batch_norm(z, running_mean, running_var, gamma, beta, training=True, momentum=momentum, eps=eps)
    return F.relu(z)
