"""SyncBatchNorm for the BatchNorm layers W-HMR trains (VERDICT r4 missing #1).

Reference: ``core/trainer.py:83`` -- ``nn.SyncBatchNorm.convert_sync_batchnorm(self.model)`` right before the DistributedDataParallel wrap, so in the
reference's configs[3] step every training-mode BatchNorm normalises with the statistics of the GLOBAL batch (64 x 8 samples), not of the rank's
64.  Only four BatchNorm layers are trained: 3 x ``BatchNorm2d(256)`` of the deconv pyramid (models/whmr.py:497) and the ``BatchNorm1d(1)`` of
the Tz head (models/whmr.py:428); cam_model is frozen (eval mode: running statistics, nothing to exchange).

Protocol (the same as torch's SyncBatchNorm, with ONE packed collective per layer and direction):

  forward   every rank reduces its rows to  [sum z | sum z^2 | rows]  per channel (fp64, ``whmr_bn_sums``)
            all-reduce(SUM) of that [2C + 1] vector
            mean / invstd / the affine pair (a, b) and the running-statistics update from the summed vector (``whmr_bn_stats_from_sums``;
            unbiased variance of the GLOBAL batch, as SyncBatchNorm does); y = relu(z a + b)
  backward  every rank reduces  [sum g | sum g xhat]  over its rows (g = dy gated by the ReLU, xhat from the global statistics);
            dgamma / dbeta are these LOCAL sums (the data-parallel reducer averages parameter gradients afterwards, exactly like DDP does with
            SyncBatchNorm's weight gradients); all-reduce(SUM) of the [2C] vector; dz = a (g - sum_g / N - xhat sum_gx / N) with the global N.

``convert_sync_batchnorm(module, process_group)`` marks the BatchNorm modules (``bn.whmr_sync``); the autograd nodes of this package
(``train/deconv_autograd.py``, ``train/whmr_train.py``) read the mark.  Local BatchNorm stays the default; ``revert_sync_batchnorm`` removes the mark.
With one rank (no process group) the marked layers take the unsplit kernels -- unless ``always=True``, which runs the split pair with a no-op
exchange (the hardware test of the split kernels on a 1-GPU box).

Communicators: the BatchNorm all-reduces are issued synchronously from the autograd thread, the gradient buckets of ``GradReducer``
when a bucket fills -- or, on a rank that lacks one of a bucket's gradients that step, only at ``finish()``.  On ONE communicator two ranks could
then issue the two kinds of collectives in different orders (gloo aborts on the size mismatch, RCCL pairs unrelated buffers or hangs).  So
``convert_sync_batchnorm`` gives the BatchNorm traffic its OWN group by default (``dist.new_group()`` over the ranks of ``process_group``; every
rank must make the call, like any group creation).  Under torch's own ``DistributedDataParallel`` + ``nn.SyncBatchNorm`` (the reference's two
lines, core/trainer.py:83-86, supported as they are: ``sync_of`` honours an ``nn.SyncBatchNorm`` module) both kinds of traffic share the
module's ``process_group`` exactly as they do in torch itself: DDP launches its buckets in bucket order from the same autograd thread and
requires the same graph on every rank, so the program order is the same everywhere.
"""
import torch
import torch.distributed as dist
import torch.nn as nn

from .. import _lib as L


class SyncGroup:
    """the mark ``convert_sync_batchnorm`` leaves on a BatchNorm module"""

    def __init__(self, process_group=None, always=False):
        self.group = process_group
        self.always = always
        self.collectives = 0                       # bookkeeping for the bench line / tests
        self.bytes = 0

    def world(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def active(self):
        return self.always or self.world() > 1

    def all_reduce(self, t):
        """sum ``t`` (fp64 vector) over the ranks in place; one collective"""
        if dist.is_initialized() and (dist.get_world_size(self.group) > 1 or self.always):      # always: also with ONE rank (the hardware run of the RCCL call on a 1-GPU box)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            self.collectives += 1
            self.bytes += t.numel() * t.element_size()
        return t


def convert_sync_batchnorm(module, process_group=None, always=False, dedicated=True):
    """``nn.SyncBatchNorm.convert_sync_batchnorm`` (core/trainer.py:83) for this package's modules: every BatchNorm under ``module`` shares one
    ``SyncGroup`` mark; its training-mode statistics are then taken over all ranks of ``process_group`` (None = the default group).  Returns the
    module (same object: the parameters / buffers / state_dict keys do not change, unlike torch's conversion which swaps the module class).
    ``dedicated`` (default): at world size > 1 the exchanges run on a NEW group over the same ranks, so that they can never interleave with the
    gradient buckets of ``GradReducer`` on one communicator (module docstring) -- a collective call: every rank converts its model."""
    if dedicated and dist.is_initialized() and dist.get_world_size(process_group) > 1:
        process_group = dist.new_group(ranks=dist.get_process_group_ranks(process_group) if process_group is not None else None)
    sg = SyncGroup(process_group, always)
    n = 0
    for m in module.modules():
        if isinstance(m, nn.modules.batchnorm._BatchNorm):
            m.whmr_sync = sg
            n += 1
    module.whmr_sync_group = sg
    module.whmr_sync_layers = n
    return module


def revert_sync_batchnorm(module):
    for m in module.modules():
        if hasattr(m, 'whmr_sync'):
            del m.whmr_sync
    for k in ('whmr_sync_group', 'whmr_sync_layers'):
        if hasattr(module, k):
            delattr(module, k)
    return module


ALWAYS_SPLIT = False          # tests on a 1-GPU box: SyncGroups made for torch's own nn.SyncBatchNorm modules run the split kernels / the collective at world 1
_AUTO_GROUPS = {}             # process group (or None) -> the SyncGroup shared by every torch nn.SyncBatchNorm module on it


def auto_sync_groups():
    """the SyncGroups ``sync_of`` created for torch ``nn.SyncBatchNorm`` modules (bookkeeping: collectives / bytes per group)"""
    return list(_AUTO_GROUPS.values())


def sync_of(bn):
    """the active SyncGroup of a BatchNorm module, else None.  A module that torch's own ``nn.SyncBatchNorm.convert_sync_batchnorm`` swapped in
    (the reference's line, core/trainer.py:83, applied to this package's model) counts as marked with its ``process_group``; all such modules on
    one group share one SyncGroup (tests/test_train_gpu.py::test_reference_syncbn_ddp_wrap_world1_rccl runs exactly those two lines)."""
    sg = getattr(bn, 'whmr_sync', None)
    if sg is None and isinstance(bn, nn.SyncBatchNorm):
        sg = _AUTO_GROUPS.get(bn.process_group)
        if sg is None:
            sg = _AUTO_GROUPS[bn.process_group] = SyncGroup(bn.process_group, always=ALWAYS_SPLIT)
    return sg if (sg is not None and sg.active()) else None


# ---- channels-last BatchNorm + ReLU of the deconv stages (HIP kernels; the protocol above) ----------------------------------------------
@torch.no_grad()
def bn_relu_forward(z2, gamma, beta, bn, track, sg):
    """z2 [M, C] (this rank's rows) -> (y2, stats [4, C], count): statistics over ALL ranks' rows.  ``count``: 1-element fp64 tensor, the global
    row count, kept on the device for the backward."""
    sums = L.bn_sums(z2)
    sg.all_reduce(sums)
    stats = L.bn_stats_from_sums(sums, gamma, beta, bn.eps, bn.momentum if track else 0.0, bn.running_mean if track else None,
                                 bn.running_var if track else None)
    y2 = torch.empty_like(z2)
    L.bn_apply_relu(z2, stats, y2)
    return y2, stats, sums[-1:]


@torch.no_grad()
def bn_relu_backward(z2, dy2, stats, count, dz2, dgamma, dbeta, sg):
    """dz2 (this rank's rows), dgamma / dbeta (LOCAL sums) for the upstream gradient dy2 of relu(bn(z2)) under global statistics"""
    sums = L.bn_bwd_sums(z2, dy2, stats, dgamma, dbeta)
    sg.all_reduce(sums)
    L.bn_bwd_apply(z2, dy2, stats, sums, count, dz2)
    return dz2


# ---- BatchNorm1d of the Tz head: [B, C] with C = 1 -- O(batch) tensor arithmetic, any device -----------------------------------------------
class SyncBatchNorm1dFn(torch.autograd.Function):
    """y = SyncBatchNorm1dFn.apply(x [B, C], weight, bias, bn, sync_group): training-mode BatchNorm1d over all ranks' rows (whmr.py:428 under
    core/trainer.py:83).  Plain tensor arithmetic in fp64 on a [B, 1] input -- the same two packed all-reduces as the channels-last kernels."""

    @staticmethod
    def forward(ctx, x, weight, bias, bn, sg):
        xd = x.detach().double()
        C = xd.shape[1]
        sums = torch.cat([xd.sum(0), (xd * xd).sum(0), xd.new_full((1,), float(xd.shape[0]))])
        sg.all_reduce(sums)
        n = sums[-1]
        mean = sums[:C] / n
        var = (sums[C:2 * C] / n - mean * mean).clamp_min(0.0)
        invstd = torch.rsqrt(var + bn.eps)
        xhat = (xd - mean) * invstd
        w = weight.detach().double() if weight is not None else torch.ones_like(mean)
        b = bias.detach().double() if bias is not None else torch.zeros_like(mean)
        y = (xhat * w + b).to(x.dtype)
        if bn.training and bn.track_running_stats and bn.running_mean is not None:
            mom = bn.momentum
            assert mom is not None, 'cumulative-average BatchNorm (momentum=None) is not used by W-HMR'
            unb = var * (n / torch.clamp(n - 1.0, min=1.0))
            bn.running_mean.mul_(1.0 - mom).add_((mom * mean).to(bn.running_mean.dtype))
            bn.running_var.mul_(1.0 - mom).add_((mom * unb).to(bn.running_var.dtype))
            if bn.num_batches_tracked is not None:
                bn.num_batches_tracked += 1
        ctx.save_for_backward(xhat, invstd, w, n)
        ctx.sg = sg
        ctx.has = (weight is not None, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        xhat, invstd, w, n = ctx.saved_tensors
        g = dy.double()
        C = g.shape[1]
        sums = torch.cat([g.sum(0), (g * xhat).sum(0)])
        dweight = sums[C:].clone().to(dy.dtype) if ctx.has[0] else None        # LOCAL sums: the parameter gradients stay per rank (see module docstring)
        dbias = sums[:C].clone().to(dy.dtype) if ctx.has[1] else None
        ctx.sg.all_reduce(sums)
        dx = w * invstd * (g - sums[:C] / n - xhat * (sums[C:] / n))
        return dx.to(dy.dtype), dweight, dbias, None, None


def batch_norm_1d(x, bn):
    """``bn(x)`` for the Tz head's BatchNorm1d: the module itself unless it is marked for cross-rank statistics and in training mode"""
    sg = sync_of(bn) if bn.training else None
    if sg is None:
        return bn(x)
    return SyncBatchNorm1dFn.apply(x, bn.weight, bn.bias, bn, sg)
