"""Whole-training-step HIP graph: capture forward + loss + backward once, replay it per iteration.

A W-HMR training step is ~1750 kernel launches of which ~1100 are short head kernels (regressor Linear nodes, SMPL, sampler, BatchNorm
finalisers ...); issued from Python they leave the GPU idle for ~4.5 ms of a 34 ms step at batch 64.  Every launcher of this library only
enqueues on the current stream and allocates nothing (include/whmr_hip.h), torch's allocator serves the intermediate tensors from the
graph's private pool, so the step is capturable as it stands.  Rules the caller keeps (the usual whole-network-capture rules):
static input tensors (copy new batches INTO them), no host reads inside ``fn``, gradients are read from ``p.grad`` after ``replay()``
and the optimizer runs outside the graph.

While a capture is running whmr_forward_train keeps to two streams and to autograd's own order of issue: with the Tz head's tail on a third stream
hipStreamEndCapture crashed, and with the side stream's nodes raised in the ready queue (whmr_train._backward_first) the replayed graph ran into a GPU
memory fault (round 6, one box each, not diagnosed; both are host-side overlap measures that a replayed graph does not need -- it has no host in the loop).
"""
import torch


def capture_train_step(model, fn, warmup=3):
    """fn() -> loss runs forward + backward with ``p.grad = None`` first.  Returns (replay, loss_tensor): ``replay()`` re-executes the step;
    ``loss_tensor`` and every ``p.grad`` are the static outputs.

    The eager warm-up runs on a side stream: the parameters' AccumulateGrad nodes must not belong to the legacy default stream when the
    capture starts (gradient hooks, e.g. GradReducer's, keep those nodes alive -- create the reducer after the capture or not at all on one
    GPU).  The ViT's cached bf16 weight copies are dropped so that the casts are recorded into the graph: a real run changes the weights
    between replays."""
    vit = getattr(getattr(model, 'feature_extractor', None), 'backbone', None)
    from .. import _lib as L
    side = L.side_stream(torch.cuda.current_device(), 2)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warmup):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    if vit is not None and hasattr(vit, '_wcache'):
        vit._wcache.clear()
    if vit is not None and vit.__dict__.get('_train_operands') is not None:
        vit.__dict__['_train_operands'].versions = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        loss = fn()
    return graph.replay, loss
