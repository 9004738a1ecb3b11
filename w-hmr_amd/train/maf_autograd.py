"""MAF sampler in training mode: forward (fp32 point MLP) + hand-written backward.

Reference semantics: autograd of ``MAF_Extractor.sampling`` / ``.forward`` (models/maf_extractor.py:75-143) driven by
``core/trainer.py:410-470``.  The sample points and the camera are constants of the graph (models/whmr.py:586-592 detaches
``markers`` / ``pred_cam``; the iteration-0 grid is a buffer), so the differentiated inputs are the feature map and the three
Conv1d layers.  Backward = one HIP launch (recompute of the 8-point tiles, MLP backward, fp32-atomic scatter of d(f) into the
gradient map, point-minor GEMM operands) + four small fp32 GEMMs for dW / db (K = B*P).  On the last feature map the scatter is deferred to
``MapForkFn`` (per-point records, added to the heads' gradient in place).
"""
import torch

from .. import _lib as L
from ..core.cfgs import cfg
from ..core.constants import FOCAL_LENGTH


def _weights(ext):
    """fp32-only weight pack (no bf16 copies: the training forward runs the fp32 point MLP so that the backward differentiates
    exactly the function that was evaluated)."""
    ps = [ext.conv0.weight, ext.conv0.bias, ext.conv1.weight, ext.conv1.bias, ext.conv2.weight, ext.conv2.bias]
    keep = [ps[0].detach()[:, :, 0].t().contiguous(), ps[1].detach().contiguous(), ps[2].detach()[:, :, 0].t().contiguous(),
            ps[3].detach().contiguous(), ps[4].detach()[:, :, 0].t().contiguous(), ps[5].detach().contiguous()]
    keep += [ps[i].detach()[:, :, 0].contiguous() for i in (0, 2, 4)]           # Conv1d layout [out][in]
    w = L.WhmrMafWeights()
    w.w0t, w.b0, w.w1t, w.b1, w.w2t, w.b2 = [t.data_ptr() for t in keep[:6]]
    return w, keep


class MAFSampleFn(torch.autograd.Function):
    """feat [B, 32*P] = MAFSampleFn.apply(fmap_nchw_view, w0, b0, w1, b1, w2, b2, ext, pts2d, pts3d, cam).

    ``fmap_nchw_view``: logical [B,256,H,W] (channels-last memory; bf16 or fp32); its gradient comes back as an fp32 tensor of the
    same logical shape and channels-last memory."""

    @staticmethod
    def forward(ctx, fmap, w0, b0, w1, b1, w2, b2, ext, pts2d, pts3d, cam, sink=None):
        if not fmap.is_cuda:
            raise RuntimeError('whmr_amd MAF sampler runs on a HIP device only (no CPU fallback)')
        B = fmap.shape[0]
        pts = pts2d if pts2d is not None else pts3d
        P = pts.shape[1]
        w, keep = _weights(ext)
        out = torch.empty(B, 32 * P, dtype=torch.float32, device=fmap.device)
        pts2d = None if pts2d is None else pts2d.detach().float().contiguous()
        pts3d = None if pts3d is None else pts3d.detach().float().contiguous()
        cam = None if cam is None else cam.detach().float().contiguous()
        L.maf_sample(fmap.detach(), w, out, pts2d=pts2d, pts3d=pts3d, cam=cam, focal=FOCAL_LENGTH, res_w=float(cfg.IMG_RES.WIDTH),
                     res_h=float(cfg.IMG_RES.HEIGHT))
        ctx.saved = (fmap.detach(), w, keep, pts2d, pts3d, cam, P)
        ctx.need_map = ctx.needs_input_grad[0]
        ctx.sink = sink if ctx.need_map else None
        return out

    @staticmethod
    def backward(ctx, d_out):
        fmap, w, keep, pts2d, pts3d, cam, P = ctx.saved
        ctx.saved = None
        B, _, H, W = fmap.shape
        dev = fmap.device
        f32 = dict(dtype=torch.float32, device=dev)
        n = B * P
        XT, DT = torch.empty(448, n, **f32), torch.empty(224, n, **f32)
        d_map = record = None
        if ctx.sink is not None:
            # the map's gradient is finished by MapForkFn's backward (below): per-point records now, added to the other consumers' gradient there
            record = torch.empty(n, L.MAF_RECORD, **f32)
        elif ctx.need_map:
            # gradient map in the map's own dtype: a bf16 map is filled directly (CAS-added channel pairs) instead of a zero-filled fp32 map
            # that autograd then casts -- half the fill bytes and no cast pass (0.5 ms per batch-64 step)
            d_map = torch.zeros(B, H, W, 256, dtype=fmap.dtype, device=dev).permute(0, 3, 1, 2)           # channels-last memory, logical NCHW
        L.maf_sample_bwd(fmap, w, keep[6], keep[7], keep[8], d_out.float().contiguous(), d_map, XT, DT, pts2d=pts2d, pts3d=pts3d, cam=cam,
                         focal=FOCAL_LENGTH, res_w=float(cfg.IMG_RES.WIDTH), res_h=float(cfg.IMG_RES.HEIGHT), record=record)
        if record is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            ctx.sink.update(record=record, ready=ev, P=P)
            ctx.sink = None
        dw0 = torch.empty(128, 256, **f32)
        dw1 = torch.empty(64, 384, **f32)
        dw2 = torch.empty(32, 320, **f32)
        L.gemm(DT[0:128], XT[128:384], dw0)
        L.gemm(DT[128:192], XT[0:384], dw1)
        L.gemm(DT[192:224], XT[384:448], dw2[:, :64])
        L.gemm(DT[192:224], XT[128:384], dw2[:, 64:])
        db = torch.empty(224, 1, **f32)
        L.gemm(DT, torch.ones(1, n, **f32), db)
        db = db.view(-1)
        return (d_map, dw0.unsqueeze(-1), db[0:128], dw1.unsqueeze(-1), db[128:192], dw2.unsqueeze(-1), db[192:224], None, None, None, None, None)


class MapForkFn(torch.autograd.Function):
    """(map for the sampler, map for the other consumers) = MapForkFn.apply(fmap_nhwc [B,H,W,256], sink).

    The last feature map feeds the Tz head and the IUV head (large GEMMs on a side stream) and the stage-3 sampler (a scatter of 4 texels per point on
    the main stream).  As plain autograd fan-out that costs a dense zero-filled gradient map for the sampler and a full-size add of the two maps
    (0.07 + 0.2 ms at batch 64, the add on the side stream's critical chain).  Through this node the sampler (``MAFSampleFn(..., sink)``) leaves
    per-point records in ``sink`` and returns no map gradient; this node's backward runs once BOTH branches have run (autograd counts the edge of a
    ``None`` gradient too), on the stream of its forward, and adds the records to the gradient the heads left -- in place."""

    @staticmethod
    def forward(ctx, fmap, sink):
        ctx.set_materialize_grads(False)
        ctx.sink = sink
        ctx.meta = (tuple(fmap.shape), fmap.dtype, fmap.device)
        return fmap.view_as(fmap), fmap.view_as(fmap)

    @staticmethod
    def backward(ctx, g_sampler, g_heads):
        sink, (shape, dt, dev) = ctx.sink, ctx.meta
        ctx.sink = None
        g = g_heads
        if g_sampler is not None:                          # a consumer that is not the deferring sampler
            g = g_sampler if g is None else g + g_sampler
        rec = sink.pop('record', None)
        if rec is None:
            return g, None
        if g is None:
            g = torch.zeros(shape, dtype=dt, device=dev)
        elif not (g.is_contiguous() and g.dtype in (torch.float32, torch.bfloat16)):
            g = g.contiguous().to(dt)
        st = torch.cuda.current_stream(dev)
        st.wait_event(sink.pop('ready'))                   # the records were written on the sampler's stream
        L.maf_scatter(rec, g.permute(0, 3, 1, 2), sink.pop('P'))
        if not torch.cuda.is_current_stream_capturing():
            rec.record_stream(st)                          # allocated on the sampler's stream, read on this one
        return g, None
