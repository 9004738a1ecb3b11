"""Dense-correspondence (IUV) auxiliary supervision of the training step: ground truth from the HIP rasteriser + the losses.

Reference: core/trainer.py:442-464 renders the fitted mesh to an IUV image with pytorch3d on every step (``IUV_Renderer.verts2iuvimg``), crops
it to the ViTPose feature-map width (:454-455), turns it into target maps (``iuv_img2map``) and applies ``body_uv_losses`` (:255-298) to the
``dp_head`` outputs.  Here the image comes from ``whmr_amd.utils.renderer.IUV_Renderer`` (csrc/rasterize.hip: three launches, no host stall).
The losses have two forms with the same values: ``body_uv_losses`` keeps the reference's signature (target maps in, O(B x 25 x 128 x 96) tensor
arithmetic through torch autograd: ~2 ms of a batch-64 step), and ``IUVLossFn`` is the training step's path -- csrc/iuv_loss.hip reads the IUV
head's channels-last logits and the rendered image once per direction and hands the convolution's backward its padded gradient operand.
"""
import torch
import torch.nn.functional as F

from .. import _lib as L
from ..core.cfgs import cfg
from ..utils.iuvmap import iuv_img2map


def gt_camera_from_translation(cam_t, focal_length=5000.0, img_res=None):
    """core/trainer.py:443-446: weak-perspective (s, tx, ty) of the fitted camera translation"""
    img_res = float(cfg.IMG_RES.HEIGHT if img_res is None else img_res)
    cam = torch.zeros_like(cam_t)
    cam[:, 1:] = cam_t[:, :2]
    cam[:, 0] = (2.0 * focal_length / img_res) / cam_t[:, 2]
    return cam


@torch.no_grad()
def render_iuv_targets(iuv_maker, verts, gt_camera, vitpose_crop=True, maps=True):
    """-> (iuv_image_gt [B,3,H,W'], (Umap, Vmap, Imap, Annmap)) as core/trainer.py:447-464 builds them (valid_fit = all); ``maps=False`` skips the
    target maps (None): the fused losses read the image itself."""
    img = iuv_maker.verts2iuvimg(verts, cam=gt_camera)
    if vitpose_crop:
        img = img[:, :, :, 16:-16]                                      # trainer.py:454-455 (cfg.MODEL.PyMAF.BACKBONE == 'vitpose')
    return img, (iuv_img2map(img) if maps else None)


class IUVHeadOutput(dict):
    """One ``dp_out`` entry of the training forward.  The reference's four NCHW maps (predict_u / predict_v / predict_uv_index / predict_ann_index,
    models/iuv_predictor.py:86-91) are channel slices of ONE channels-last tensor ``nhwc`` [B, H, W, 90] -- the IUV head runs as a single implicit
    GEMM.  The fp32 NCHW views are created on first access: the fused loss path reads ``nhwc`` and never pays for them."""
    KEYS = ('predict_uv_index', 'predict_ann_index', 'predict_u', 'predict_v')

    def __init__(self, nhwc, sizes=(25, 25, 25, 15)):
        super().__init__()
        self.nhwc, self._sizes = nhwc, tuple(sizes)

    def _fill(self):
        if not dict.__len__(self):
            u, v, idx, ann = torch.split(self.nhwc.float(), self._sizes, dim=-1)
            for k, t in zip(self.KEYS, (idx, ann, u, v)):
                dict.__setitem__(self, k, t.permute(0, 3, 1, 2))

    def __getitem__(self, k):
        self._fill()
        return dict.__getitem__(self, k)

    def get(self, k, default=None):
        self._fill()
        return dict.get(self, k, default)

    def __contains__(self, k):
        return k in self.KEYS

    def __iter__(self):
        return iter(self.KEYS)

    def __len__(self):
        return len(self.KEYS)

    def keys(self):
        self._fill()
        return dict.keys(self)

    def values(self):
        self._fill()
        return dict.values(self)

    def items(self):
        self._fill()
        return dict.items(self)


class IUVLossFn(torch.autograd.Function):
    """(loss_U, loss_V, loss_IndexUV, loss_segAnn) [4] = IUVLossFn.apply(nhwc logits [B, H, W, 90], iuv_image_gt [B, 3, H, W], point_weight): the
    values of ``body_uv_losses(..., iuv_img2map(iuv_image_gt))`` (core/trainer.py:255-298 on utils/iuvmap.py:67-110 targets) from one pass of
    csrc/iuv_loss.hip; the backward is a second pass that writes the gradient as the zero-padded [B*H*W, ld] matrix ConvNHWCFn's backward uses as is."""

    @staticmethod
    def forward(ctx, y, iuv, point_weight):
        if not y.is_cuda:
            raise RuntimeError('whmr_amd runs on a HIP device only (no CPU fallback)')
        y, iuv = y.detach(), iuv.detach().float()
        ctx.saved, ctx.w = (y, iuv), float(point_weight)
        return L.iuv_losses(y, iuv, point_weight)

    @staticmethod
    def backward(ctx, g):
        y, iuv = ctx.saved
        ctx.saved = None
        B, H, W, Cc = y.shape
        ld = y.stride(2)
        dyp = L.iuv_losses_bwd(y, iuv, ctx.w, g, ld)
        # dy is a VIEW of the padded [B*H*W, ld] operand: ConvNHWCFn.backward finds the whole buffer through dy._base.  The tag is the producer's
        # explicit promise "laid out [M, ld], columns [Cc, ld) are zero" (csrc/iuv_loss.hip writes them) -- heads_autograd._padded_base takes nothing else
        dyp.whmr_zero_padded = ld
        return dyp.view(B, H, W, ld)[..., :Cc], None, None


def body_uv_losses(u_pred, v_pred, index_pred, ann_pred, uvia_list):
    """core/trainer.py:255-298 (has_iuv = None): cross entropy on the 25-way part index and the 15-way annotation index, smooth-L1 on U / V
    inside the body, the latter two summed over pixels / batch size and weighted by cfg.LOSS.POINT_REGRESSION_WEIGHTS."""
    Umap, Vmap, Imap, Annmap = uvia_list
    B = index_pred.size(0)
    # F.cross_entropy on [B, C, H, W] logits / [B, H, W] targets == the reference's permute + view(-1, C) form (mean over all pixels)
    loss_index = F.cross_entropy(index_pred, torch.argmax(Imap, dim=1))
    w = float(cfg.LOSS.POINT_REGRESSION_WEIGHTS)
    if w > 0:
        # the reference gathers u_pred[Imap > 0] (a host-synchronising boolean index); multiplying by the 0/1 mask gives the same sum -- masked-out
        # entries are 0 on both sides and smooth_l1(0, 0) = 0 -- without leaving the stream (the step stays graph-capturable)
        fg = (Imap > 0).to(u_pred.dtype)
        loss_u = F.smooth_l1_loss(u_pred * fg, Umap, reduction='sum') / B * w
        loss_v = F.smooth_l1_loss(v_pred * fg, Vmap, reduction='sum') / B * w
    else:
        loss_u = loss_v = torch.zeros((), device=index_pred.device)
    loss_ann = None
    if ann_pred is not None:
        loss_ann = F.cross_entropy(ann_pred, torch.argmax(Annmap, dim=1))
    return loss_u, loss_v, loss_index, loss_ann


def aux_supervision_loss(dp_out, uvia_list, iuv_image_gt=None):
    """sum over the dp_head outputs of the four terms (trainer.py:466-482; equal map sizes -- the yaml default).  With the rendered image at hand
    and a ``dp_out`` entry of the training forward (``IUVHeadOutput``) the fused kernel runs; otherwise the map form (``uvia_list`` may then be
    None when the image is given)."""
    total = 0.0
    for d in dp_out:
        y = getattr(d, 'nhwc', None)
        if (iuv_image_gt is not None and y is not None and y.is_cuda and y.shape[-1] == 90
                and tuple(iuv_image_gt.shape) == (y.shape[0], 3, y.shape[1], y.shape[2])):
            total = total + IUVLossFn.apply(y, iuv_image_gt, float(cfg.LOSS.POINT_REGRESSION_WEIGHTS)).sum()
            continue
        if uvia_list is None:
            uvia_list = iuv_img2map(iuv_image_gt)
        lu, lv, li, la = body_uv_losses(d['predict_u'], d['predict_v'], d['predict_uv_index'], d['predict_ann_index'], uvia_list)
        total = total + lu + lv + li + (la if la is not None else 0.0)
    return total
