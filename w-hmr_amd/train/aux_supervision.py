"""Dense-correspondence (IUV) auxiliary supervision of the training step: ground truth from the HIP rasteriser + the losses.

Reference: core/trainer.py:442-464 renders the fitted mesh to an IUV image with pytorch3d on every step (``IUV_Renderer.verts2iuvimg``), crops
it to the ViTPose feature-map width (:454-455), turns it into target maps (``iuv_img2map``) and applies ``body_uv_losses`` (:255-298) to the
``dp_head`` outputs.  Here the image comes from ``whmr_amd.utils.renderer.IUV_Renderer`` (csrc/rasterize.hip: three launches, no host stall);
the target maps and the losses are O(B x 25 x 128 x 96) tensor arithmetic on the device through torch autograd (they back-propagate into
``dp_out`` only, whose producer -- the IUV head -- has a HIP backward).
"""
import torch
import torch.nn.functional as F

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
def render_iuv_targets(iuv_maker, verts, gt_camera, vitpose_crop=True):
    """-> (iuv_image_gt [B,3,H,W'], (Umap, Vmap, Imap, Annmap)) as core/trainer.py:447-464 builds them (valid_fit = all)."""
    img = iuv_maker.verts2iuvimg(verts, cam=gt_camera)
    if vitpose_crop:
        img = img[:, :, :, 16:-16]                                      # trainer.py:454-455 (cfg.MODEL.PyMAF.BACKBONE == 'vitpose')
    return img, iuv_img2map(img)


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
    """sum over the dp_head outputs of the four terms (trainer.py:466-482; equal map sizes -- the yaml default)"""
    total = 0.0
    for d in dp_out:
        lu, lv, li, la = body_uv_losses(d['predict_u'], d['predict_v'], d['predict_uv_index'], d['predict_ann_index'], uvia_list)
        total = total + lu + lv + li + (la if la is not None else 0.0)
    return total
