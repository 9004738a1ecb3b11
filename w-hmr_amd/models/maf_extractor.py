"""Mesh-aligned feature extractor -- reference surface of models/maf_extractor.py:17-143, one fused HIP launch inside.

``sampling(points, im_feat)`` / ``forward(p, ..., cam=)`` / ``reduce_dim(feature)`` keep their meaning and return
``(mesh_align_feat [B, 32*P], point_feat [B, 256, P])``.  ``im_feat`` / ``cam`` stay mutable attributes written by
WHMR.forward (whmr.py:564,593): an instance is not re-entrant, exactly like the reference.
"""
import torch
import torch.nn as nn

from .. import _lib as L
from ..core.cfgs import cfg
from ..core.constants import FOCAL_LENGTH


class MAF_Extractor(nn.Module):
    def __init__(self, device=None, Dmap=None):
        super().__init__()
        ch = list(cfg.MODEL.PyMAF.MLP_DIM)
        assert ch == [256, 128, 64, 32], 'the fused sampler kernel is built for MLP_DIM [256,128,64,32]'
        self.num_views = 1
        for l in range(len(ch) - 1):            # maf_extractor.py:33-46: skip-concat of the raw feature at layers 1, 2
            self.add_module('conv%d' % l, nn.Conv1d(ch[l] + (ch[0] if l else 0), ch[l + 1], 1))
        self.im_feat = None
        self.cam = None
        # 6890 -> 431 down-sampling map (maf_extractor.py:70-71); unused on the live path, kept for state_dict parity
        self.register_buffer('Dmap', Dmap if Dmap is not None else torch.zeros(431, 6890))
        self.crop_size = cfg.IMG_RES.WIDTH
        self._wcache = None

    def _weights(self):
        ps = [self.conv0.weight, self.conv0.bias, self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias]
        ver = tuple((p.device, p._version, p.data_ptr()) for p in ps)
        if self._wcache is None or self._wcache[0] != ver:
            keep = [ps[0].detach()[:, :, 0].t().contiguous(), ps[1].detach().contiguous(),
                    ps[2].detach()[:, :, 0].t().contiguous(), ps[3].detach().contiguous(),
                    ps[4].detach()[:, :, 0].t().contiguous(), ps[5].detach().contiguous()]
            w = L.WhmrMafWeights()
            w.w0t, w.b0, w.w1t, w.b1, w.w2t, w.b2 = [t.data_ptr() for t in keep]
            if ps[0].is_cuda:                      # bf16 [out][in] copies: the MFMA variant of the sampler (bf16 feature maps)
                keep += [L.cast_bf16(ps[i].detach()[:, :, 0].float().contiguous()) for i in (0, 2, 4)]
                w.w0b, w.w1b, w.w2b = [t.data_ptr() for t in keep[6:]]
            self._wcache = (ver, w, keep)
        return self._wcache[1]

    def _run(self, im_feat, pts2d=None, pts3d=None, cam=None, out=None, want_point_feat=True):
        B = im_feat.shape[0]
        P = (pts2d if pts2d is not None else pts3d).shape[1] if (pts2d is not None or pts3d is not None) else im_feat.shape[2]
        if out is None:
            out = torch.empty(B, 32 * P, dtype=torch.float32, device=im_feat.device)
        pf = torch.empty(B, 256, P, dtype=torch.float32, device=im_feat.device) if want_point_feat else None
        prof = L.PROFILE is not None and (pts2d is not None or pts3d is not None)
        if prof:                                   # bench.py's instrumented step: algorithmic bytes = P x 4 texels x 256 channels gathered + P x 32 fp32 written
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        L.maf_sample(im_feat, self._weights(), out, pts2d=pts2d, pts3d=pts3d, cam=cam, point_feat=pf, focal=FOCAL_LENGTH,
                     res_w=float(cfg.IMG_RES.WIDTH), res_h=float(cfg.IMG_RES.HEIGHT))
        if prof:
            e1.record()
            L.PROFILE.append(('maf_sample', B * P * (4 * 256 * im_feat.element_size() + 32 * 4.0), e0, e1))
        return out, pf

    @torch.no_grad()
    def reduce_dim(self, feature):
        """maf_extractor.py:75-101: [B,256,N] point features -> [B, 32*N]."""
        return self._run(feature.float(), want_point_feat=False)[0]

    @torch.no_grad()
    def sampling(self, points, im_feat=None, z_feat=None, out=None, want_point_feat=True):
        """maf_extractor.py:103-124: points [B,N,2] in [-1,1] (x,y), bilinear, align_corners=True, zero padding."""
        im_feat = self.im_feat if im_feat is None else im_feat
        return self._run(im_feat, pts2d=points.float().contiguous(), out=out, want_point_feat=want_point_feat)

    @torch.no_grad()
    def forward(self, p, center=None, scale=None, img_focal=None, img_center=None, s_feat=None, cam=None, out=None,
                want_point_feat=True, **kwargs):
        """maf_extractor.py:126-143: weak-perspective projection of p [B,N,3] with cam [B,3], then sampling (fused)."""
        cam = self.cam if cam is None else cam
        im_feat = self.im_feat if s_feat is None else s_feat
        return self._run(im_feat, pts3d=p.float().contiguous(), cam=cam if (cam.dtype == torch.float32 and cam.stride(-1) == 1) else cam.float().contiguous(),
                         out=out,
                         want_point_feat=want_point_feat)
