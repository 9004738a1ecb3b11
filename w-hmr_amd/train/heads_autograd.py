"""Autograd nodes of the small heads around the ViT / deconv / sampler / SMPL stages (training mode), all on the HIP GEMM kernels.

  LinearFn       nn.Linear (Regressor fc1/fc2/dec*, whmr.py:118-126; Tz-head Block and est_Tz linears, whmr.py:423-427): fp32 GEMM,
                 dX = dY . W, dW = dY^T . X (transposed operands from whmr_transpose_cast), db = column sum.
  ConvNHWCFn     Conv2d without bias / padding on a channels-last map (Tz head, whmr.py:419-420): forward = implicit GEMM with the
                 NHWC gather; dW = dY^T . col(X) with col(X)^T from whmr_im2col_t (or from dY when that side is narrower); dX = flipped-kernel
                 gather conv (stride 1), S*S residue-class gather convs (stride S, no padding) or col2im(dY . W) (whmr_col2im) otherwise.
  DownsampleFn   the mesh down-sampling products sub_verts = Dmap0 . verts, temp_verts = Dmap1 . sub_verts (whmr.py:182-183), applied
                 from the CSR form of the (densified-in-the-reference) matrices; backward = CSR of the transposes.
"""
import torch

import os

from .. import _lib as L

USE_TN = os.environ.get('WHMR_TN_GEMM', '1') != '0'      # weight gradients on the (gathering) TN kernel (A/B switch)


def _f32(t):
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


class LinearFn(torch.autograd.Function):
    """y = LinearFn.apply(x [M,K], weight [N,K], bias [N] | None)  (fp32)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        if not x.is_cuda:
            raise RuntimeError('whmr_amd runs on a HIP device only (no CPU fallback)')
        x = _f32(x.detach())
        w = _f32(weight.detach())
        y = torch.empty(x.shape[0], w.shape[0], dtype=torch.float32, device=x.device)
        L.gemm(x, w, y, bias=None if bias is None else _f32(bias.detach()))
        ctx.saved = (x, w)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved
        ctx.saved = None
        dy = _f32(dy)
        dev = dy.device
        dx = dw = db = None
        # dX = dY . W and dW = dY^T . X straight from dY / X / W as stored (reduction-major operands of the skinny fp32 kernel): no transposed
        # copies -- three launches fewer per Linear, and the heads' backward is a chain of small launches on the critical stream
        direct = w.shape[0] <= 1024 and x.shape[0] <= 1024
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            if direct:
                L.gemm(dy, w, dx, trans_w=True)
            else:
                L.gemm(dy, L.transpose_cast(w, torch.float32, pad_to=1), dx)
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            if direct:
                L.gemm(dy, x, dw, trans_a=True, trans_w=True)
            else:
                L.gemm(L.transpose_cast(dy, torch.float32, pad_to=1), L.transpose_cast(x, torch.float32, pad_to=1), dw)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.empty(w.shape[0], dtype=torch.float32, device=dev)
            L.colsum(dy, db)
        return dx, dw, db


class LayerNormFn(torch.autograd.Function):
    """y = LayerNormFn.apply(x [R, C] fp32, weight, bias, eps): whmr_layernorm forward, whmr_layernorm_bwd backward (timm Block of the Tz head)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        if not x.is_cuda:
            raise RuntimeError('whmr_amd runs on a HIP device only (no CPU fallback)')
        x = _f32(x.detach())
        y = torch.empty_like(x)
        L.layernorm(x, _f32(weight.detach()), _f32(bias.detach()), y, eps)
        ctx.saved, ctx.eps = (x, weight), eps
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved
        ctx.saved = None
        C = x.shape[-1]
        dx = torch.empty_like(x)
        dg = torch.empty(C, dtype=torch.float32, device=x.device)
        db = torch.empty(C, dtype=torch.float32, device=x.device)
        L.layernorm_bwd(x, _f32(dy), _f32(weight.detach()), None, dx, dg, db, ctx.eps)
        return dx, dg, db, None


class GeluFn(torch.autograd.Function):
    """exact-erf GELU: whmr_gelu_fwd / whmr_gelu_bwd (fp32)"""

    @staticmethod
    def forward(ctx, pre):
        pre = _f32(pre.detach())
        out = torch.empty_like(pre)
        L.gelu_fwd(pre, out)
        ctx.saved = pre
        return out

    @staticmethod
    def backward(ctx, dh):
        pre = ctx.saved
        ctx.saved = None
        dpre = torch.empty_like(pre)
        L.gelu_bwd(pre, _f32(dh), dpre)
        return dpre


class AttentionF32Fn(torch.autograd.Function):
    """att [B*N, H*d] = AttentionF32Fn.apply(qkv [B*N, 3*H*d] fp32, B, N, H, d, scale): softmax(scale q k^T) v per (image, head) on the fp32
    kernels (forward whmr_attention, backward whmr_attention_bwd_f32) -- the timm Attention of the Tz head's Block (whmr.py:423,574)."""

    @staticmethod
    def forward(ctx, qkv, B, N, H, d, scale):
        qkv = _f32(qkv.detach())
        out = torch.empty(B * N, H * d, dtype=torch.float32, device=qkv.device)
        L.attention(qkv, out, B, N, H, d, scale)
        ctx.saved, ctx.dims = qkv, (B, N, H, d, scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv = ctx.saved
        ctx.saved = None
        B, N, H, d, scale = ctx.dims
        return L.attention_bwd_f32(qkv, _f32(dout), B, N, H, d, scale), None, None, None, None, None


class MatToAAFn(torch.autograd.Function):
    """aa [n, 3] = MatToAAFn.apply(R [n, 9] | [n, 3, 3]): rotation_matrix_to_angle_axis (utils/geometry.py:54-83) with its backward
    (whmr_mat_to_aa / whmr_mat_to_aa_bwd) -- where the reference's training graph carries it: theta's pose (whmr.py:174), global_pose (:632-633)."""

    @staticmethod
    def forward(ctx, R):
        if not R.is_cuda:
            raise RuntimeError('whmr_amd runs on a HIP device only (no CPU fallback)')
        Rm = _f32(R.detach().reshape(-1, 9))
        ctx.saved, ctx.shape = Rm, R.shape
        return L.mat_to_aa(Rm)

    @staticmethod
    def backward(ctx, d_aa):
        Rm = ctx.saved
        ctx.saved = None
        return L.mat_to_aa_bwd(Rm, _f32(d_aa)).view(ctx.shape)


def _padded_base(dy, M, npad, dt):
    """A producer that already holds the gradient of a ConvNHWCFn output as the zero-padded [M, npad] matrix (IUVLossFn: csrc/iuv_loss.hip writes it
    that way) returns the [..., :Cout] VIEW of it; the convolution's backward then takes the whole buffer from the view's base and skips the
    zero-fill + copy that would rebuild that operand.  The producer OPTS IN by tagging the buffer (``base.whmr_zero_padded = npad``: "columns
    [Cout, npad) are zero and I am laid out [M, npad]"); geometry alone is not trusted, and the view must be exactly
    ``base.view(..., npad)[..., :Cout]`` -- same storage offset, same strides in EVERY dimension."""
    base = dy._base
    if base is None or getattr(base, 'whmr_zero_padded', None) != npad:
        return None
    if base.dtype != dt or base.numel() != M * npad or not base.is_contiguous() or base.data_ptr() != dy.data_ptr():
        return None
    want = base.view(*dy.shape[:-1], npad)[..., :dy.shape[-1]]
    if dy.dim() < 2 or dy.stride() != want.stride() or dy.shape != want.shape:
        return None
    return base.view(M, npad)


# Rows the weight-gradient operand dY is widened to = the smallest row tile the gathering TN kernel is built with (whmr_conv_dw_tn_bf16: its 64-row
# instantiation changed the results of smpl_skin_bwd_kernel running beside it on the other stream, see gemm_tn.hip, and is gone).
TN_ROW_PAD = 128
GROUP_DX = os.environ.get('WHMR_TRAIN_GROUP_DX', '1') != '0'        # the residue-class data gradients of a strided convolution as one grouped launch
_residue_index = {}


def _residue_class_index(npad, KH, KW, Cin, S, dev):
    """-> (idx, offsets): ``wm.reshape(-1)[idx]`` = the S*S matrices w4[:, ry::S, rx::S, :].flip(1, 2).permute(3, 1, 2, 0) of ConvNHWCFn.backward, one
    after the other (class ry*S + rx starts at offsets[ry*S + rx]); built once per geometry."""
    key = (npad, KH, KW, Cin, S, dev)
    hit = _residue_index.get(key)
    if hit is None:
        ar = torch.arange(npad * KH * KW * Cin, dtype=torch.int64).view(npad, KH, KW, Cin)
        parts = [ar[:, ry::S, rx::S, :].flip(1, 2).permute(3, 1, 2, 0).reshape(-1) for ry in range(S) for rx in range(S)]
        offs, o = [], 0
        for q in parts:
            offs.append(o)
            o += q.numel()
        hit = _residue_index[key] = (torch.cat(parts).to(dev), offs)
    return hit


def _flipped_taps_index(npad, KH, KW, Cin, dev):
    """``wm.reshape(-1)[idx]`` = wm.view(npad, KH, KW, Cin).flip(1, 2).permute(3, 1, 2, 0) flattened: the weight operand [Cin, (ky, kx, co)] of a stride-1
    'same' convolution's data gradient; built once per geometry."""
    key = ('flip', npad, KH, KW, Cin, dev)
    hit = _residue_index.get(key)
    if hit is None:
        ar = torch.arange(npad * KH * KW * Cin, dtype=torch.int64).view(npad, KH, KW, Cin)
        hit = _residue_index[key] = ar.flip(1, 2).permute(3, 1, 2, 0).reshape(-1).to(dev)
    return hit


class ConvNHWCFn(torch.autograd.Function):
    """y [B,OH,OW,Cout] (dt) = ConvNHWCFn.apply(x [B,IH,IW,Cin] (dt), weight [Cout,Cin,KH,KW], stride, dt, padding=0, bias=None): Conv2d on a
    channels-last map (Tz head: 7x7 s3 / s2 without bias or padding, whmr.py:419-420; IUV head: 3x3 s1 p1 with bias, iuv_predictor.py:71-91)."""

    @staticmethod
    def forward(ctx, x, weight, stride, dt, padding=0, bias=None, passthrough=False):
        if not x.is_cuda:
            raise RuntimeError('whmr_amd runs on a HIP device only (no CPU fallback)')
        ctx.set_materialize_grads(False)
        ctx.passthrough = passthrough
        x = x.detach()
        assert x.dtype == dt and x.is_contiguous()
        B, IH, IW, Cin = x.shape
        Cout, _, KH, KW = weight.shape
        OH, OW = (IH + 2 * padding - KH) // stride + 1, (IW + 2 * padding - KW) // stride + 1
        npad = Cout if dt == torch.float32 else (Cout + 63) // 64 * 64           # the bf16 tiles want whole 64-column groups
        wm = torch.zeros(npad, KH * KW * Cin, dtype=torch.float32, device=x.device)
        wm[:Cout] = weight.detach().float().permute(0, 2, 3, 1).reshape(Cout, -1)   # [co, (ky, kx, ci)]
        wm = L.cast_bf16(wm) if dt == torch.bfloat16 else wm
        bp = None
        if bias is not None:
            bp = torch.zeros(npad, dtype=torch.float32, device=x.device)
            bp[:Cout] = bias.detach().float()
        y = torch.empty(B * OH * OW, npad, dtype=dt, device=x.device)
        L.gemm(x, wm, y, bias=bp, conv=dict(IH=IH, IW=IW, Cin=Cin, OH=OH, OW=OW, KW=KW, SH=stride, SW=stride, PH=padding, PW=padding))
        ctx.saved = (x, wm)
        ctx.dims = (B, IH, IW, Cin, Cout, KH, KW, OH, OW, stride, padding, npad, dt)
        ctx.has_bias = bias is not None
        # padded channel groups are returned as a strided view (the consumer's dtype cast / permute reads it once; no compaction pass)
        y = y.view(B, OH, OW, npad)[..., :Cout] if npad != Cout else y.view(B, OH, OW, Cout)
        # passthrough: the input map is handed on as a second output.  A map with several consumers (the last feature map feeds the Tz head, the
        # IUV head and the stage-3 sampler) is threaded through them as a chain; each backward then ADDS its data gradient to the one arriving
        # from the consumers behind it, in place in the GEMM epilogue, instead of autograd summing full-size maps (0.2 ms per add at batch 64).
        return (y, x.view_as(x)) if passthrough else y

    @staticmethod
    def backward(ctx, dy, dx_in=None):
        if dy is None:                                                    # only the handed-on map was used
            return (dx_in,) + (None,) * 6
        x, wm = ctx.saved
        ctx.saved = None
        B, IH, IW, Cin, Cout, KH, KW, OH, OW, S, P, npad, dt = ctx.dims
        dev = x.device
        M, K = B * OH * OW, KH * KW * Cin
        dyp = _padded_base(dy, M, npad, dt) if npad != Cout else None
        if dyp is not None:
            pass
        elif npad != Cout:
            dyp = torch.zeros(M, npad, dtype=dt, device=dev)
            dyp[:, :Cout] = dy.reshape(M, Cout)
        else:
            dyp = dy.reshape(M, Cout).to(dt).contiguous()
        pad = 64 if dt == torch.bfloat16 else 8
        dx = dw = db = None
        same = S == 1 and OH == IH and OW == IW
        if ctx.needs_input_grad[1] and USE_TN and dt == torch.bfloat16 and Cin % 256 == 0 and M % 32 == 0 and x.is_contiguous():
            # gathering TN kernel: dW[co, (ky, kx, ci)] = sum_m dY[m, co] . X[pixel(m) + tap, ci] from dY and X as they are -- no transposed
            # dY, no (transposed) column matrix of X.  dY is widened to a multiple of 128 columns (the kernel's smallest row tile) if need be.
            na = (npad + TN_ROW_PAD - 1) // TN_ROW_PAD * TN_ROW_PAD
            if na != npad:
                dya = torch.zeros(M, na, dtype=dt, device=dev)
                dya[:, :npad] = dyp
            else:
                dya = dyp
            dwm = torch.empty(na, K, dtype=torch.float32, device=dev)
            dba = torch.empty(na, dtype=torch.float32, device=dev) if (ctx.has_bias and ctx.needs_input_grad[5]) else None
            L.conv_dw_tn(dya, x, dwm, OH, OW, KH, KW, S, P, db=dba)                    # bias gradient = column sums of dY, same pass
            dw = dwm[:Cout].view(Cout, KH, KW, Cin).permute(0, 3, 1, 2)
            if dba is not None:
                db = dba[:Cout]
        elif ctx.needs_input_grad[1] and same and npad % 64 == 0 and npad < Cin:
            # stride-1 'same' convolution with fewer output than input channels (IUV head: 128 padded vs 256): gather the SMALLER operand.
            # dW[co, ci, ky, kx] = sum_m' dY[m' - shift(ky, kx), co] . X[m', ci]: the transposed column matrix is built from dY with the
            # flipped taps (KH*KW*npad rows instead of KH*KW*Cin) and X is transposed once.
            xt = L.transpose_cast(x.view(B * IH * IW, Cin), dt, pad_to=pad)        # [Cin, Mpad]
            dcolt = L.im2col_t(dyp.view(B, OH, OW, npad), IH, IW, KH, KW, 1, KH - 1 - P, pad_to=pad)     # [(k'y, k'x, co), Mpad]
            dwm = torch.empty(KH * KW * npad, Cin, dtype=torch.float32, device=dev)
            L.gemm(dcolt, xt, dwm)
            dw = dwm.view(KH, KW, npad, Cin).flip(0, 1)[:, :, :Cout].permute(2, 3, 0, 1)              # [co, ci, ky, kx]
            del dcolt, xt
        elif ctx.needs_input_grad[1]:
            dyt = L.transpose_cast(dyp, dt, pad_to=pad)                            # [npad, Mpad]
            colt = L.im2col_t(x, OH, OW, KH, KW, S, P, pad_to=pad)                 # [K, Mpad]
            dwm = torch.empty(npad, K, dtype=torch.float32, device=dev)
            L.gemm(dyt, colt, dwm)
            dw = dwm[:Cout].view(Cout, KH, KW, Cin).permute(0, 3, 1, 2)
            del colt, dyt
        if db is None and ctx.has_bias and ctx.needs_input_grad[5]:
            dbp = torch.empty(npad, dtype=torch.float32, device=dev)
            L.colsum(dyp, dbp)
            db = dbp[:Cout]
        if ctx.needs_input_grad[0]:
            # gradient already left on the handed-on map by the consumers behind this one: accumulate into it in the epilogue when the kernel can
            acc = (dx_in is not None and dx_in.dtype == dt and dx_in.is_contiguous() and tuple(dx_in.shape) == (B, IH, IW, Cin)
                   and (same and npad % 64 == 0 or (dt == torch.bfloat16 and P == 0 and S > 1 and KH >= S and KW >= S and npad % 64 == 0 and Cin % 8 == 0)))
            dx = dx_in if acc else torch.empty(B, IH, IW, Cin, dtype=dt, device=dev)
            if same and npad % 64 == 0:
                # stride 1, 'same' padding: the data gradient is itself a convolution of dY with the flipped kernel -- an implicit GEMM
                # with the NHWC gather (K = KH*KW*npad), no column matrix at all
                # (one gather of wm through a cached index instead of float / flip / copy / cast: four launches on the side stream's chain)
                w4 = wm.reshape(-1).index_select(0, _flipped_taps_index(npad, KH, KW, Cin, dev)).view(Cin, KH * KW * npad)
                L.gemm(dyp.view(B, OH, OW, npad), w4, dx.view(B * IH * IW, Cin), accumulate=acc,
                       conv=dict(IH=OH, IW=OW, Cin=npad, OH=IH, OW=IW, KW=KW, SH=1, SW=1, PH=KH - 1 - P, PW=KW - 1 - P))
            elif P == 0 and S > 1 and KH >= S and KW >= S and (dt == torch.float32 or npad % 64 == 0):
                # strided convolution (Tz head, 7x7 s3 / s2): the input pixels split into S*S residue classes (iy mod S, ix mod S); class
                # (ry, rx) only ever meets the taps ky = ry + S*t, kx = rx + S*u, so its data gradient is a small stride-1 convolution of dY
                # (ceil((KH-ry)/S) x ceil((KW-rx)/S) taps, flipped) -- S*S implicit GEMMs that scatter into the interleaved pixels, instead
                # of a 2 GB column-space gradient plus col2im.
                # The S*S weight matrices (flipped taps of one class, [Cin, (t, u, co)]) are ONE gather of wm through a cached index; in bf16 the
                # S*S GEMMs -- each under two rounds of tiles, K = 4 .. 9 taps -- are ONE grouped launch (whmr_gemm_bf16_group).
                idx, offs = _residue_class_index(npad, KH, KW, Cin, S, dev)
                wp_all = wm.reshape(-1).index_select(0, idx)
                dy_img = dyp.view(B, OH, OW, npad)
                descs = []
                for ry in range(S):
                    for rx in range(S):
                        Ty, Tx = len(range(ry, KH, S)), len(range(rx, KW, S))
                        Jy, Jx = (IH - ry + S - 1) // S, (IW - rx + S - 1) // S
                        o = offs[ry * S + rx]
                        wp = wp_all[o:o + Cin * Ty * Tx * npad].view(Cin, Ty * Tx * npad)
                        kw = dict(conv=dict(IH=OH, IW=OW, Cin=npad, OH=Jy, OW=Jx, KW=Tx, SH=1, SW=1, PH=Ty - 1, PW=Tx - 1),
                                  scatter=dict(c_off=(ry * IW + rx) * Cin, osb=IH * IW * Cin, osy=S * IW * Cin, osx=S * Cin), accumulate=acc)
                        if GROUP_DX and dt == torch.bfloat16 and S * S <= 9:
                            descs.append(L.gemm(dy_img, wp, dx, desc_only=True, **kw))
                        else:
                            L.gemm(dy_img, wp, dx, **kw)
                if descs:
                    L.gemm_group(descs, 192 if Cin > 64 else 65)
            else:
                wt = L.transpose_cast(wm, dt, pad_to=1)                            # [K, npad]
                dcol = torch.empty(M, K, dtype=dt, device=dev)
                L.gemm(dyp, wt, dcol)
                L.col2im(dcol, dx, OH, OW, KH, KW, S, P)
            if dx_in is not None and not acc:
                dx = dx + dx_in.to(dx.dtype)
        elif dx_in is not None:
            dx = dx_in
        return dx, dw, None, None, None, db, None


class TzComposedFn(torch.autograd.Function):
    """tokens [B * 5, OH2 * OW2] (fp32) = TzComposedFn.apply(x [B, H, W, 256] bf16, w0 [64, 256, 7, 7], w1 [5, 64, 7, 7], passthrough=False): the Tz head's
    two bias-free convolutions (whmr.py:418-421 as applied at :567-571) as ONE Conv2d(256, 5, k25, s6) in the training graph.

    The two-convolution graph pays for the 64-channel map between them three times: conv0 forward (134 GF), its weight gradient and its data gradient, all
    at N = 64 / K = 4-9 taps where the GEMM kernels run at a third of their rate (0.35 + 0.38 + 0.46 ms at batch 64).  Through the composed weight
    Wc = compose(w0, w1) the same function and the same gradients cost 53 GF each: with the map read as [B, H, W/6, 6 C] (every byte once),
        forward   P = X . G^T (implicit GEMM, 6 x 1 window at stride 6 x 1; G = whmr_tz_compose(w1p . w0)), tokens = fold(P)
        dG        = dP^T . X          (the gathering TN kernel with row / column strides 6 / 1; dP = unfold(d tokens), 128 columns)
        dX        = dP . G            (six GEMMs of K = 128, one per window row, each writing its 1536 contiguous channels of every pixel once)
        dw1, dw0  = two small fp32 GEMMs of dT = compose^T(dG) with w0 / w1 (chain rule through the composition).
    bf16 numerics only (the fp32 mode keeps the two convolutions: its parity tests compare them with the reference to 1e-4)."""

    @staticmethod
    def fits(x):
        """geometry envelope: bf16 contiguous map, W a multiple of the composed stride, whole 256-channel groups (the TN gather), B * ceil(H / 6) * W / 6
        rows a multiple of 32 (its K step) -- 352 rows per image at 128 x 96, i.e. every batch size of the model's geometry"""
        B, H, W, C = x.shape
        return (x.dtype == torch.bfloat16 and x.is_contiguous() and W % 6 == 0 and C % 256 == 0 and (B * ((H + 5) // 6) * (W // 6)) % 32 == 0
                and (H + 5) // 6 >= ((H - 7) // 3 + 1 - 7) // 2 + 5 and W // 6 >= ((W - 7) // 3 + 1 - 7) // 2 + 5)

    @staticmethod
    def forward(ctx, x, w0, w1, passthrough=False):
        if not x.is_cuda:
            raise RuntimeError('whmr_amd runs on a HIP device only (no CPU fallback)')
        ctx.set_materialize_grads(False)
        x = x.detach()
        B, H, W, C = x.shape
        assert TzComposedFn.fits(x) and tuple(w0.shape) == (64, C, 7, 7) and tuple(w1.shape) == (5, 64, 7, 7)
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        w0m = w0.detach().float().reshape(64, C * 49)
        w1p = w1.detach().float().permute(2, 3, 0, 1).reshape(245, 64).contiguous()          # [(u, v, o), c1]
        T = torch.empty(245, C * 49, **f32)
        L.gemm(w1p, w0m, T, trans_w=True)
        g = L.tz_compose(T, C, torch.empty(128, 36 * C, dtype=torch.bfloat16, device=dev))
        H2, W2 = ((H - 7) // 3 + 1 - 7) // 2 + 1, ((W - 7) // 3 + 1 - 7) // 2 + 1
        OHp, OWp = (H + 5) // 6, W // 6
        assert OHp >= H2 + 4 and OWp >= W2 + 4
        M = B * OHp * OWp
        ns = 2 if M >= 8192 else 1
        P = torch.empty(ns, M, 128, **f32)
        kw = dict(tile=64, raw_splits=2) if ns == 2 else {}
        L.gemm(x.view(B, H, OWp, 6 * C), g, P if ns > 1 else P[0], conv=dict(IH=H, IW=OWp, Cin=6 * C, OH=OHp, OW=OWp, KW=1, SH=6, SW=1, PH=0, PW=0), **kw)
        t = torch.empty(B * 5, H2 * W2, **f32)
        L.tz_fold(P, t, B, OHp, OWp, H2, W2, halves=1, nsplit=ns, split_stride=M * 128)
        ctx.saved = (x, g, w0m, w1p)
        ctx.dims = (B, H, W, C, OHp, OWp, H2, W2)
        ctx.passthrough = passthrough
        return (t, x.view_as(x)) if passthrough else t

    @staticmethod
    def backward(ctx, dt, dx_in=None):
        if dt is None:
            return dx_in, None, None, None
        x, g, w0m, w1p = ctx.saved
        ctx.saved = None
        B, H, W, C, OHp, OWp, H2, W2 = ctx.dims
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        dP = L.tz_unfold(dt.float().contiguous(), torch.empty(B * OHp * OWp, 128, dtype=torch.bfloat16, device=dev), B, OHp, OWp, H2, W2)
        dw0 = dw1 = dx = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dG = torch.empty(128, 36 * C, **f32)
            L.conv_dw_tn(dP, x.view(B, H, OWp, 6 * C), dG, OHp, OWp, 6, 1, (6, 1), 0)
            dT = L.tz_compose_bwd(dG, C, torch.empty(245, C * 49, **f32))
            if ctx.needs_input_grad[2]:
                dw1p = torch.empty(245, 64, **f32)
                L.gemm(dT, w0m, dw1p)                                                  # [(u, v, o), c1] = dT . w0^T
                dw1 = dw1p.view(7, 7, 5, 64).permute(2, 3, 0, 1)
            if ctx.needs_input_grad[1]:
                dw0m = torch.empty(64, C * 49, **f32)
                L.gemm(w1p, dT, dw0m, trans_a=True, trans_w=True)                      # [c1, (ci, a, b)] = w1p^T . dT
                dw0 = dw0m.view(64, C, 7, 7)
        if ctx.needs_input_grad[0]:
            acc = dx_in is not None and dx_in.dtype == torch.bfloat16 and dx_in.is_contiguous() and tuple(dx_in.shape) == (B, H, W, C)
            dx = dx_in if acc else torch.empty(B, H, W, C, dtype=torch.bfloat16, device=dev)
            gT = g.t().contiguous()                                                    # [(q, p, ci), 128]
            img = dP.view(B, OHp, OWp, 128)
            descs = []
            for q in range(6):                                                         # window row q: map rows y = 6 Y + q < H
                OHq = (H - 1 - q) // 6 + 1
                descs.append(L.gemm(img, gT[q * 6 * C:(q + 1) * 6 * C], dx, desc_only=True, accumulate=acc,
                                    conv=dict(IH=OHp, IW=OWp, Cin=128, OH=OHq, OW=OWp, KW=1, SH=1, SW=1, PH=0, PW=0),
                                    scatter=dict(c_off=q * W * C, osb=H * W * C, osy=6 * W * C, osx=6 * C)))
            L.gemm_group(descs, 192)
            if dx_in is not None and not acc:
                dx = dx + dx_in.to(dx.dtype)
        elif dx_in is not None:
            dx = dx_in
        return dx, dw0, dw1, None


def downsample_csr(d0, d1, cache):
    """CSR triples of Dmap0, Dmap1 and their transposes, rebuilt when a buffer changes (one host sync each, outside any graph capture)."""
    key = (d0.data_ptr(), d0._version, d1.data_ptr(), d1._version)
    if cache.get('key') != key:
        cache.update(key=key, d0=L.dense_to_csr(d0), d1=L.dense_to_csr(d1), d0t=L.dense_to_csr(d0.t()), d1t=L.dense_to_csr(d1.t()))
    return cache


class DownsampleFn(torch.autograd.Function):
    """sub_verts [B,n0,3], temp_verts [B,n1,3] = DownsampleFn.apply(verts [B,6890,3], Dmap0 [n0,6890], Dmap1 [n1,n0], cache_dict).
    The matrices are applied in compressed form (~3 non-zeros per row in data/mesh_downsampling.npz; any dense content works)."""

    @staticmethod
    def forward(ctx, verts, d0, d1, cache):
        ctx.set_materialize_grads(False)
        c = downsample_csr(d0, d1, cache)
        sub = L.csr_apply3(c['d0'], verts.detach(), d0.shape[0])
        tmp = L.csr_apply3(c['d1'], sub, d1.shape[0])
        ctx.cache, ctx.dims = c, (d0.shape[1], d0.shape[0])
        return sub, tmp

    @staticmethod
    def backward(ctx, d_sub, d_tmp):
        c = ctx.cache
        n_v, n0 = ctx.dims
        ds = None
        if d_tmp is not None:
            ds = L.csr_apply3(c['d1t'], d_tmp, n0)                                   # Dmap1^T . d_tmp
        if d_sub is not None:
            ds = d_sub.float().contiguous() if ds is None else ds + d_sub
        if ds is None:
            return None, None, None, None
        return L.csr_apply3(c['d0t'], ds, n_v), None, None, None


class RegressorPostFn(torch.autograd.Function):
    """kp_2d, kp_2d_w, cam_t, focal = RegressorPostFn.apply(joints [B,J,3], cam [B,3], Tz [B], bbox_h, center, orig_shape, stage, consts):
    the regressor tail of whmr.py:142-173 in one launch forward and one backward (consts = (FOCAL_LENGTH, IMG_RES.WIDTH, IMG_RES.HEIGHT))."""

    @staticmethod
    def forward(ctx, joints, cam, Tz, bbox_h, center, orig_shape, stage, consts):
        if not joints.is_cuda:
            raise RuntimeError('whmr_amd runs on a HIP device only (no CPU fallback)')
        ctx.set_materialize_grads(False)            # cam_t / focal (and kp_2d_w at TRAIN.STAGE 1) usually carry no loss: the kernel takes null cotangents
        j, c, t = _f32(joints.detach()), _f32(cam.detach()), _f32(Tz.detach())
        bh, ce, os_ = _f32(bbox_h), _f32(center), _f32(orig_shape)
        B, J = j.shape[0], j.shape[1]
        dev = j.device
        kp, kw = torch.empty(B, J, 2, device=dev), torch.empty(B, J, 2, device=dev)
        ct, fo = torch.empty(B, 3, device=dev), torch.empty(B, device=dev)
        L._check(L.lib().whmr_regressor_post_train(j.data_ptr(), c.data_ptr(), t.data_ptr(), bh.data_ptr(), ce.data_ptr(), os_.data_ptr(), B, J,
                                                   consts[0], consts[1], consts[2], kp.data_ptr(), kw.data_ptr(), ct.data_ptr(), fo.data_ptr(),
                                                   L._stream()), 'whmr_regressor_post_train')
        ctx.saved = (j, c, t, bh, ce, os_)
        ctx.stage, ctx.consts = int(stage), consts
        return kp, kw, ct, fo

    @staticmethod
    def backward(ctx, d_kp, d_kw, d_ct, d_fo):
        j, c, t, bh, ce, os_ = ctx.saved
        ctx.saved = None
        B, J = j.shape[0], j.shape[1]
        dev = j.device
        g = [None if x is None else _f32(x) for x in (d_kp, d_kw, d_ct, d_fo)]
        dj, dc, dt = torch.empty(B, J, 3, device=dev), torch.empty(B, 3, device=dev), torch.empty(B, device=dev)
        L._check(L.lib().whmr_regressor_post_train_bwd(j.data_ptr(), c.data_ptr(), t.data_ptr(), bh.data_ptr(), ce.data_ptr(), os_.data_ptr(), B, J,
                                                       ctx.consts[0], ctx.consts[1], ctx.consts[2], ctx.stage, L._ptr(g[0]), L._ptr(g[1]),
                                                       L._ptr(g[2]), L._ptr(g[3]), dj.data_ptr(), dc.data_ptr(), dt.data_ptr(), L._stream()),
                 'whmr_regressor_post_train_bwd')
        return dj, dc, dt, None, None, None, None, None
