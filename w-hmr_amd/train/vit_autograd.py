"""Forward-with-activations and hand-driven backward of the ViT backbone (north_star "forward/backward path").

Reference semantics: autograd of ``ViT.forward_features`` (models/ViTPose/mmpose/models/backbones/vit.py:61-140,313-332) as
driven by ``core/trainer.py:410-470`` (loss.backward()).  Here the backward pass is written out layer by layer:

  * every matrix product -- dX = dY.W and dW = dY^T.X of the four Linears per block and of the patch embedding -- runs on the
    forward GEMM kernels (``whmr_gemm_bf16`` / ``whmr_gemm_f32``): W^T and the transposed activations come from
    ``whmr_transpose_cast`` (rows zero-padded to a multiple of 64 so that the token dimension can be the GEMM's K);
  * LayerNorm backward (+ residual-stream accumulation), exact-erf GELU backward and the bias column sums are HIP kernels
    (``train_ops.hip``), all deterministic (two-stage reductions, no atomics);
  * the attention core's backward is the MFMA kernel ``whmr_attention_bwd`` (bf16 mode, head dim 64, 64 < N <= 224; P is
    recomputed from the saved qkv and the log-sum-exp the training forward writes); the fp32 parity mode and other shapes
    run the fp32 HIP kernel ``whmr_attention_bwd_f32`` (any head dim, N <= 256).

``numerics='fp32'`` (exact-f32 MFMA) is the parity mode against the CPU reference's autograd; ``'bf16'`` casts GEMM operands
to bf16 and keeps the residual-stream gradient, LayerNorm statistics and all weight gradients in fp32.
Stochastic depth (vit.py:132-139,233; drop_path_rate 0.3 for ViTPose-B, 0.5 for -L): in training mode block i multiplies each of its two
residual branches by a per-sample mask / keep_prob (keep_prob = 1 - linspace(0, rate, depth)[i]).  The factor rides in the proj / fc2
GEMM epilogue (``row_scale``) and, in the backward pass, in the cast that makes the GEMM operand out of the residual-stream gradient
(``whmr_scale_rows_cast``).  Masks are drawn with torch's generator on the device (``ViT.drop_masks`` injects them in tests).
The Dropout layers of the reference ViT have p = 0 (drop_rate / attn_drop_rate defaults).
"""
import os

import torch

from .. import _lib as L


class _Saved:
    pass


def _dt(m):
    return torch.bfloat16 if m.numerics == 'bf16' else torch.float32          # 'bf16x3' is an inference mode: it trains in fp32


def _op(t, dt):
    """GEMM operand copy of an fp32 activation / gradient in the compute dtype."""
    if t.dtype == dt:
        return t
    return L.cast_bf16(t) if dt == torch.bfloat16 else t.float()


@torch.no_grad()
def vit_forward_train(m, x):
    """Same arithmetic as ViT.forward_tokens except that GELU is its own (exact-erf) pass; keeps what the backward needs."""
    if not x.is_cuda:
        raise RuntimeError('whmr_amd.ViT runs on a HIP device only (no CPU fallback)')
    B, Cin, H, W = x.shape
    P, pad, D = m.patch_size, m.patch_pad, m.embed_dim
    Hp, Wp = (H + 2 * pad - P) // P + 1, (W + 2 * pad - P) // P + 1
    N, M = Hp * Wp, B * Hp * Wp
    assert N + 1 == m.pos_embed.shape[1], 'input size does not match pos_embed (vit.py:231)'
    dt, dev = _dt(m), x.device
    f32 = dict(dtype=torch.float32, device=dev)
    s = _Saved()
    s.dims = (B, N, Hp, Wp, M)
    s.cols = torch.empty(M, Cin * P * P, dtype=dt, device=dev)
    L.patch_im2col(x.float(), s.cols, P, pad)
    pos = m.pos_embed[0, 1:] + m.pos_embed[0, :1]
    t = torch.empty(M, D, **f32)
    s.ops = ops = _weight_operands(m) if dt == torch.bfloat16 else None

    def wop(p, shape=None):                      # W in the compute dtype: the weight operand of y = x . W^T
        return ops[id(p)][0] if ops is not None else m._w(p, shape, eff='fp32')      # ops is None <=> dt is fp32 ('fp32' and 'bf16x3' models)

    L.gemm(s.cols, wop(m.patch_embed.proj.weight, (D, Cin * P * P)), t, bias=m.patch_embed.proj.bias, residual=pos, res_row_mod=N)
    s.layers = []
    hidden = m.blocks[0].mlp.fc1.weight.shape[0] if m.depth else 0
    # stochastic depth: one [2*depth, B] draw per forward (timm drop_path: mask = floor(keep_prob + U[0,1))), expanded to token rows
    rows = None
    if m.training and m.drop_path_rate > 0.0:
        cache = m.__dict__.setdefault('_keep_prob', {})              # built once per device, outside any graph capture (an H2D copy is not capturable)
        keep = cache.get(dev)
        if keep is None:
            keep = cache[dev] = (1.0 - torch.tensor(m.dpr, **f32)).repeat_interleave(2).view(-1, 1)
        masks = m.drop_masks.to(dev).float() if m.drop_masks is not None else torch.floor(keep + torch.rand(2 * m.depth, B, **f32))
        # mask / keep_prob per (branch, sample), expanded to token rows (row m = (b, n) -> sample b) for ALL branches in two launches
        # (was a divide + an expansion per branch: 48 small launches, 0.2 ms of the step)
        rows = (masks / keep).repeat_interleave(N, dim=1)            # [2 * depth, M]
    for li, blk in enumerate(m.blocks):
        a = _Saved()
        a.rs_attn = a.rs_mlp = None
        if rows is not None and m.dpr[li] > 0.0:
            a.rs_attn, a.rs_mlp = rows[2 * li], rows[2 * li + 1]
        a.t_in = t
        a.h1 = torch.empty(M, D, dtype=dt, device=dev)
        L.layernorm(a.t_in, blk.norm1.weight, blk.norm1.bias, a.h1, 1e-6)
        a.qkv = torch.empty(M, 3 * D, dtype=dt, device=dev)
        L.gemm(a.h1, wop(blk.attn.qkv.weight), a.qkv, bias=blk.attn.qkv.bias)
        a.att = torch.empty(M, D, dtype=dt, device=dev)
        a.lse = None
        if _hip_attention_bwd(m, N):
            a.lse = torch.empty(B * m.num_heads * N, **f32)
            L.attention_fwd_train(a.qkv, a.att, a.lse, B, N, m.num_heads, D // m.num_heads, m.scale)
        else:
            L.attention(a.qkv, a.att, B, N, m.num_heads, D // m.num_heads, m.scale)
        a.t_mid = torch.empty(M, D, **f32)
        L.gemm(a.att, wop(blk.attn.proj.weight), a.t_mid, bias=blk.attn.proj.bias, residual=a.t_in, row_scale=a.rs_attn)
        a.h2 = torch.empty(M, D, dtype=dt, device=dev)
        L.layernorm(a.t_mid, blk.norm2.weight, blk.norm2.bias, a.h2, 1e-6)
        a.pre = torch.empty(M, hidden, dtype=dt, device=dev)
        a.hid = torch.empty_like(a.pre)
        if FUSE_GELU and dt == torch.bfloat16:
            # fc1's epilogue leaves both the pre-activation (the backward needs it) and its GELU: no separate pass over the [M, 4D] map
            L.gemm(a.h2, wop(blk.mlp.fc1.weight), a.hid, bias=blk.mlp.fc1.bias, act=L.ACT_GELU, pre_out=a.pre)
        else:
            L.gemm(a.h2, wop(blk.mlp.fc1.weight), a.pre, bias=blk.mlp.fc1.bias)
            L.gelu_fwd(a.pre, a.hid)
        t = torch.empty(M, D, **f32)
        L.gemm(a.hid, wop(blk.mlp.fc2.weight), t, bias=blk.mlp.fc2.bias, residual=a.t_mid, row_scale=a.rs_mlp)
        s.layers.append(a)
    s.t_last = t
    out = torch.empty(M, D, **f32)
    L.layernorm(t, m.last_norm.weight, m.last_norm.bias, out, 1e-6)
    return out, s


def _weight_operands(m):
    """bf16 mode: the W and W^T operand copies of every matmul weight of the backbone, re-made by one launch per optimizer step
    (L.WeightOperands / whmr_weights_prepare) instead of a cast or a transposing cast per weight and direction (68 launches per step for ViT-B)"""
    wo = m.__dict__.get('_train_operands')
    if wo is None or wo.stale():
        pe = m.patch_embed.proj.weight
        ps = [pe] + [w for blk in m.blocks for w in (blk.attn.qkv.weight, blk.attn.proj.weight, blk.mlp.fc1.weight, blk.mlp.fc2.weight)]
        wo = m.__dict__['_train_operands'] = L.WeightOperands(ps, {id(pe): (pe.shape[0], pe.numel() // pe.shape[0])})
    return wo.refresh()


def _hip_attention_bwd(m, N):
    """The MFMA attention backward kernel covers the bf16 mode at head dim 64 and 64 < N <= 224 tokens (ViT-B/L at 224^2, 256x192)."""
    return m.numerics == 'bf16' and m.embed_dim // m.num_heads == 64 and 64 < N <= 224


def _attention_bwd(qkv, d_att, B, N, H, dh, scale, dt):
    """Backward of softmax(q k^T * scale) v per (image, head) outside the MFMA kernel's envelope (fp32 parity mode, or a bf16 run with N outside
    (64, 224]): the fp32 HIP kernel ``whmr_attention_bwd_f32`` (P recomputed from the saved qkv, deterministic); bf16 operands are widened first."""
    dq = L.attention_bwd_f32(qkv.float() if qkv.dtype != torch.float32 else qkv, d_att.float() if d_att.dtype != torch.float32 else d_att,
                             B, N, H, dh, scale)
    return dq if dt == torch.float32 else L.cast_bf16(dq)


# weight-gradient branch of the ViT backward on a side stream (A/B switch, OFF by default).  With the transposed-copy dW path it hid the
# memory-bound transposes beside the dX GEMMs (backbone-only step, 224^2, batch 64: 15.25 -> 14.23 ms); the TN kernel removes those
# transposes altogether (15.16 -> 14.15 ms in line) and then the second stream buys nothing (14.08) -- and inside the full W-HMR step it costs
# 0.3-0.6 ms (256x192, eager and graph-replayed).  The same treatment of the deconv / conv nodes measured neutral to -3 % and is not in the tree.
OVERLAP_DW = os.environ.get('WHMR_OVERLAP_DW', '0') != '0'
FUSE_GELU = os.environ.get('WHMR_FUSE_GELU', '1') != '0'      # bf16 mode: GELU inside fc1's epilogue (pre-activation kept as a second output) and its
                                                               # backward inside fc2's data-gradient epilogue (A/B switch)
USE_TN = os.environ.get('WHMR_TN_GEMM', '1') != '0'          # weight gradients on whmr_gemm_tn_bf16 (A/B switch; fp32 mode and odd shapes keep the transposed-copy path)
# the four weight gradients of a layer in ONE launch of the TN kernel (whmr_gemm_tn_bf16_group: two K slices instead of 7-28 per product, a quarter of
# the fp32 partial-tile traffic); A/B switch
GROUP_DW = os.environ.get('WHMR_TN_GROUP', '1') != '0'
def _side_stream(dev):
    return L.side_stream(dev, 2)


@torch.no_grad()
def vit_backward(m, s, dout):
    """dout: gradient of the [M, D] token output (after last_norm).  Returns {parameter: gradient} (fp32, parameter-shaped)."""
    B, N, Hp, Wp, M = s.dims
    D, dt, dev = m.embed_dim, _dt(m), dout.device
    f32 = dict(dtype=torch.float32, device=dev)
    grads = {}

    mpad = 64 if dt == torch.bfloat16 else 1     # the bf16 kernel needs K % 64 == 0; the token dimension is zero-padded up to it

    def wt(p, shape=None):                       # W^T in the compute dtype: the weight operand of dX = dY . W
        if getattr(s, 'ops', None) is not None:
            return s.ops[id(p)][1]
        w = p.detach()
        return L.transpose_cast(w.reshape(shape) if shape is not None else w, dt, pad_to=1)

    # The weight-gradient branch of every Linear (transpose dY (+ bias column sums), transpose X, dW = dY^T . X) only depends on dY and the saved
    # input, and nothing on the critical path (dX -> GELU / attention / LayerNorm backward -> next dX) waits for it: it runs on a SIDE stream.
    # Its memory-bound transposes then sit beside the MFMA-bound dX GEMMs of the main stream on the same CUs, and either branch fills the CUs the
    # other one's tile-grid tails leave idle.  Joined at the end of the node; same kernels, same bits.
    main = torch.cuda.current_stream(dev)
    side = _side_stream(dev) if OVERLAP_DW else None
    capturing = torch.cuda.is_current_stream_capturing()
    side_made = []
    if side is not None:
        side.wait_stream(main)

    pending = []                                                                       # weight-gradient products of the current layer (GROUP_DW)

    def run_pending():
        if not pending:
            return
        if len(pending) > 1 and L.gemm_tn_group_ok(pending) and L.gemm_tn_group(pending):
            pass
        else:
            for a_, b_, dw_, db_ in pending:
                L.gemm_tn(a_, b_, dw_, db=db_)
        pending.clear()

    def dw_branch(dy_op, x_saved, lin):
        w = lin.weight
        n_out, k_in = dy_op.shape[1], x_saved.shape[1]
        if USE_TN and L.gemm_tn_ok(dy_op, x_saved):
            # dW = dY^T . X straight from the token-major operands (gemm_tn.hip: transposing LDS reads) -- no transposed copies
            db = None
            if lin.bias is not None:                                                   # column sums of dY ride along in the same kernel
                db = torch.empty(n_out, **f32)
                grads[lin.bias] = db
                side_made.append(db)
            dw = torch.empty(n_out, k_in, **f32)
            if GROUP_DW and side is None and n_out % 256 == 0:
                pending.append((dy_op, x_saved, dw, db))                               # launched with the layer's other three (run_pending)
            else:
                L.gemm_tn(dy_op, x_saved, dw, db=db)
            grads[w] = dw.view_as(w)
            side_made.append(dw)
            return
        if lin.bias is not None:                                                       # db from the same pass that transposes dY
            db = torch.empty(n_out, **f32)
            dyt = L.transpose_colsum(dy_op, db, pad_to=mpad)                           # [Nout, Mpad]
            grads[lin.bias] = db
            side_made.append(db)
        else:
            dyt = L.transpose_cast(dy_op, dt, pad_to=mpad)
        xt = L.transpose_cast(x_saved, dt, pad_to=mpad)                                # [Kin, Mpad]
        dw = torch.empty(n_out, k_in, **f32)
        L.gemm(dyt, xt, dw)
        grads[w] = dw.view_as(w)
        side_made.append(dw)

    def linear_bwd(dy_op, x_saved, lin, need_dx=True, wshape=None, dx_dtype=torch.float32, gelu_bwd_of=None):
        """dy_op [M, Nout] (compute dtype), x_saved [M, Kin] -> dx [M, Kin] (fp32 unless dx_dtype); records dW, db.
        gelu_bwd_of: the pre-activation whose GELU produced x_saved -- dx is then multiplied by gelu'(pre) in the epilogue (= d pre)."""
        if side is None:
            dw_branch(dy_op, x_saved, lin)
        else:
            ev = torch.cuda.Event()
            ev.record(main)                                                            # dY (and the saved input) are complete on the main stream
            with torch.cuda.stream(side):
                side.wait_event(ev)
                dw_branch(dy_op, x_saved, lin)
            if not capturing:                                                          # main-stream tensors read by the side stream
                dy_op.record_stream(side)
                x_saved.record_stream(side)
        if not need_dx:
            return None
        w = lin.weight
        dx = torch.empty(M, x_saved.shape[1], dtype=dx_dtype, device=dev)
        L.gemm(dy_op, wt(w, wshape), dx, gelu_bwd_of=gelu_bwd_of)                      # W^T: [Kin, Nout]
        return dx

    # Early delivery to the data-parallel reducer (GradReducer.publish): everything in ``grads`` that belongs to finished blocks is handed over
    # in the order the parameters were registered (reverse = backward order inside a block), so that a full bucket's all-reduce runs on
    # RCCL's stream beside the backward of the earlier blocks.  Only when the gradients are produced on the node's own stream.
    sink = getattr(m, 'grad_sink', None) if side is None else None
    published = set()
    order = {id(q): i for i, q in enumerate(m.parameters())}

    def flush():
        if sink is None:
            return
        ready = sorted((q for q in grads if id(q) not in published), key=lambda q: -order[id(q)])
        for q in ready:
            g = grads[q]
            if sink(q, g if g.shape == q.shape else g.view_as(q)):
                published.add(id(q))

    dt_grad = torch.empty(M, D, **f32)                                                 # gradient of the fp32 residual stream
    dg, db = torch.empty(D, **f32), torch.empty(D, **f32)
    # bf16 mode: every LayerNorm backward also writes the compute-dtype operand of the NEXT branch's backward GEMMs (dx times that branch's
    # stochastic-depth row factor) -- the cast / scale_rows_cast pass per branch (22 launches of 10 us per step) is a by-product of a pass that
    # holds the row in registers anyway
    fuse_cast = FUSE_GELU and dt == torch.bfloat16
    rev = list(zip(reversed(list(m.blocks)), reversed(s.layers)))

    def operand():
        return torch.empty(M, D, dtype=dt, device=dev) if fuse_cast else None

    dy_next = operand()
    L.layernorm_bwd(s.t_last, dout.contiguous().float(), m.last_norm.weight, None, dt_grad, dg, db, 1e-6, cast_out=dy_next,
                    row_scale=rev[0][1].rs_mlp if (fuse_cast and rev) else None)
    grads[m.last_norm.weight], grads[m.last_norm.bias] = dg, db
    for bi, (blk, a) in enumerate(rev):
        # t_out = t_mid + fc2(gelu(fc1(LN2(t_mid))))
        # (stochastic depth: the branch sees mask / keep_prob * d t_out; the skip path sees d t_out unchanged)
        if fuse_cast:
            dy = dy_next
        else:
            dy = _op(dt_grad, dt) if a.rs_mlp is None else L.scale_rows_cast(dt_grad, a.rs_mlp, dt)
        if FUSE_GELU and dt == torch.bfloat16:
            d_pre = linear_bwd(dy, a.hid, blk.mlp.fc2, dx_dtype=dt, gelu_bwd_of=a.pre)        # fc2's data gradient * gelu'(pre) in one pass
        else:
            d_hid = linear_bwd(dy, a.hid, blk.mlp.fc2, dx_dtype=dt)                        # only feeds the GELU backward: compute dtype
            d_pre = torch.empty(M, a.pre.shape[1], dtype=dt, device=dev)
            L.gelu_bwd(a.pre, d_hid, d_pre)
        d_h2 = linear_bwd(d_pre, a.h2, blk.mlp.fc1)
        dg, db = torch.empty(D, **f32), torch.empty(D, **f32)
        dy_attn = operand()
        L.layernorm_bwd(a.t_mid, d_h2, blk.norm2.weight, dt_grad, dt_grad, dg, db, 1e-6, cast_out=dy_attn,
                        row_scale=a.rs_attn if fuse_cast else None)                      # dt_grad now = d t_mid
        grads[blk.norm2.weight], grads[blk.norm2.bias] = dg, db
        # t_mid = t_in + proj(attention(qkv(LN1(t_in))))
        if fuse_cast:
            dy = dy_attn
        else:
            dy = _op(dt_grad, dt) if a.rs_attn is None else L.scale_rows_cast(dt_grad, a.rs_attn, dt)
        d_att = linear_bwd(dy, a.att, blk.attn.proj)
        if a.lse is not None:
            d_qkv = torch.empty_like(a.qkv)
            L.attention_bwd(a.qkv, a.att, d_att, a.lse, d_qkv, B, N, m.num_heads, D // m.num_heads, m.scale)
        else:
            d_qkv = _attention_bwd(a.qkv, d_att, B, N, m.num_heads, D // m.num_heads, m.scale, dt)
        d_h1 = linear_bwd(d_qkv, a.h1, blk.attn.qkv)
        dg, db = torch.empty(D, **f32), torch.empty(D, **f32)
        dy_next = operand()                                                                # for the next (earlier) block's MLP branch, or the patch embedding
        L.layernorm_bwd(a.t_in, d_h1, blk.norm1.weight, dt_grad, dt_grad, dg, db, 1e-6, cast_out=dy_next,
                        row_scale=rev[bi + 1][1].rs_mlp if (fuse_cast and bi + 1 < len(rev)) else None)   # dt_grad now = d t_in
        grads[blk.norm1.weight], grads[blk.norm1.bias] = dg, db
        run_pending()
        flush()                                                                            # this block's gradients may start their exchange
    # t_0 = cols . Wp^T + b + (pos_embed[1:] + pos_embed[:1])
    pe = m.patch_embed.proj
    linear_bwd(dy_next if fuse_cast else _op(dt_grad, dt), s.cols, pe, need_dx=False)
    run_pending()
    dpos = torch.empty(N * D, **f32)
    L.colsum(dt_grad.view(B, N * D), dpos)                                            # sum over the batch
    dpos = dpos.view(N, D)
    gpe = torch.zeros_like(m.pos_embed, dtype=torch.float32)
    gpe[0, 1:] = dpos
    gpe[0, 0] = dpos.sum(0)
    grads[m.pos_embed] = gpe
    flush()
    for q in list(grads):                                                              # delivered already: the node returns None for them
        if id(q) in published:
            grads[q] = None
    if side is not None:                                                               # join: the gradients are consumed on the main stream
        main.wait_stream(side)
        if not capturing:
            for t in side_made:
                t.record_stream(main)
    return grads


class ViTFn(torch.autograd.Function):
    """tokens = ViTFn.apply(x, module, *module.parameters()): autograd node whose backward is ``vit_backward``."""

    @staticmethod
    def forward(ctx, x, module, *params):
        out, saved = vit_forward_train(module, x)
        ctx.module, ctx.saved, ctx.params = module, saved, params
        return out

    @staticmethod
    def backward(ctx, dout):
        grads = vit_backward(ctx.module, ctx.saved, dout)
        ctx.saved = None
        return (None, None) + tuple(grads.get(p) for p in ctx.params)
