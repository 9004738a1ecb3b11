"""ctypes binding of libwhmr_hip.so (C ABI: include/whmr_hip.h) + thin tensor-level wrappers.

PyTorch is used only for device memory and the current HIP stream.  There is NO fallback: if the shared
library is missing, or a tensor is not on a HIP device, these functions raise.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libwhmr_hip.so')
_lib = None


class WhmrGemm(C.Structure):
    _fields_ = [('A', C.c_void_p), ('W', C.c_void_p), ('C', C.c_void_p),
                ('bias', C.c_void_p), ('residual', C.c_void_p), ('zeros', C.c_void_p),
                ('M', C.c_int32), ('N', C.c_int32), ('K', C.c_int32),
                ('lda', C.c_int32), ('ldc', C.c_int32), ('ldr', C.c_int32),
                ('res_row_mod', C.c_int32), ('act', C.c_int32), ('out_bf16', C.c_int32), ('a_mode', C.c_int32),
                ('IH', C.c_int32), ('IW', C.c_int32), ('Cin', C.c_int32), ('OH', C.c_int32), ('OW', C.c_int32),
                ('KW', C.c_int32), ('SH', C.c_int32), ('SW', C.c_int32), ('PH', C.c_int32), ('PW', C.c_int32),
                ('c_mode', C.c_int32),
                ('c_off', C.c_int64), ('osb', C.c_int64), ('osy', C.c_int64), ('osx', C.c_int64),
                ('workspace', C.c_void_p), ('workspace_bytes', C.c_int64),
                ('n_phase', C.c_int32), ('epi_flags', C.c_int32),
                ('phase_w_stride', C.c_int64), ('phase_cy', C.c_int64), ('phase_cx', C.c_int64), ('split_k', C.c_int64), ('row_scale', C.c_void_p),
                ('C2', C.c_void_p)]


class WhmrGemmBlk(C.Structure):
    """struct whmr_gemm_blk_desc (include/whmr_hip.h): blocked-layout bf16 GEMM of the ViT inference path"""
    _fields_ = [('A', C.c_void_p), ('W', C.c_void_p), ('C', C.c_void_p), ('bias', C.c_void_p), ('res', C.c_void_p),
                ('M', C.c_int32), ('N', C.c_int32), ('K', C.c_int32), ('epi', C.c_int32), ('res_rows', C.c_int32), ('tile', C.c_int32),
                ('xhat', C.c_void_p), ('stats_out', C.c_void_p), ('stats_in', C.c_void_p), ('colsum', C.c_void_p), ('ln_eps', C.c_float),
                ('A_lo', C.c_void_p), ('W_lo', C.c_void_p), ('C_lo', C.c_void_p),
                ('shift', C.c_void_p), ('shift_stats', C.c_void_p), ('shift_out', C.c_void_p), ('xhat_lo', C.c_void_p)]


class WhmrSmplModel(C.Structure):
    _fields_ = [('v_template', C.c_void_p), ('shapedirs', C.c_void_p), ('posedirs', C.c_void_p),
                ('lbs_weights', C.c_void_p), ('J_template', C.c_void_p), ('J_shapedirs', C.c_void_p),
                ('J_regressor', C.c_void_p), ('J_regressor_extra', C.c_void_p), ('parents', C.c_void_p),
                ('extra_vertex_ids', C.c_void_p), ('joint_map', C.c_void_p), ('marker_ids', C.c_void_p),
                ('n_markers', C.c_int32)]


class WhmrStageTail(C.Structure):
    _fields_ = [('verts', C.c_void_p), ('posed_joints', C.c_void_p), ('regd', C.c_void_p), ('joints49', C.c_void_p), ('smpl_joints45', C.c_void_p),
                ('markers', C.c_void_p), ('R', C.c_int32),
                ('state', C.c_void_p), ('state_stride', C.c_int64), ('aa', C.c_void_p), ('Tz', C.c_void_p), ('bbox_h', C.c_void_p),
                ('center', C.c_void_p), ('orig_shape', C.c_void_p), ('focal0', C.c_float), ('res_w', C.c_float), ('res_h', C.c_float),
                ('theta', C.c_void_p), ('kp2d', C.c_void_p), ('kp2d_w', C.c_void_p), ('cam_t', C.c_void_p), ('focal', C.c_void_p),
                ('bbox_info', C.c_void_p), ('rotmat', C.c_void_p), ('xc_next', C.c_void_p), ('ld_next', C.c_int64), ('F_next', C.c_int32)]


class WhmrMafWeights(C.Structure):
    _fields_ = [('w0t', C.c_void_p), ('b0', C.c_void_p), ('w1t', C.c_void_p), ('b1', C.c_void_p),
                ('w2t', C.c_void_p), ('b2', C.c_void_p), ('w0b', C.c_void_p), ('w1b', C.c_void_p), ('w2b', C.c_void_p)]


class WhmrTnItem(C.Structure):
    """struct whmr_tn_item (include/whmr_hip.h): one product of a grouped weight-gradient launch"""
    _fields_ = [('A', C.c_void_p), ('lda', C.c_long), ('B', C.c_void_p), ('ldb', C.c_long), ('C', C.c_void_p), ('ldc', C.c_long),
                ('db', C.c_void_p), ('Mo', C.c_int32), ('No', C.c_int32)]


_P, _I, _F, _L = C.c_void_p, C.c_int, C.c_float, C.c_long
_SIGS = {
    'whmr_gemm_bf16': [C.POINTER(WhmrGemm), _I, _P],
    'whmr_gemm_f32': [C.POINTER(WhmrGemm), _I, _P],
    'whmr_gemm_f32_set_big': [_I],
    'whmr_gemm_tn_bf16': [_P, _L, _P, _L, _P, _L, _P, _I, _I, _I, _I, _P, _L, _P],
    'whmr_gemm_tn_bf16_group': [C.POINTER(WhmrTnItem), _I, _I, _P, _L, _P],
    'whmr_conv_dw_tn_bf16': [_P, _L, _P, _L, _P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _L, _P, _P],
    'whmr_conv_dw_tn2_bf16': [_P, _L, _P, _L, _P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _L, _P, _P],
    'whmr_tz_compose': [_P, _I, _P, _I, _P],
    'whmr_tz_compose_bwd': [_P, _I, _P, _P],
    'whmr_tz_unfold': [_P, _P, _I, _I, _I, _I, _I, _P],
    'whmr_gemm_bf16_big': [C.POINTER(WhmrGemm), _I, _P],
    'whmr_gemm_bf16_split': [C.POINTER(WhmrGemm), _I, _I, _P],
    'whmr_gemm_bf16_group': [_P, _I, _I, _P],
    'whmr_gemm_bf16_split_raw': [C.POINTER(WhmrGemm), _I, _I, _P],
    'whmr_set_option': [_I, _I],
    'whmr_gemm_blk': [C.POINTER(WhmrGemmBlk), _P],
    'whmr_gemm_blk_tile': [C.POINTER(WhmrGemmBlk), _I, _P],
    'whmr_gemm_blk_set_tile': [_I, _I],
    'whmr_gemm_blk_chain': [C.POINTER(WhmrGemmBlk), C.POINTER(WhmrGemmBlk), _P, _P, _P],
    'whmr_layernorm_blk': [_P, _P, _P, _P, _I, _I, _F, _I, _P],
    'whmr_patch_im2col_blk': [_P, _P, _I, _I, _I, _I, _I, _I, _L, _L, _L, _L, _P],
    'whmr_attention_blk': [_P, _P, _I, _I, _I, _F, _P],
    'whmr_split3_bf16': [_P, _P, _L, _I, _P],
    'whmr_cam_head': [_P, _I, _I, _F, _F, _F, _F, _I, _I, _P, _P, _P],
    'whmr_orient_state': [_P, _P, _L, _P, _L, _I, _I, _P],
    'whmr_orient_tail': [_P, _P, _P, _P, _P, _I, _P],
    'whmr_layernorm_blk_x3': [_P, _P, _P, _P, _P, _P, _I, _I, _F, _P],
    'whmr_layernorm_blk_mean': [_P, _P, _P, _P, _P, _I, _I, _F, _P],
    'whmr_patch_im2col_blk_x3': [_P, _P, _P, _I, _I, _I, _I, _I, _I, _L, _L, _L, _L, _P],
    'whmr_attention_blk_x3': [_P, _P, _P, _P, _I, _I, _I, _F, _P],
    'whmr_attention_x3_set_variant': [_I],
    'whmr_attention_set_variant': [_I],
    'whmr_layernorm': [_P, _P, _P, _P, _I, _I, _F, _I, _P],
    'whmr_patch_im2col': [_P, _P, _I, _I, _I, _I, _I, _I, _L, _L, _L, _L, _I, _P],
    'whmr_cast_f32_bf16': [_P, _P, _L, _P],
    'whmr_attention': [_P, _P, _I, _I, _I, _I, _F, _I, _P],
    'whmr_rot_to_mat': [_P, _P, _I, _I, _P],
    'whmr_mat_to_aa': [_P, _P, _I, _P],
    'whmr_mat_to_aa_bwd': [_P, _P, _P, _I, _P],
    'whmr_perspective': [_P, _P, _I, _P, _P, _I, _P, _P, _F, _P, _I, _I, _P],
    'whmr_weak_projection': [_P, _P, _P, _I, _I, _F, _F, _F, _P],
    'whmr_smpl_pose_chain': [C.POINTER(WhmrSmplModel), _P, _L, _P, _L, _I, _I, _P, _P, _P, _P, _P, _P],
    'whmr_regressor_post': [_P, _L, _P, _P, _P, _P, _P, _P, _I, _F, _F, _F, _P, _P, _P, _P, _P, _P],
    'whmr_smpl_skin': [C.POINTER(WhmrSmplModel), _P, _L, _P, _P, _P, _I, _P, _P],
    'whmr_smpl_stage_tail': [C.POINTER(WhmrSmplModel), C.POINTER(WhmrStageTail), _I, _P, _P],
    'whmr_smpl_stage_tail_csr': [C.POINTER(WhmrSmplModel), C.POINTER(WhmrStageTail), _P, _P, _P, _I, _P],
    'whmr_smpl_blend_skin': [C.POINTER(WhmrSmplModel), _P, _P, _L, _P, _P, _I, _P, _P],
    'whmr_smpl_blend_skin_x3': [C.POINTER(WhmrSmplModel), _P, _P, _L, _P, _P, _I, _P, _P],
    'whmr_smpl_blend_skin_stamps': [_P],
    'whmr_smpl_joints': [C.POINTER(WhmrSmplModel), _P, _P, _I, _P, _P, _P, _P, _P],
    'whmr_maf_sample': [_P, _I, _L, _L, _L, _L, _I, _I, _P, _P, _P, _L, _F, _F, _F, C.POINTER(WhmrMafWeights), _I, _I, _P, _L, _P, _P],
    'whmr_crop_normalize': [_P, _I, _I, _L, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    'whmr_attention_fwd_train': [_P, _P, _P, _I, _I, _I, _I, _F, _P],
    'whmr_attention_bwd': [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P],
    'whmr_tz_conv1': [_P, _I, _P, _P, _I, _I, _I, _P],
    'whmr_tz_fold': [_P, _I, _I, _I, _L, _P, _I, _I, _I, _I, _I, _P],
    'whmr_mfma_ceiling': [_I, _I, _P, _P, _P],
    'whmr_hbm_copy': [_P, _P, _L, _P],
    'whmr_clock_probe_begin': [_P, C.c_double, _P],
    'whmr_clock_probe_end': [_P, _P],
    'whmr_debug_lds_canary': [_I, _I, _I, _P, _P],
    'whmr_debug_global_canary': [_P, _I, _I, _I, _P, _P],
    'whmr_debug_pkfma_canary': [_I, _I, _P, _P],
    'whmr_debug_mfma32_stream': [_I, _I, _P, _P],
    'whmr_estimate_translation': [_P, _P, _I, _I, _I, _I, _F, _F, _F, _P, _P],
    'whmr_transpose_cast': [_P, _I, _L, _P, _I, _L, _I, _I, _I, _P],
    'whmr_colsum': [_P, _I, _L, _I, _I, _P, _I, _P, _P],
    'whmr_transpose_colsum': [_P, _L, _P, _L, _I, _I, _I, _P, _I, _P, _P],
    'whmr_layernorm_bwd': [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P, _P, _P, _P],
    'whmr_gelu_fwd': [_P, _P, _I, _L, _P],
    'whmr_gelu_bwd': [_P, _I, _P, _I, _P, _I, _L, _P],
    'whmr_regressor_state': [_P, _P, _L, _P, _L, _P, _L, _I, _P, _L, _I, _P],
    'whmr_tz_tail': [_P, _I, _I, _I, _P, _P, _I, _P, _P, _P, _F, _P, _P],
    'whmr_conv_im2col': [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _L, _L, _L, _L, _P],
    'whmr_bn_stats': [_P, _I, _L, _I, _P, _P, _F, _F, _P, _P, _P, _P, _P],
    'whmr_bn_apply_relu': [_P, _I, _P, _P, _I, _L, _I, _P],
    'whmr_bn_relu_bwd': [_P, _I, _P, _I, _P, _P, _I, _P, _P, _I, _L, _I, _P, _P],
    'whmr_bn_sums': [_P, _I, _L, _I, _P, _P, _P],
    'whmr_bn_stats_from_sums': [_P, _I, _P, _P, _F, _F, _P, _P, _P, _P],
    'whmr_bn_bwd_sums': [_P, _I, _P, _I, _P, _P, _P, _I, _L, _I, _P, _P, _P],
    'whmr_bn_bwd_apply': [_P, _I, _P, _I, _P, _P, _P, _P, _I, _L, _I, _P, _P],
    'whmr_im2col_t': [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _L, _P],
    'whmr_smpl_joints_bwd': [C.POINTER(WhmrSmplModel), _P, _P, _P, _I, _P, _P, _P, _P],
    'whmr_smpl_skin_bwd': [C.POINTER(WhmrSmplModel), _P, _L, _P, _P, _P, _P, _I, _I, _P, _P, _P],
    'whmr_smpl_chain_bwd': [C.POINTER(WhmrSmplModel), _P, _P, _L, _P, _P, _P, _I, _P, _P, _P],
    'whmr_maf_sample_bwd': [_P, _I, _L, _L, _L, _L, _I, _I, _P, _P, _P, _L, _F, _F, _F, C.POINTER(WhmrMafWeights), _P, _P, _P, _I, _I, _P, _L, _P, _I, _L, _L, _L, _L, _P, _P, _L, _P],
    'whmr_maf_scatter': [_P, _I, _I, _I, _P, _I, _L, _L, _L, _L, _P],
    'whmr_col2im': [_P, _I, _L, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    'whmr_csr_apply3': [_P, _P, _P, _P, _I, _P, _I, _I, _I, _P],
    'whmr_regressor_post_train': [_P, _P, _P, _P, _P, _P, _I, _I, _F, _F, _F, _P, _P, _P, _P, _P],
    'whmr_regressor_post_train_bwd': [_P, _P, _P, _P, _P, _P, _I, _I, _F, _F, _F, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    'whmr_attention_bwd_f32': [_P, _P, _P, _P, _I, _I, _I, _I, _F, _P],
    'whmr_scale_rows_cast': [_P, _P, _P, _I, _I, _I, _P],
    'whmr_iuv_rasterize': [_P, _I, _I, _P, _I, _P, _I, _P, _P, _F, _F, _F, _F, _F, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    'whmr_iuv_losses': [_P, _I, _L, _P, _L, _L, _L, _L, _I, _I, _I, _F, _P, _P, _P],
    'whmr_iuv_losses_bwd': [_P, _I, _L, _P, _L, _L, _L, _L, _I, _I, _I, _F, _P, _P, _L, _P],
    'whmr_weights_prepare': [_P, _I, _I, _P],
    'whmr_maxpool_nhwc': [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    'whmr_avgpool_nhwc': [_P, _P, _I, _I, _I, _I, _P],
}
EXPORTS = tuple(_SIGS)


def lib():
    """Load the HIP library (once).  Raises if it has not been built -- there is no CPU path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError('%s not found: build it with `python -m whmr_amd.build` (hipcc, gfx950). '
                              'The W-HMR hot path has no non-HIP fallback.' % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, args in _SIGS.items():
            fn = getattr(l, name)
            fn.argtypes = args
            fn.restype = C.c_int
        _lib = l
        for slot, key in enumerate(('WHMR_BLK_TILE_QKV', 'WHMR_BLK_TILE_PROJ', 'WHMR_BLK_TILE_FC1', 'WHMR_BLK_TILE_FC2')):     # A/B: force a tile per ViT shape
            if os.environ.get(key):
                l.whmr_gemm_blk_set_tile(slot, int(os.environ[key], 0))
        # the split-bf16 attention's variant is ONE word (bit 0: old kernel, bits 4-7: stagger + 1): compose both switches before the single call, so
        # that WHMR_ATTN_X3_STAGGER does not silently clear WHMR_ATTN_OLD for the bf16x3 kernel only
        x3_variant = 0
        if os.environ.get('WHMR_ATTN_OLD', '0') != '0':         # A/B: the blocked attention (bf16 and bf16x3) on the round-2 / round-3 kernels
            l.whmr_attention_set_variant(1 | 16)
            x3_variant |= 1
        if os.environ.get('WHMR_ATTN_X3_STAGGER'):               # A/B: k half-microseconds between the CU quarters of the split-bf16 attention (default: by shape)
            x3_variant |= (int(os.environ['WHMR_ATTN_X3_STAGGER']) + 1) << 4
        if x3_variant:
            l.whmr_attention_x3_set_variant(x3_variant)
    return _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _check(err, what):
    if err != 0:
        raise RuntimeError('%s failed: hipError %d' % (what, err))


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError('whmr_amd kernels need HIP device tensors (got %s); there is no CPU fallback' % t.device)


_zeros = {}
_splitk_ws = {}
_SIDE_STREAMS = {}


def side_stream(device, slot):
    """The package's side streams: FOUR per device, created together at the first request and shared by every user (slot 0: the heavy chain of the
    forward / training graph, 1: camera head | Tz head's tail, 2: the ViT backward's weight-gradient stream, graph warm-ups, probes of bench.py,
    3: ClockProbe -- nothing else, its wave spins until it is released).  torch hands every ``torch.cuda.Stream()`` another of its 32 pooled HIP
    streams and HIP spreads those over a few hardware queues: a process that had created a dozen streams (the default bench.py run: graphs, camera
    streams, probes) ran the training step 0.6 ms slower than a fresh one -- its side streams shared hardware queues.  Roles that share a slot never
    run at the same time; sharing only adds ordering."""
    device = torch.device(device)
    if device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    pool = _SIDE_STREAMS.get(device)
    if pool is None:
        pool = _SIDE_STREAMS[device] = [torch.cuda.Stream(device=device) for _ in range(4)]
    return pool[slot]


def splitk_workspace(device):
    """Scratch for split-K partial sums (fp32 skinny GEMMs; bf16 GEMMs with too few tiles to fill 256 CUs): one buffer per
    (device, stream) -- GEMMs of two streams may run concurrently (WHMR.forward overlaps the regressor loop with the deconvs)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    w = _splitk_ws.get(key)
    if w is None:
        w = _splitk_ws[key] = torch.empty(128 << 20, dtype=torch.uint8, device=device)
    return w


def zero_page(device):
    z = _zeros.get(device)
    if z is None:
        z = _zeros[device] = torch.zeros(1024, dtype=torch.uint8, device=device)
    return z


ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2
PROFILE = None        # bench.py sets this to a list: (kernel, flops, start_event, end_event) per GEMM launch


def gemm(a, w, out, bias=None, residual=None, res_row_mod=0, act=ACT_NONE, conv=None, scatter=None, M=None,
         lda=None, glds=True, tile=None, phases=None, res_first=False, splits=None, row_scale=None, accumulate=False,
         trans_a=False, trans_w=False, pre_out=None, gelu_bwd_of=None, split3_out=None, split_parts=3, raw_splits=None, desc_only=False):
    """out = act(a . w^T + bias) + residual.  a/w dtype selects the kernel (bf16 MFMA or exact-fp32 MFMA).

    conv = dict(IH, IW, Cin, OH, OW, KW, SH, SW, PH, PW): a is an NHWC image [B, IH, IW, Cin], rows are output pixels.
    scatter = dict(c_off, osb, osy, osx): row (b, oy, ox) is written at that offset of ``out`` (needs conv dims).
    accumulate: out += a . w^T (+ bias) in place -- ``out`` is its own residual, also in scatter mode (bf16 kernel: epi_flags bit 2).
    trans_a / trans_w (fp32, at most 1024 output rows): the operand is given REDUCTION-MAJOR -- a as [K, M], w as [K, N] (dense) -- so the
    backward products of nn.Linear (dX = dY . W, dW = dY^T . X) read dY / X / W as they are (epi_flags bits 4 / 5, skinny kernel).
    """
    if pre_out is not None:                 # bf16 kernel, act = GELU: out = gelu(z), pre_out = z = a . w^T + bias (the training forward keeps both)
        _dev(pre_out)
        assert act == ACT_GELU and residual is None and out.dtype == torch.bfloat16 and pre_out.dtype == torch.bfloat16
        assert pre_out.shape == out.shape and pre_out.stride() == out.stride() and conv is None and scatter is None
    if gelu_bwd_of is not None:             # bf16 kernel: out = (a . w^T) * gelu'(z), z = gelu_bwd_of (the GELU backward inside fc2's data gradient)
        assert residual is None and act == ACT_NONE and out.dtype == torch.bfloat16 and gelu_bwd_of.dtype == torch.bfloat16
        assert gelu_bwd_of.shape == out.shape and conv is None and scatter is None and not accumulate
        residual = gelu_bwd_of
    if trans_a or trans_w:
        assert a.dtype == torch.float32 and conv is None and scatter is None and phases is None and tile is None and lda is None and M is None
        assert a.dim() == 2 and w.dim() == 2 and a.stride(1) == 1 and w.is_contiguous()
        assert (a.shape[0] if trans_a else a.shape[1]) == (w.shape[0] if trans_w else w.shape[1])
    if accumulate:
        assert residual is None and row_scale is None and act == ACT_NONE and out.dtype == a.dtype
        assert scatter is None or a.dtype == torch.bfloat16, 'the scattered in-place accumulate exists in the bf16 kernel only'
        residual = out
    _dev(a, w, out, bias, residual)
    assert a.dtype == w.dtype and a.dtype in (torch.bfloat16, torch.float32)
    assert a.is_contiguous() or lda is not None or trans_a or trans_w
    assert w.is_contiguous() and w.dim() == (3 if phases else 2)
    N, K = (w.shape[1], w.shape[0]) if trans_w else w.shape[-2:]
    p = WhmrGemm()
    p.A, p.W, p.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
    p.bias = bias.data_ptr() if bias is not None else None
    p.residual = residual.data_ptr() if residual is not None else None
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == N
    if residual is not None:
        assert residual.dtype in (torch.float32, torch.bfloat16) and residual.stride(-1) == 1
        p.ldr = residual.stride(-2)
        p.epi_flags = (1 if residual.dtype == torch.bfloat16 else 0) | (2 if res_first else 0)
        if accumulate and scatter is not None:
            p.ldr, p.epi_flags = 8, p.epi_flags | 4                     # the residual is addressed like the scattered output
        assert residual.dtype == torch.float32 or a.dtype == torch.bfloat16, 'bf16 residuals exist in the bf16 kernel only'
    p.res_row_mod = res_row_mod
    p.act = act
    if row_scale is not None:           # out = residual + row_scale[m] * act(a . w^T + bias): stochastic depth (vit.py:132-139)
        _dev(row_scale)
        assert row_scale.dtype == torch.float32 and row_scale.is_contiguous() and not res_first and out.dtype == torch.float32
        p.row_scale = row_scale.data_ptr()
    assert out.dtype in (torch.bfloat16, torch.float32)
    p.out_bf16 = int(out.dtype == torch.bfloat16)
    if conv is not None:
        p.a_mode = 1
        for k in ('IH', 'IW', 'Cin', 'OH', 'OW', 'KW', 'SH', 'SW', 'PH', 'PW'):
            setattr(p, k, conv[k])
        B = a.shape[0]
        p.M = B * conv['OH'] * conv['OW'] if M is None else M
        assert K % conv['Cin'] == 0
        if conv.get('chunk_major'):          # weight columns ordered (ci chunk of 64, ky, kx, ci in chunk)
            assert a.dtype == torch.bfloat16
            p.epi_flags |= 8
        p.zeros = zero_page(a.device).data_ptr()
    elif trans_a:
        p.M, p.lda = a.shape[1], a.stride(0)
    else:
        p.M = (a.numel() // a.shape[-1]) if M is None else M
        p.lda = a.stride(-2) if lda is None else lda
        assert a.shape[-1] == K
    p.N, p.K = N, K
    p.epi_flags |= (16 if trans_a else 0) | (32 if trans_w else 0) | (128 if gelu_bwd_of is not None else 0)
    p.C2 = pre_out.data_ptr() if pre_out is not None else None
    if split3_out is not None:              # bf16 kernel, fp32 out: also write the [hi | lo | hi] split-bf16 operand form of `out` (3 N channels per row / pixel)
        _dev(split3_out)
        assert a.dtype == torch.bfloat16 and out.dtype == torch.float32 and pre_out is None and split3_out.dtype == torch.bfloat16
        assert split_parts in (2, 3) and split3_out.is_contiguous() and split3_out.numel() == split_parts * out.numel() and out.is_contiguous() and N % 4 == 0
        p.C2, p.epi_flags = split3_out.data_ptr(), p.epi_flags | 256 | (512 if split_parts == 2 else 0)       # 2 parts: [hi | lo] only
    if phases is not None:                  # dict(cy, cx): 4 stacked phase matrices w[4, N, K] (sub-pixel deconv)
        assert a.dtype == torch.bfloat16 and conv is not None and scatter is not None and w.shape[0] == 4
        p.n_phase, p.phase_w_stride, p.phase_cy, p.phase_cx = 4, N * K, phases['cy'], phases['cx']
    if scatter is not None:
        p.c_mode = 1
        for k in ('c_off', 'osb', 'osy', 'osx'):
            setattr(p, k, scatter[k])
    else:
        p.ldc = out.stride(-2) if out.dim() >= 2 else N
    if desc_only:                           # -> the descriptor (for gemm_group); the caller keeps the tensors alive until the launch
        return p
    fn = lib().whmr_gemm_bf16 if a.dtype == torch.bfloat16 else lib().whmr_gemm_f32
    if scatter is None and phases is None and (a.dtype == torch.bfloat16 or conv is None):
        ws = splitk_workspace(a.device)
        p.workspace, p.workspace_bytes = ws.data_ptr(), ws.numel()
    if raw_splits:                          # bf16 kernel: `out` = [raw_splits, M, N] fp32 planes of raw split-K partial sums (no finishing pass)
        assert a.dtype == torch.bfloat16 and tile is not None and out.dtype == torch.float32 and out.is_contiguous() and out.numel() == raw_splits * p.M * N
        ev = _profile_begin()
        _check(lib().whmr_gemm_bf16_split_raw(C.byref(p), int(tile), int(raw_splits), _stream()), 'whmr_gemm_bf16_split_raw')
        _profile_end(ev, 'gemm_bf16', 2.0 * p.M * p.N * p.K)
        return out
    if tile is not None:                    # explicit tile id (A/B tests), see gemm_bf16_big.hip
        assert a.dtype == torch.bfloat16
        if splits and splits > 1:
            _check(lib().whmr_gemm_bf16_split(C.byref(p), int(tile), int(splits), _stream()), 'whmr_gemm_bf16_split')
            return out
        _check(lib().whmr_gemm_bf16_big(C.byref(p), int(tile), _stream()), 'whmr_gemm_bf16_big')
        return out
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _check(fn(C.byref(p), 0 if glds else 1, _stream()), 'whmr_gemm')
        e1.record()
        PROFILE.append(('gemm_bf16' if a.dtype == torch.bfloat16 else 'gemm_f32', 2.0 * p.M * p.N * p.K * max(1, p.n_phase), e0, e1))
        return out
    _check(fn(C.byref(p), 0 if glds else 1, _stream()), 'whmr_gemm')
    return out


def gemm_group(descs, tile=192):
    """<= 9 gathering bf16 GEMMs (descriptors of ``gemm(..., conv=..., desc_only=True)``, bf16 output) as ONE launch of one tile shape
    (whmr_gemm_bf16_group: 192 = 192 x 256 x 64 tiles, 65 = 128 x 64 x 64)."""
    arr = (WhmrGemm * len(descs))(*descs)
    ev = _profile_begin()
    _check(lib().whmr_gemm_bf16_group(arr, len(descs), int(tile), _stream()), 'whmr_gemm_bf16_group')
    _profile_end(ev, 'gemm_bf16', sum(2.0 * d.M * d.N * d.K for d in descs))


# ---- blocked layouts (gemm_blk.hip): [R, C] stored as [ceil(R/32)][C/E][32][E], E = 8 (bf16) / 4 (fp32) -----------------------------
EPI_BF16, EPI_BF16_GELU, EPI_F32_RES, EPI_F32_POS = 0, 1, 2, 3


def to_blocked(t):
    """row-major [R, C] (bf16 or fp32) -> blocked copy (rows padded to a multiple of 32 with zeros).  Host-side packing of weights / test data."""
    _dev(t)
    R, Cc = t.shape
    E = 8 if t.dtype == torch.bfloat16 else 4
    assert t.dtype in (torch.bfloat16, torch.float32) and Cc % E == 0
    Rp = (R + 31) // 32 * 32
    if Rp != R:
        t = torch.cat([t, t.new_zeros(Rp - R, Cc)], 0)
    return t.reshape(Rp // 32, 32, Cc // E, E).permute(0, 2, 1, 3).contiguous()


def from_blocked(t, R):
    """blocked [R/32][C/E][32][E] -> row-major [R, C] copy (tests / hand-over to row-major consumers)"""
    nb, nu, _, E = t.shape
    return t.permute(0, 2, 1, 3).reshape(nb * 32, nu * E)[:R].contiguous()


def split_bf16(t):
    """fp32 tensor -> (hi, lo) bf16 pair with hi + lo = t to 16 significand bits (the operand pair of the bf16x3 numerics).  Weight preparation
    and test data; activations are split inside the kernels that produce them."""
    t = t.float()
    hi = t.bfloat16()
    return hi, (t - hi.float()).bfloat16()


def gemm_blk(a, w, out, M, **kw):
    """out = epi(a . w^T + bias [+ res]) on blocked operands: a [M/32][K/8][32][8] bf16, w [N/32][K/8][32][8] bf16,
    out bf16 [M/32][N/8][32][8] (epi 0/1) or fp32 [M/32][N/4][32][4] (epi 2: + blocked res, may be `out`; epi 3: + res[m % res_rows] row-major).
    a_lo / w_lo (/ out_lo for epi 0/1): the lo halves of split-bf16 operands -> the bf16x3 kernel (three MFMAs per product, fp32-grade result).
    Keyword arguments: see ``_blk_desc``."""
    p, x3, N, K = _blk_desc(a, w, out, M, **kw)
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _check(lib().whmr_gemm_blk(C.byref(p), _stream()), 'whmr_gemm_blk')
        e1.record()
        PROFILE.append(('gemm_bf16x3' if x3 else 'gemm_bf16', 2.0 * M * N * K, e0, e1))
        return out
    _check(lib().whmr_gemm_blk(C.byref(p), _stream()), 'whmr_gemm_blk')
    return out


_CHAIN_SCRATCH = {}


def gemm_blk_chain(fc1, fc2):
    """PILOT (round 6): the fc1 -> fc2 pair of a transformer layer as ONE persistent launch (whmr_gemm_blk_chain).  fc1 / fc2: dicts of gemm_blk's
    arguments (a, w, out, M, + keywords); fc2['a'] must be fc1['out'].  -> True when the pair was launched that way, False when the C entry does not
    take the pair (the caller then issues the two launches).  The device-side error flag (a wait that hit its limit) is readable as ``chain_error(dev)``."""
    f1, f2 = dict(fc1), dict(fc2)
    p1, x31, N1, K1 = _blk_desc(f1.pop('a'), f1.pop('w'), f1.pop('out'), f1.pop('M'), **f1)
    p2, x32, N2, K2 = _blk_desc(f2.pop('a'), f2.pop('w'), f2.pop('out'), f2.pop('M'), **f2)
    if x31 or x32:
        return False
    dev = fc1['a'].device
    sc = _CHAIN_SCRATCH.get(dev)
    if sc is None:
        sc = _CHAIN_SCRATCH[dev] = (torch.zeros(4096, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev))
    if (p1.M + 319) // 320 > sc[0].numel():
        return False
    ev = _profile_begin()
    rc = lib().whmr_gemm_blk_chain(C.byref(p1), C.byref(p2), sc[0].data_ptr(), sc[1].data_ptr(), _stream())
    if rc == 1:                                  # hipErrorInvalidValue: not a pair the pilot takes, nothing was launched
        return False
    _check(rc, 'whmr_gemm_blk_chain')
    if ev is not None:                           # one launch, two products: booked as two GEMM rows of half the span each
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        PROFILE.append(('gemm_bf16_chain', 2.0 * p1.M * (N1 * K1 + N2 * K2), ev, e1))
    return True


def chain_error(device):
    """1 when a wait of a chained launch on ``device`` ever ran into its time limit (synchronises)"""
    sc = _CHAIN_SCRATCH.get(device)
    return 0 if sc is None else int(sc[1].item())


def _blk_desc(a, w, out, M, bias=None, epi=EPI_BF16, res=None, res_rows=0, tile=0, xhat=None, stats_out=None, stats_in=None, colsum=None,
              ln_eps=1e-6, a_lo=None, w_lo=None, out_lo=None, shift=None, shift_stats=None, shift_out=None, xhat_lo=None):
    """-> (whmr_gemm_blk_desc, split-bf16?, N, K) for gemm_blk's arguments (shape / dtype checks included)"""
    _dev(a, w, out, bias, res)
    assert a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and a.is_contiguous() and w.is_contiguous() and out.is_contiguous()
    N, K = w.shape[0] * 32, w.shape[1] * 8
    assert a.shape[1] * 8 == K and a.shape[0] * 32 >= M and out.shape[0] * 32 >= M
    assert out.dtype == (torch.float32 if epi >= 2 else torch.bfloat16) and out.shape[1] * out.shape[3] == N
    p = WhmrGemmBlk()
    p.A, p.W, p.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
    p.bias = bias.data_ptr() if bias is not None else None
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == N
    if res is not None:
        assert res.dtype == torch.float32 and res.is_contiguous()
        p.res = res.data_ptr()
    p.M, p.N, p.K, p.epi, p.res_rows, p.tile = M, N, K, epi, res_rows, tile
    if xhat is not None:                     # LayerNorm folding, producer side: bf16 copy of the stream + row partial sums [rows, N/64, 2]
        _dev(xhat, stats_out)
        assert epi >= 2 and xhat.dtype == torch.bfloat16 and xhat.is_contiguous() and xhat.shape[0] * 32 >= M and xhat.shape[1] * 8 == N
        assert stats_out.dtype == torch.float32 and stats_out.is_contiguous() and stats_out.numel() >= a.shape[0] * 32 * (N // 256) * 2
        p.xhat, p.stats_out = xhat.data_ptr(), stats_out.data_ptr()
        rows = a.shape[0] * 32
        for name, t_, need in (('shift', shift, rows), ('shift_stats', shift_stats, rows * (N // 256) * 2), ('shift_out', shift_out, rows)):
            if t_ is not None:               # per-row shift of the folded LayerNorm (see include/whmr_hip.h)
                _dev(t_)
                assert t_.dtype == torch.float32 and t_.is_contiguous() and t_.numel() >= need, name
                setattr(p, name, t_.data_ptr())
        assert shift_stats is None or shift_stats.data_ptr() != stats_out.data_ptr()
    if stats_in is not None:                 # consumer side: a = xhat of the raw stream, w = gamma-scaled weights, bias = b + W.beta
        _dev(stats_in, colsum)
        assert epi < 2 and stats_in.dtype == torch.float32 and colsum.dtype == torch.float32 and colsum.numel() == N and K % 256 == 0
        p.stats_in, p.colsum, p.ln_eps = stats_in.data_ptr(), colsum.data_ptr(), ln_eps
    x3 = a_lo is not None
    if x3:
        _dev(a_lo, w_lo, out_lo)
        assert a_lo.dtype == torch.bfloat16 and a_lo.shape == a.shape and a_lo.is_contiguous() and w_lo is not None and w_lo.shape == w.shape and w_lo.is_contiguous()
        p.A_lo, p.W_lo = a_lo.data_ptr(), w_lo.data_ptr()
        if xhat is not None:
            _dev(xhat_lo)
            assert xhat_lo is not None and xhat_lo.shape == xhat.shape and xhat_lo.dtype == torch.bfloat16 and xhat_lo.is_contiguous()
            p.xhat_lo = xhat_lo.data_ptr()
        if epi < 2:
            assert out_lo is not None and out_lo.shape == out.shape and out_lo.dtype == torch.bfloat16 and out_lo.is_contiguous()
            p.C_lo = out_lo.data_ptr()
    p._keep = (a, w, out, bias, res, xhat, stats_out, stats_in, colsum, a_lo, w_lo, out_lo, shift, shift_stats, shift_out, xhat_lo)
    return p, x3, N, K


def layernorm_blk(x, weight, bias, out, rows, eps, out_std=False, mean_out=None):
    """LayerNorm of a blocked fp32 stream -> blocked bf16 operand (out_std False) or row-major fp32 [rows, C] (out_std True);
    mean_out (bf16 operand form only): fp32 [ceil(rows/32)*32] receives the row means"""
    _dev(x, weight, bias, out)
    assert x.dtype == torch.float32 and x.is_contiguous() and out.is_contiguous()
    Cdim = x.shape[1] * 4
    assert out.dtype == (torch.float32 if out_std else torch.bfloat16)
    if mean_out is not None:
        _dev(mean_out)
        assert not out_std and mean_out.dtype == torch.float32 and mean_out.is_contiguous() and mean_out.numel() >= x.shape[0] * 32
        _check(lib().whmr_layernorm_blk_mean(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), out.data_ptr(), mean_out.data_ptr(), rows, Cdim, eps,
                                             _stream()), 'whmr_layernorm_blk_mean')
        return out
    ev = _profile_begin()
    _check(lib().whmr_layernorm_blk(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), out.data_ptr(), rows, Cdim, eps, int(out_std), _stream()),
           'whmr_layernorm_blk')
    _profile_end(ev, 'layernorm_bytes', float(rows) * Cdim * (4 + out.element_size()))          # fp32 stream read once, operand / output written once
    return out


def patch_im2col_blk(x, out, patch, pad, out_lo=None):
    _dev(x, out)
    assert x.dtype == torch.float32 and x.dim() == 4 and out.dtype == torch.bfloat16 and out.is_contiguous()
    B, Cin, H, W = x.shape
    sb, sc, sh, sw = x.stride()
    if out_lo is not None:                   # bf16x3: pixels as a hi / lo pair
        _dev(out_lo)
        assert out_lo.dtype == torch.bfloat16 and out_lo.shape == out.shape and out_lo.is_contiguous()
        _check(lib().whmr_patch_im2col_blk_x3(x.data_ptr(), out.data_ptr(), out_lo.data_ptr(), B, Cin, H, W, patch, pad, sb, sc, sh, sw, _stream()),
               'whmr_patch_im2col_blk_x3')
        return out
    ev = _profile_begin()
    _check(lib().whmr_patch_im2col_blk(x.data_ptr(), out.data_ptr(), B, Cin, H, W, patch, pad, sb, sc, sh, sw, _stream()), 'whmr_patch_im2col_blk')
    _profile_end(ev, 'patch_gather_bytes', float(x.numel()) * 4 + float(out.numel()) * 2)             # image read once (fp32), tokens written once (bf16)
    return out


def attention_blk(qkv, out, B, N, H, scale, qkv_lo=None, out_lo=None):
    _dev(qkv, out)
    assert qkv.is_contiguous() and out.is_contiguous() and qkv.dtype == torch.bfloat16 and out.dtype == torch.bfloat16
    if qkv_lo is not None:                   # bf16x3: operand / result pairs
        _dev(qkv_lo, out_lo)
        assert qkv_lo.shape == qkv.shape and qkv_lo.dtype == torch.bfloat16 and qkv_lo.is_contiguous()
        assert out_lo.shape == out.shape and out_lo.dtype == torch.bfloat16 and out_lo.is_contiguous()
        ev = _profile_begin()
        _check(lib().whmr_attention_blk_x3(qkv.data_ptr(), qkv_lo.data_ptr(), out.data_ptr(), out_lo.data_ptr(), B, N, H, scale, _stream()),
               'whmr_attention_blk_x3')
        _profile_end(ev, 'attention', 4.0 * B * H * N * N * 64)              # QK^T + PV, head dim 64 (algorithmic flops; three MFMAs per product issued)
        return out
    ev = _profile_begin()
    _check(lib().whmr_attention_blk(qkv.data_ptr(), out.data_ptr(), B, N, H, scale, _stream()), 'whmr_attention_blk')
    _profile_end(ev, 'attention', 4.0 * B * H * N * N * 64)
    return out


def cam_head(logits, D, pitch_range, roll_range, B):
    """logits [Bf, 3 D] fp32 (vfov | pitch | roll bins) -> (cam_rotmat, render_rotmat) [B, 3, 3]; Bf == 1 is broadcast (one launch)"""
    _dev(logits)
    assert logits.dtype == torch.float32 and logits.stride(1) == 1 and logits.shape[1] >= 3 * D
    Bf = logits.shape[0]
    cam = torch.empty(B, 3, 3, dtype=torch.float32, device=logits.device)
    ren = torch.empty(B, 3, 3, dtype=torch.float32, device=logits.device)
    _check(lib().whmr_cam_head(logits.data_ptr(), logits.stride(0), D, pitch_range[0], pitch_range[1], roll_range[0], roll_range[1], Bf, B,
                               cam.data_ptr(), ren.data_ptr(), _stream()), 'whmr_cam_head')
    return cam, ren


def orient_state(cam_rotmat, rotmat, xc, F):
    """xc[:, F:F+15] = [rot6d(cam_rotmat) | rotmat[:, 0]] (whmr.py:295-297), one launch; rotmat [B, 24, 3, 3] (any batch stride)"""
    _dev(cam_rotmat, rotmat, xc)
    B = xc.shape[0]
    cam_rotmat = _f32c(cam_rotmat.contiguous())
    assert rotmat.dtype == torch.float32 and rotmat[0].is_contiguous() and xc.dtype == torch.float32 and xc.stride(1) == 1 and xc.shape[1] >= F + 15
    _check(lib().whmr_orient_state(cam_rotmat.data_ptr(), rotmat.data_ptr(), rotmat.stride(0), xc.data_ptr(), xc.stride(0), F, B, _stream()), 'whmr_orient_state')


def orient_tail(r, pose_aa, rotmat):
    """r [B, 9], pose_aa [B, 72], rotmat [B, 24, 3, 3] (contiguous) -> (g_pose [B, 72], g_rotmat [B, 24, 3, 3]) in one launch (whmr.py:301-305,630-640)"""
    _dev(r, pose_aa, rotmat)
    B = r.shape[0]
    assert r.is_contiguous() and pose_aa.is_contiguous() and rotmat.is_contiguous() and r.dtype == pose_aa.dtype == rotmat.dtype == torch.float32
    g_pose = torch.empty(B, 72, dtype=torch.float32, device=r.device)
    g_rot = torch.empty(B, 24, 3, 3, dtype=torch.float32, device=r.device)
    _check(lib().whmr_orient_tail(r.data_ptr(), pose_aa.data_ptr(), rotmat.data_ptr(), g_pose.data_ptr(), g_rot.data_ptr(), B, _stream()), 'whmr_orient_tail')
    return g_pose, g_rot


def split3(x):
    """fp32 [..., C] (contiguous) -> bf16 [..., 3C] = [hi | lo | hi]: the K-concatenated activation operand of the bf16x3 numerics for the
    row-major / convolution GEMMs (weights: ``split3_weight``)"""
    _dev(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    Cc = x.shape[-1]
    out = torch.empty(*x.shape[:-1], 3 * Cc, dtype=torch.bfloat16, device=x.device)
    _check(lib().whmr_split3_bf16(x.data_ptr(), out.data_ptr(), x.numel() // Cc, Cc, _stream()), 'whmr_split3_bf16')
    return out


def split3_weight(w):
    """fp32 weight [..., C] -> bf16 [..., 3C] = [W_hi | W_hi | W_lo] along the last axis (the partner of ``split3``)"""
    hi, lo = split_bf16(w)
    return torch.cat([hi, hi, lo], dim=-1).contiguous()


def layernorm_blk_x3(x, weight, bias, out_hi, out_lo, rows, eps, mean_out=None):
    """LayerNorm of a blocked fp32 stream -> blocked split-bf16 operand pair (bf16x3 numerics); mean_out: fp32 [ceil(rows/32)*32] row means"""
    _dev(x, weight, bias, out_hi, out_lo, mean_out)
    assert x.dtype == torch.float32 and x.is_contiguous() and out_hi.is_contiguous() and out_lo.is_contiguous()
    assert out_hi.dtype == torch.bfloat16 and out_lo.dtype == torch.bfloat16 and out_hi.shape == out_lo.shape
    assert mean_out is None or (mean_out.dtype == torch.float32 and mean_out.is_contiguous() and mean_out.numel() >= x.shape[0] * 32)
    _check(lib().whmr_layernorm_blk_x3(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), out_hi.data_ptr(), out_lo.data_ptr(), _ptr(mean_out), rows,
                                       x.shape[1] * 4, eps, _stream()), 'whmr_layernorm_blk_x3')
    return out_hi


def scale_rows_cast(src, scale, dtype):
    """(dtype)(scale[m] * src[m, :]) for an fp32 [M, C] matrix -- stochastic-depth mask on a gradient, fused with the operand cast"""
    _dev(src, scale)
    assert src.dtype == torch.float32 and src.is_contiguous() and scale.dtype == torch.float32 and scale.numel() == src.shape[0]
    dst = torch.empty(src.shape, dtype=dtype, device=src.device)
    _check(lib().whmr_scale_rows_cast(src.data_ptr(), scale.data_ptr(), dst.data_ptr(), src.shape[0], src.shape[1], int(dtype == torch.bfloat16),
                                      _stream()), 'whmr_scale_rows_cast')
    return dst


def iuv_rasterize(verts, faces, tex, cam, K, focal, orig_size, out_size, vmap=None, want_faces=False):
    """IUV image [B, 3, H, W] of the meshes ``verts`` [B, Vsrc, 3] (see include/whmr_hip.h: whmr_iuv_rasterize).  K = (fx, fy, px, py)."""
    _dev(verts, faces, tex, cam, vmap)
    verts, cam, tex = _f32c(verts.contiguous()), _f32c(cam.contiguous()), _f32c(tex.contiguous())
    assert faces.dtype == torch.int32 and faces.is_contiguous() and faces.shape[1] == 3
    B, Vsrc = verts.shape[0], verts.shape[1]
    V = tex.shape[0]
    if vmap is not None:
        assert vmap.dtype == torch.int64 and vmap.numel() == V
    else:
        assert V == Vsrc
    H, W = out_size
    dev = verts.device
    scr = torch.empty(B, V, 3, dtype=torch.float32, device=dev)
    zbuf = torch.empty(B, H, W, dtype=torch.int64, device=dev)
    out = torch.empty(B, 3, H, W, dtype=torch.float32, device=dev)
    fo = torch.empty(B, H, W, dtype=torch.int32, device=dev) if want_faces else None
    _check(lib().whmr_iuv_rasterize(verts.data_ptr(), B, Vsrc, _ptr(vmap), V, faces.data_ptr(), faces.shape[0], tex.data_ptr(), cam.data_ptr(),
                                    K[0], K[1], K[2], K[3], focal, orig_size[0], orig_size[1], H, W, scr.data_ptr(), zbuf.data_ptr(), out.data_ptr(),
                                    _ptr(fo), _stream()), 'whmr_iuv_rasterize')
    return (out, fo) if want_faces else out


def _iuv_loss_args(y, iuv):
    """y [B, H, W, 90] logits whose pixels are rows of one [B*H*W, ld] matrix (ConvNHWCFn's padded output view); iuv [B, 3, H, W] fp32, any strides"""
    _dev(y, iuv)
    B, H, W, Cc = y.shape
    ld = y.stride(2)
    assert Cc == 90 and y.stride(3) == 1 and y.stride(1) == W * ld and y.stride(0) == H * W * ld and y.dtype in (torch.float32, torch.bfloat16)
    assert iuv.dtype == torch.float32 and tuple(iuv.shape) == (B, 3, H, W)
    return (y.data_ptr(), int(y.dtype == torch.bfloat16), ld, iuv.data_ptr()) + tuple(iuv.stride()) + (B, H, W)


def iuv_losses(y, iuv, point_weight):
    """(loss_U, loss_V, loss_IndexUV, loss_segAnn) [4] of core/trainer.py:255-298 from the IUV head's channels-last logits (u 25 | v 25 | index 25 |
    ann 15) and the rendered ground-truth IUV image (include/whmr_hip.h: whmr_iuv_losses)."""
    a = _iuv_loss_args(y, iuv)
    P = y.shape[0] * y.shape[1] * y.shape[2]
    partial = torch.empty((P + 127) // 128 * 4, dtype=torch.float32, device=y.device)
    out = torch.empty(4, dtype=torch.float32, device=y.device)
    _check(lib().whmr_iuv_losses(*a, float(point_weight), partial.data_ptr(), out.data_ptr(), _stream()), 'whmr_iuv_losses')
    return out


def iuv_losses_bwd(y, iuv, point_weight, g, ldg):
    """gradient of sum_k g[k] * loss_k with respect to y -> [B*H*W, ldg] in y's dtype, columns >= 90 zero"""
    a = _iuv_loss_args(y, iuv)
    _dev(g)
    g = _f32c(g.contiguous())
    assert g.numel() == 4 and ldg >= 90
    P = y.shape[0] * y.shape[1] * y.shape[2]
    dy = torch.empty(P, ldg, dtype=y.dtype, device=y.device)
    _check(lib().whmr_iuv_losses_bwd(*a, float(point_weight), g.data_ptr(), dy.data_ptr(), ldg, _stream()), 'whmr_iuv_losses_bwd')
    return dy


def layernorm(x, weight, bias, out, eps):
    _dev(x, weight, bias, out)
    assert x.dtype == torch.float32 and x.is_contiguous() and out.is_contiguous()
    Cdim = x.shape[-1]
    _check(lib().whmr_layernorm(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), out.data_ptr(), x.numel() // Cdim,
                                Cdim, eps, int(out.dtype == torch.bfloat16), _stream()), 'whmr_layernorm')
    return out


def patch_im2col(x, out, patch, pad):
    _dev(x, out)
    assert x.dtype == torch.float32 and x.dim() == 4
    B, Cin, H, W = x.shape
    sb, sc, sh, sw = x.stride()
    _check(lib().whmr_patch_im2col(x.data_ptr(), out.data_ptr(), B, Cin, H, W, patch, pad, sb, sc, sh, sw,
                                   int(out.dtype == torch.bfloat16), _stream()), 'whmr_patch_im2col')
    return out


def cast_bf16(src):
    _dev(src)
    src = src.contiguous()
    assert src.dtype == torch.float32
    dst = torch.empty(src.shape, dtype=torch.bfloat16, device=src.device)
    _check(lib().whmr_cast_f32_bf16(src.data_ptr(), dst.data_ptr(), src.numel(), _stream()), 'whmr_cast_f32_bf16')
    return dst


def attention(qkv, out, B, N, H, d, scale):
    _dev(qkv, out)
    assert qkv.is_contiguous() and out.is_contiguous() and qkv.dtype == out.dtype
    _check(lib().whmr_attention(qkv.data_ptr(), out.data_ptr(), B, N, H, d, scale,
                                int(qkv.dtype == torch.bfloat16), _stream()), 'whmr_attention')
    return out


def _ptr(t):
    return t.data_ptr() if t is not None else None


def _f32c(t):
    assert t.dtype == torch.float32 and t.is_contiguous(), 'expected a contiguous fp32 tensor'
    return t


ROT6D, GRAM_SCHMIDT, RODRIGUES = 0, 1, 2


def rot_to_mat(x, mode):
    """x [n, 6|9|3] -> [n, 3, 3]  (rot6d_to_rotmat / unbiased_gram_schmidt / batch_rodrigues)."""
    _dev(x)
    x = _f32c(x.contiguous())
    n = x.numel() // (6, 9, 3)[mode]
    out = torch.empty(n, 3, 3, dtype=torch.float32, device=x.device)
    _check(lib().whmr_rot_to_mat(x.data_ptr(), out.data_ptr(), n, mode, _stream()), 'whmr_rot_to_mat')
    return out


def mat_to_aa_bwd(R, d_aa):
    """d_R [n, 9] = (d aa / d R)^T d_aa [n, 3]: backward of ``mat_to_aa`` (include/whmr_hip.h)"""
    _dev(R, d_aa)
    R, d_aa = R.reshape(-1, 9).float().contiguous(), d_aa.reshape(-1, 3).float().contiguous()
    n = R.shape[0]
    assert d_aa.shape[0] == n
    out = torch.empty(n, 9, dtype=torch.float32, device=R.device)
    _check(lib().whmr_mat_to_aa_bwd(R.data_ptr(), d_aa.data_ptr(), out.data_ptr(), n, _stream()), 'whmr_mat_to_aa_bwd')
    return out


def mat_to_aa(R):
    _dev(R)
    R = _f32c(R.contiguous())
    n = R.numel() // 9
    out = torch.empty(n, 3, dtype=torch.float32, device=R.device)
    _check(lib().whmr_mat_to_aa(R.data_ptr(), out.data_ptr(), n, _stream()), 'whmr_mat_to_aa')
    return out


def perspective(points, rotation, translation, focal, center, post_div=None, post_shift=0.0):
    """perspective_projection (+ optional out / post_div[b] + post_shift).  focal: tensor [B] or python float."""
    _dev(points, rotation, translation, center, post_div)
    points = _f32c(points.contiguous())
    B, P = points.shape[0], points.shape[1]
    if not torch.is_tensor(focal):
        focal = torch.full((1,), float(focal), dtype=torch.float32, device=points.device)
    focal = _f32c(focal.contiguous())
    assert focal.numel() in (1, B)
    fstride = 1 if focal.numel() == B else 0
    rstride = 0
    if rotation is not None:
        rotation = _f32c(rotation.contiguous())
        assert rotation.shape[0] in (1, B)
        rstride = 9 if rotation.shape[0] == B else 0
    out = torch.empty(B, P, 2, dtype=torch.float32, device=points.device)
    _check(lib().whmr_perspective(points.data_ptr(), _ptr(rotation), rstride, _f32c(translation.contiguous()).data_ptr(),
                                  focal.data_ptr(), fstride, _ptr(center.contiguous() if center is not None else None),
                                  _ptr(post_div.contiguous() if post_div is not None else None), post_shift,
                                  out.data_ptr(), B, P, _stream()), 'whmr_perspective')
    return out


def weak_projection(points, cam, focal=1000.0, res_w=256.0, res_h=256.0):
    _dev(points, cam)
    points, cam = _f32c(points.contiguous()), _f32c(cam.contiguous())
    B, P = points.shape[0], points.shape[1]
    out = torch.empty(B, P, 2, dtype=torch.float32, device=points.device)
    _check(lib().whmr_weak_projection(points.data_ptr(), cam.data_ptr(), out.data_ptr(), B, P, focal, res_w, res_h,
                                      _stream()), 'whmr_weak_projection')
    return out


def _rows(t, width):
    """pointer + row stride of a [B, width] fp32 matrix whose rows are contiguous (it may be a column slice)"""
    assert t.dtype == torch.float32 and t.dim() == 2 and t.shape[1] == width and (t.stride(1) == 1 or width == 1)
    return t.data_ptr(), t.stride(0)


def smpl_pose_chain(model, pose9, betas, do_gs, rotmat, aa, A, posed_joints, pose_feat):
    """pose9 [B,216] and betas [B,10]: rows contiguous, any row stride (slices of the regressor state buffer are fine)"""
    B = betas.shape[0]
    pp, ps = _rows(pose9, 216)
    bp, bs = _rows(betas, 10)
    _check(lib().whmr_smpl_pose_chain(C.byref(model), pp, ps, bp, bs, B, int(do_gs), _ptr(rotmat), _ptr(aa), A.data_ptr(),
                                      _ptr(posed_joints), _ptr(pose_feat), _stream()), 'whmr_smpl_pose_chain')


def smpl_blend_skin(model, posedirs_tiled, betas, pose_feat, A, verts):
    """pose-corrective offsets (fp32 MFMA from the re-tiled posedirs) + shape blend + skinning in one launch"""
    bp, bs = _rows(betas, 10)
    assert posedirs_tiled.dtype == torch.float32 and posedirs_tiled.is_contiguous() and tuple(posedirs_tiled.shape) == (108, 208, 192)
    _check(lib().whmr_smpl_blend_skin(C.byref(model), posedirs_tiled.data_ptr(), bp, bs, pose_feat.data_ptr(), A.data_ptr(), betas.shape[0],
                                      verts.data_ptr(), _stream()), 'whmr_smpl_blend_skin')


def smpl_blend_skin_x3(model, posedirs_x3, betas, pose_feat, A, verts):
    """the same launch with the pose-corrective offsets on split-bf16 operands (include/whmr_hip.h: whmr_smpl_blend_skin_x3)"""
    bp, bs = _rows(betas, 10)
    assert posedirs_x3.dtype == torch.bfloat16 and posedirs_x3.is_contiguous() and tuple(posedirs_x3.shape) == (108, 13, 2, 2, 192, 8)
    _check(lib().whmr_smpl_blend_skin_x3(C.byref(model), posedirs_x3.data_ptr(), bp, bs, pose_feat.data_ptr(), A.data_ptr(), betas.shape[0],
                                         verts.data_ptr(), _stream()), 'whmr_smpl_blend_skin_x3')


def smpl_skin(model, betas, pose_feat, A, verts, pose_off=None):
    bp, bs = _rows(betas, 10)
    _check(lib().whmr_smpl_skin(C.byref(model), bp, bs, pose_feat.data_ptr(), A.data_ptr(), _ptr(pose_off),
                                betas.shape[0], verts.data_ptr(), _stream()), 'whmr_smpl_skin')


def regressor_post(state, aa, joints49, Tz, bbox_h, center, orig_shape, focal0=1000.0, res_w=256.0, res_h=256.0):
    """state [B,229] = [pose | shape | cam] rows (any row stride) -> theta, kp_2d, kp_2d_w, cam_t, focal (one launch)"""
    _dev(state, aa, joints49, Tz, bbox_h, center, orig_shape)
    B = state.shape[0]
    sp, ss = _rows(state, 229)
    f32 = dict(dtype=torch.float32, device=state.device)
    theta, kp, kpw = torch.empty(B, 85, **f32), torch.empty(B, 49, 2, **f32), torch.empty(B, 49, 2, **f32)
    cam_t, focal = torch.empty(B, 3, **f32), torch.empty(B, **f32)
    _check(lib().whmr_regressor_post(sp, ss, _f32c(aa).data_ptr(), _f32c(joints49).data_ptr(), _f32c(Tz).data_ptr(),
                                     _f32c(bbox_h).data_ptr(), _f32c(center).data_ptr(), _f32c(orig_shape).data_ptr(), B,
                                     focal0, res_w, res_h, theta.data_ptr(), kp.data_ptr(), kpw.data_ptr(), cam_t.data_ptr(),
                                     focal.data_ptr(), _stream()), 'whmr_regressor_post')
    return theta, kp, kpw, cam_t, focal


def _stage_tail_desc(t, verts, posed_joints, joints49, smpl_joints45, markers, post, nxt):
    """fill a WhmrStageTail; returns the projection outputs (or None)"""
    _dev(verts, posed_joints, joints49, smpl_joints45, markers)
    B = verts.shape[0]
    t.verts, t.posed_joints, t.joints49 = verts.data_ptr(), posed_joints.data_ptr(), _ptr(joints49)
    t.smpl_joints45, t.markers, t.R = _ptr(smpl_joints45), _ptr(markers), 33 if smpl_joints45 is not None else 9
    out = None
    if post is not None:
        st = post['state']
        _dev(st, post['aa'], post['Tz'], post['bbox_h'], post['center'], post['orig_shape'])
        sp, ss = _rows(st, 229)
        f32 = dict(dtype=torch.float32, device=verts.device)
        out = (torch.empty(B, 85, **f32), torch.empty(B, 49, 2, **f32), torch.empty(B, 49, 2, **f32), torch.empty(B, 3, **f32), torch.empty(B, **f32))
        t.state, t.state_stride, t.aa, t.Tz = sp, ss, _f32c(post['aa']).data_ptr(), _f32c(post['Tz']).data_ptr()
        t.bbox_h, t.center, t.orig_shape = _f32c(post['bbox_h']).data_ptr(), _f32c(post['center']).data_ptr(), _f32c(post['orig_shape']).data_ptr()
        t.focal0, t.res_w, t.res_h = post['focal0'], post['res_w'], post['res_h']
        t.theta, t.kp2d, t.kp2d_w, t.cam_t, t.focal = (o.data_ptr() for o in out)
        if nxt is not None:
            _dev(nxt['bbox_info'], nxt['rotmat'], nxt['xc'])
            assert nxt['xc'].dtype == torch.float32 and nxt['xc'].stride(1) == 1 and nxt['xc'].shape[1] >= nxt['F'] + 234
            t.bbox_info, t.rotmat = _f32c(nxt['bbox_info']).data_ptr(), _f32c(nxt['rotmat']).data_ptr()
            t.xc_next, t.ld_next, t.F_next = nxt['xc'].data_ptr(), nxt['xc'].stride(0), nxt['F']
    return out


def smpl_stage_tail(model, verts, posed_joints, joints49, smpl_joints45, markers, post=None, nxt=None, csr=None):
    """One launch for the tail of a regressor stage (whmr_smpl_stage_tail).  post = dict(state, aa, Tz, bbox_h, center, orig_shape, focal0, res_w, res_h)
    -> returns (theta, kp_2d, kp_2d_w, cam_t, focal); nxt = dict(bbox_info, rotmat, xc, F): the next stage's input buffer gets its state columns.
    csr = (ptr, col, val) of the regressor rows: the joint regression runs as a CSR gather inside the same launch (whmr_smpl_stage_tail_csr)."""
    B = verts.shape[0]
    t = WhmrStageTail()
    out = _stage_tail_desc(t, verts, posed_joints, joints49, smpl_joints45, markers, post, nxt)
    if csr is not None:
        _check(lib().whmr_smpl_stage_tail_csr(C.byref(model), C.byref(t), csr[0].data_ptr(), csr[1].data_ptr(), csr[2].data_ptr(), B, _stream()),
               'whmr_smpl_stage_tail_csr')
        return out
    scratch = torch.empty(B * 33 * 3, dtype=torch.float32, device=verts.device)
    _check(lib().whmr_smpl_stage_tail(C.byref(model), C.byref(t), B, scratch.data_ptr(), _stream()), 'whmr_smpl_stage_tail')
    return out


def smpl_joints(model, verts, posed_joints, joints49, smpl_joints45, markers):
    scratch = torch.empty(verts.shape[0] * 33 * 3, dtype=torch.float32, device=verts.device)
    _check(lib().whmr_smpl_joints(C.byref(model), verts.data_ptr(), _ptr(posed_joints), verts.shape[0], _ptr(joints49),
                                  _ptr(smpl_joints45), _ptr(markers), scratch.data_ptr(), _stream()), 'whmr_smpl_joints')


def maf_sample(fmap_nchw, weights, out, pts2d=None, pts3d=None, cam=None, point_feat=None, focal=1000.0, res_w=256.0,
               res_h=256.0):
    """fmap_nchw: logical [B, 256, H, W] tensor of any strides (channels-last memory makes the gather coalesced);
    with no points given, a [B, 256, P] tensor of already-sampled features (MLP only).  out: [B, >= 32*P] rows."""
    _dev(fmap_nchw, out, pts2d, pts3d, cam, point_feat)
    assert fmap_nchw.shape[1] == 256 and fmap_nchw.dtype in (torch.float32, torch.bfloat16)
    if pts2d is None and pts3d is None:
        B, _, P = fmap_nchw.shape
        (sb, sc, sx), sy, H, W = fmap_nchw.stride(), 0, 1, P
    else:
        B, _, H, W = fmap_nchw.shape
        sb, sc, sy, sx = fmap_nchw.stride()
        P = (pts2d if pts2d is not None else pts3d).shape[1]
    assert out.dtype == torch.float32 and out.stride(-1) == 1
    assert cam is None or (cam.dtype == torch.float32 and cam.stride(-1) == 1 and cam.shape[-1] == 3)
    _check(lib().whmr_maf_sample(fmap_nchw.data_ptr(), int(fmap_nchw.dtype == torch.bfloat16), sb, sc, sy, sx, H, W,
                                 _ptr(pts2d), _ptr(pts3d), _ptr(cam), cam.stride(0) if cam is not None else 3, focal, res_w, res_h,
                                 C.byref(weights), B, P,
                                 out.data_ptr(), out.stride(0), _ptr(point_feat), _stream()), 'whmr_maf_sample')
    return out


def tz_tail(tok, w0, b0, w1, b1, bn4, eps, out):
    B, T, D = tok.shape
    _check(lib().whmr_tz_tail(_f32c(tok).data_ptr(), B, T, D, w0.data_ptr(), b0.data_ptr(), w0.shape[0], w1.data_ptr(),
                              b1.data_ptr(), bn4.data_ptr(), eps, out.data_ptr(), _stream()), 'whmr_tz_tail')
    return out


def conv_im2col(x, KH, KW, S, pad, Kpad):
    """NCHW fp32 -> cols [B*OH*OW, Kpad] bf16 (stem conv); returns (cols, OH, OW)"""
    _dev(x)
    B, Cin, H, W = x.shape
    OH, OW = (H + 2 * pad - KH) // S + 1, (W + 2 * pad - KW) // S + 1
    cols = torch.empty(B * OH * OW, Kpad, dtype=torch.bfloat16, device=x.device)
    sb, sc, sh, sw = x.stride()
    _check(lib().whmr_conv_im2col(x.data_ptr(), cols.data_ptr(), B, Cin, H, W, KH, KW, S, pad, Kpad, sb, sc, sh, sw, _stream()),
           'whmr_conv_im2col')
    return cols, OH, OW


def maxpool_nhwc(x, k, s, pad):
    _dev(x)
    assert x.dtype in (torch.bfloat16, torch.float32) and x.is_contiguous()
    B, H, W, Cc = x.shape
    OH, OW = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    y = torch.empty(B, OH, OW, Cc, dtype=x.dtype, device=x.device)
    _check(lib().whmr_maxpool_nhwc(x.data_ptr(), y.data_ptr(), B, H, W, Cc, k, s, pad, int(x.dtype == torch.bfloat16), _stream()),
           'whmr_maxpool_nhwc')
    return y


def avgpool_nhwc(x):
    _dev(x)
    assert x.dtype in (torch.bfloat16, torch.float32) and x.is_contiguous()
    B, H, W, Cc = x.shape
    y = torch.empty(B, Cc, dtype=torch.float32, device=x.device)
    _check(lib().whmr_avgpool_nhwc(x.data_ptr(), y.data_ptr(), B, H * W, Cc, int(x.dtype == torch.bfloat16), _stream()),
           'whmr_avgpool_nhwc')
    return y


def _profile_begin():
    if PROFILE is None:
        return None
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    return e0


def _profile_end(e0, name, work):
    if e0 is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        PROFILE.append((name, work, e0, e1))


def gemm_tn_ok(a, b):
    """shape / alignment envelope of whmr_gemm_tn_bf16 for a [K, Mo], b [K, No] (row strides free)"""
    return (a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and a.dim() == 2 and b.dim() == 2 and a.shape[0] == b.shape[0]
            and a.stride(1) == 1 and b.stride(1) == 1 and a.shape[0] % 32 == 0 and a.shape[1] % 64 == 0 and b.shape[1] % 256 == 0
            and a.stride(0) % 8 == 0 and b.stride(0) % 8 == 0 and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0)


def gemm_tn(a, b, out, splits=0, db=None):
    """out [Mo, No] fp32 = a^T . b for reduction-major bf16 operands a [K, Mo], b [K, No] (weight gradients: dW = dY^T . X, no transposed copies)"""
    _dev(a, b, out)
    assert gemm_tn_ok(a, b) and out.dtype == torch.float32 and out.shape == (a.shape[1], b.shape[1]) and out.stride(1) == 1
    ws = splitk_workspace(a.device)
    assert db is None or (db.dtype == torch.float32 and db.is_contiguous() and db.numel() == a.shape[1])
    ev = _profile_begin()
    _check(lib().whmr_gemm_tn_bf16(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0), _ptr(db), a.shape[1], b.shape[1],
                                   a.shape[0], int(splits), ws.data_ptr(), ws.numel(), _stream()), 'whmr_gemm_tn_bf16')
    _profile_end(ev, 'gemm_bf16', 2.0 * a.shape[0] * a.shape[1] * b.shape[1])
    return out


def gemm_tn_group_ok(jobs):
    """envelope of whmr_gemm_tn_bf16_group for jobs = [(a [K, Mo], b [K, No], out, db | None), ...]"""
    # mirrors EVERY check of the C entry (gemm_tn.hip), the output side included -- the caller falls back to single launches on False instead of
    # learning it from a rejected launch (the two envelopes could disagree on out.stride(0) % 4 / the 16-byte alignment of C)
    return (0 < len(jobs) <= 4 and all(
        gemm_tn_ok(a, b) and a.shape[1] % 256 == 0 and a.shape[0] == jobs[0][0].shape[0]
        and out.dtype == torch.float32 and out.dim() == 2 and tuple(out.shape) == (a.shape[1], b.shape[1]) and out.stride(1) == 1
        and out.stride(0) % 4 == 0 and out.stride(0) >= b.shape[1] and out.data_ptr() % 16 == 0
        and (db is None or (db.dtype == torch.float32 and db.is_contiguous() and db.numel() == a.shape[1]))
        for a, b, out, db in jobs))


def gemm_tn_group(jobs):
    """out_i [Mo_i, No_i] fp32 = a_i^T . b_i (+ db_i = column sums of a_i) for up to 4 products over the same K in ONE launch (the weight gradients
    of a transformer layer: two K slices instead of 7-28 per product).  -> False when the C entry rejects the envelope (include/whmr_hip.h: "the caller
    then issues the single launches"); ``gemm_tn_group_ok`` mirrors its checks, so that is not expected to happen."""
    assert gemm_tn_group_ok(jobs)
    items = (WhmrTnItem * len(jobs))()
    flops = 0.0
    for it, (a, b, out, db) in zip(items, jobs):
        _dev(a, b, out)
        assert out.dtype == torch.float32 and out.shape == (a.shape[1], b.shape[1]) and out.stride(1) == 1
        assert db is None or (db.dtype == torch.float32 and db.is_contiguous() and db.numel() == a.shape[1])
        it.A, it.lda, it.B, it.ldb, it.C, it.ldc = a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0)
        it.db, it.Mo, it.No = _ptr(db), a.shape[1], b.shape[1]
        flops += 2.0 * a.shape[0] * a.shape[1] * b.shape[1]
    ws = splitk_workspace(jobs[0][0].device)
    ev = _profile_begin()
    rc = lib().whmr_gemm_tn_bf16_group(items, len(jobs), jobs[0][0].shape[0], ws.data_ptr(), ws.numel(), _stream())
    if rc == 1:                                  # hipErrorInvalidValue = outside the C envelope: nothing was launched, the caller issues the single launches
        return False
    _check(rc, 'whmr_gemm_tn_bf16_group')
    _profile_end(ev, 'gemm_bf16', flops)
    return True


def conv_dw_tn_ok(a, img):
    """envelope of whmr_conv_dw_tn_bf16 for a [K, Mo] (2-D, rows dense) and an NHWC image [B, IH, IW, C]"""
    return (a.dtype == torch.bfloat16 and img.dtype == torch.bfloat16 and a.dim() == 2 and img.dim() == 4 and a.stride(1) == 1 and img.is_contiguous()
            and a.shape[0] % 32 == 0 and a.shape[1] % 128 == 0 and img.shape[3] % 256 == 0 and a.stride(0) % 8 == 0 and a.data_ptr() % 16 == 0
            and img.data_ptr() % 16 == 0)


def conv_dw_tn(a, img, out, OH, OW, KH, KW, S, P, splits=0, db=None):
    """out [Mo, KH*KW*C] fp32 = a^T . col(img): convolution weight gradient without a column matrix (see include/whmr_hip.h).  a [B*OH*OW, Mo] bf16,
    img [B, IH, IW, C] bf16 NHWC.  S: the stride, or (row stride, column stride)."""
    _dev(a, img, out)
    Bn, IH, IW, Cc = img.shape
    assert conv_dw_tn_ok(a, img) and a.shape[0] == Bn * OH * OW and out.dtype == torch.float32 and out.shape == (a.shape[1], KH * KW * Cc) and out.stride(1) == 1
    ws = splitk_workspace(a.device)
    ev = _profile_begin()
    if isinstance(S, (tuple, list)):
        _check(lib().whmr_conv_dw_tn2_bf16(a.data_ptr(), a.stride(0), img.data_ptr(), img.stride(2), out.data_ptr(), out.stride(0), a.shape[1], a.shape[0], Bn,
                                           OH, OW, IH, IW, Cc, KH, KW, int(S[0]), int(S[1]), P, zero_page(a.device).data_ptr(), int(splits), ws.data_ptr(),
                                           ws.numel(), _ptr(db), _stream()), 'whmr_conv_dw_tn2_bf16')
    else:
        _check(lib().whmr_conv_dw_tn_bf16(a.data_ptr(), a.stride(0), img.data_ptr(), img.stride(2), out.data_ptr(), out.stride(0), a.shape[1], a.shape[0], Bn,
                                          OH, OW, IH, IW, Cc, KH, KW, S, P, zero_page(a.device).data_ptr(), int(splits), ws.data_ptr(), ws.numel(), _ptr(db), _stream()),
               'whmr_conv_dw_tn_bf16')
    _profile_end(ev, 'gemm_bf16', 2.0 * a.shape[0] * a.shape[1] * KH * KW * Cc)
    return out


def tz_compose(T, Ci, g):
    """T [245, 49 Ci] fp32 -> g [128, 36 Ci] (bf16 / fp32): the composed Tz convolution's space-to-depth weight matrix (whmr_tz_compose)"""
    _dev(T, g)
    assert T.dtype == torch.float32 and T.is_contiguous() and tuple(T.shape) == (245, 49 * Ci) and g.is_contiguous() and tuple(g.shape) == (128, 36 * Ci)
    assert g.dtype in (torch.bfloat16, torch.float32)
    _check(lib().whmr_tz_compose(T.data_ptr(), Ci, g.data_ptr(), _bf(g), _stream()), 'whmr_tz_compose')
    return g


def tz_compose_bwd(dG, Ci, dT):
    _dev(dG, dT)
    assert dG.dtype == torch.float32 and dG.is_contiguous() and tuple(dG.shape) == (128, 36 * Ci)
    assert dT.dtype == torch.float32 and dT.is_contiguous() and tuple(dT.shape) == (245, 49 * Ci)
    _check(lib().whmr_tz_compose_bwd(dG.data_ptr(), Ci, dT.data_ptr(), _stream()), 'whmr_tz_compose_bwd')
    return dT


def tz_unfold(dtok, dP, B, OHp, OWp, OH, OW):
    """dtok [B * 5, OH * OW] fp32 -> dP [B * OHp * OWp, 128] bf16 (transpose of tz_fold)"""
    _dev(dtok, dP)
    assert dtok.dtype == torch.float32 and dtok.is_contiguous() and dtok.numel() == B * 5 * OH * OW
    assert dP.dtype == torch.bfloat16 and dP.is_contiguous() and tuple(dP.shape) == (B * OHp * OWp, 128)
    _check(lib().whmr_tz_unfold(dtok.data_ptr(), dP.data_ptr(), B, OHp, OWp, OH, OW, _stream()), 'whmr_tz_unfold')
    return dP


def gemm_f32_set_big(on):
    """A/B: 128x128 large-M fp32 GEMM kernel on (default) / off"""
    _check(lib().whmr_gemm_f32_set_big(int(on)), 'whmr_gemm_f32_set_big')


def set_option(key, value):
    """Tuning switches for A/B measurements (whmr_set_option)."""
    _check(lib().whmr_set_option(int(key), int(value)), 'whmr_set_option')


def regressor_state(xc, F, bbox_info, pose, shape, cam):
    """xc[:, F:F+234] = [bbox_info | pose(216) | shape(10) | cam(3)]; pose/shape/cam [B,.] rows, any row stride (0 = broadcast)."""
    _dev(xc, bbox_info, pose, shape, cam)
    B = xc.shape[0]
    for t, n in ((pose, 216), (shape, 10), (cam, 3)):
        assert t.dtype == torch.float32 and t.shape == (B, n) and (n == 1 or t.stride(1) == 1)
    assert bbox_info.dtype == torch.float32 and bbox_info.is_contiguous() and bbox_info.shape == (B, 5)
    _check(lib().whmr_regressor_state(bbox_info.data_ptr(), pose.data_ptr(), pose.stride(0), shape.data_ptr(), shape.stride(0),
                                      cam.data_ptr(), cam.stride(0), B, xc.data_ptr(), xc.stride(0), F, _stream()), 'whmr_regressor_state')


def crop_normalize(frame, inv_affine, patch_w, patch_h, x0, x1, out, raw, mean, std):
    _dev(frame, inv_affine, out, raw)
    assert frame.dtype == torch.uint8 and inv_affine.dtype == torch.float64 and inv_affine.is_contiguous() and out.is_contiguous()
    H, W = frame.shape[:2]
    m3, s3 = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    _check(lib().whmr_crop_normalize(frame.data_ptr(), H, W, frame.stride(0), inv_affine.data_ptr(), inv_affine.shape[0], patch_w, patch_h,
                                     x0, x1, out.data_ptr(), _ptr(raw), m3, s3, _stream()), 'whmr_crop_normalize')
    return out


def attention_x3_set_variant(old_kernel):
    """1 = the round-3 split-bf16 attention kernel for every N (A/B, tests); 0 = the persistent 16-row-tile kernel where it applies (N <= 208)"""
    _check(lib().whmr_attention_x3_set_variant(int(old_kernel)), 'whmr_attention_x3_set_variant')


def attention_set_variant(chunked):
    """bit 0: 1 = chunked online-softmax bf16 attention kernel (default), 0 = single-pass kernel; bits 1-2: timing ablations of the blocked kernels
    (wrong results); bit 3 set: fp32 attention on the VALU kernel instead of the matrix-pipe one; bit 4 set: the blocked bf16 attention on the
    round-2 kernel instead of the persistent 16-row-tile one (A/B measurements, tests)."""
    _check(lib().whmr_attention_set_variant(int(chunked)), 'whmr_attention_set_variant')


# ----------------------------------------------------------------------------- backward-pass helpers (train_ops.hip)
_train_scratch = {}
_train_scratch_retired = []        # outgrown buffers: kept alive for the graphs that recorded their pointers


def train_scratch(device, nfloats):
    """fp32 scratch of the two-stage reductions (column sums, LayerNorm backward): one buffer per (device, stream) -- two streams may reduce
    concurrently -- that only ever GROWS by allocating a new buffer while the old ones stay alive: a captured HIP graph (capture_train_step,
    GraphedForward) keeps the raw pointer it was recorded with, so a buffer is never handed back to the allocator."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    t = _train_scratch.get(key)
    if t is None or t.numel() < nfloats:
        if t is not None:
            _train_scratch_retired.append(t)
        t = _train_scratch[key] = torch.empty(max(nfloats, 1 << 20), dtype=torch.float32, device=device)
    return t


def transpose_cast(src, dtype, pad_to=64):
    """src [R,C] (fp32/bf16, row stride free) -> [C, Rpad] in ``dtype`` with Rpad = R rounded up to ``pad_to`` (zero filled)."""
    _dev(src)
    assert src.dim() == 2 and src.stride(1) == 1 and src.dtype in (torch.float32, torch.bfloat16) and dtype in (torch.float32, torch.bfloat16)
    R, Cc = src.shape
    Rpad = (R + pad_to - 1) // pad_to * pad_to
    dst = torch.empty(Cc, Rpad, dtype=dtype, device=src.device)
    _check(lib().whmr_transpose_cast(src.data_ptr(), int(src.dtype == torch.bfloat16), src.stride(0), dst.data_ptr(),
                                     int(dtype == torch.bfloat16), Rpad, R, Cc, Rpad, _stream()), 'whmr_transpose_cast')
    return dst


def transpose_colsum(src, out, pad_to=64, accumulate=False):
    """bf16 src [R,C] -> (src^T [C,Rpad] bf16, out[c] (+)= sum_r src[r,c]) in one pass over src; falls back to the two separate kernels when
    the shape misses the fast path's 8-element alignment."""
    _dev(src, out)
    assert src.dim() == 2 and src.stride(1) == 1 and out.dtype == torch.float32 and out.is_contiguous() and out.numel() == src.shape[1]
    R, Cc = src.shape
    Rpad = (R + pad_to - 1) // pad_to * pad_to
    if src.dtype != torch.bfloat16 or ((R | Cc | Rpad | src.stride(0)) & 7) or (src.data_ptr() & 15):
        colsum(src, out, accumulate)
        return transpose_cast(src, src.dtype, pad_to)
    dst = torch.empty(Cc, Rpad, dtype=torch.bfloat16, device=src.device)
    sc = train_scratch(src.device, ((R + 63) // 64) * Cc)
    _check(lib().whmr_transpose_colsum(src.data_ptr(), src.stride(0), dst.data_ptr(), Rpad, R, Cc, Rpad, out.data_ptr(), int(accumulate),
                                       sc.data_ptr(), _stream()), 'whmr_transpose_colsum')
    return dst


def colsum(x, out, accumulate=False):
    _dev(x, out)
    assert x.dim() == 2 and x.stride(1) == 1 and out.dtype == torch.float32 and out.is_contiguous() and out.numel() == x.shape[1]
    R, Cc = x.shape
    sc = train_scratch(x.device, max(64 * Cc, 1 << 20))
    _check(lib().whmr_colsum(x.data_ptr(), int(x.dtype == torch.bfloat16), x.stride(0), R, Cc, out.data_ptr(), int(accumulate),
                             sc.data_ptr(), _stream()), 'whmr_colsum')
    return out


class WeightOperands:
    """bf16 operand copies (W [N, K] and W^T [K, N]) of a fixed list of fp32 weight matrices, all re-made by ONE launch (whmr_weights_prepare)
    whenever a parameter's version moved -- the buffers (and so the pointers a captured graph holds) never change."""

    def __init__(self, params, shapes=None):
        import struct
        self.params = list(params)
        dev = self.params[0].device
        shapes = shapes or {}
        self.copies, rows, tb = {}, [], 0
        for p in self.params:
            N, K = shapes.get(id(p), (p.shape[0], p.numel() // p.shape[0]))
            assert p.dtype == torch.float32 and p.is_contiguous()
            w = torch.empty(N, K, dtype=torch.bfloat16, device=dev)
            wt = torch.empty(K, N, dtype=torch.bfloat16, device=dev)
            self.copies[id(p)] = (w, wt)
            tk = (K + 63) // 64
            rows.append(struct.pack('<QQQiiii', p.data_ptr(), w.data_ptr(), wt.data_ptr(), N, K, tb, tk))
            tb += ((N + 63) // 64) * tk
        self.total = tb
        self.table = torch.frombuffer(bytearray(b''.join(rows)), dtype=torch.uint8).to(dev)
        self.ptrs = [p.data_ptr() for p in self.params]
        self.versions = None

    def stale(self):
        return self.ptrs != [p.data_ptr() for p in self.params]       # parameter storage replaced (e.g. .to(), load with assign): rebuild

    def refresh(self):
        v = [p._version for p in self.params]
        if v != self.versions:
            _check(lib().whmr_weights_prepare(self.table.data_ptr(), len(self.params), self.total, _stream()), 'whmr_weights_prepare')
            self.versions = v
        return self.copies


def layernorm_bwd(x, dy, gamma, dres, dx, dgamma, dbeta, eps, accumulate=False, cast_out=None, row_scale=None):
    """dx = LN'(x)(dy) (+ dres); dgamma / dbeta written or accumulated.  All fp32, rows contiguous.  cast_out (bf16, same shape) also receives
    dx * row_scale[row] (row_scale [rows] fp32 or None)."""
    _dev(x, dy, gamma, dres, dx, dgamma, dbeta, cast_out, row_scale)
    assert cast_out is None or (cast_out.dtype == torch.bfloat16 and cast_out.is_contiguous() and cast_out.numel() == x.numel())
    assert row_scale is None or (cast_out is not None and row_scale.dtype == torch.float32 and row_scale.is_contiguous()
                                 and row_scale.numel() == x.numel() // x.shape[-1])
    for t in (x, dy, dx, dres):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous())
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    sc = train_scratch(x.device, 2048 * Cc)
    _check(lib().whmr_layernorm_bwd(x.data_ptr(), dy.data_ptr(), gamma.data_ptr(), _ptr(dres), dx.data_ptr(), dgamma.data_ptr(),
                                    dbeta.data_ptr(), int(accumulate), rows, Cc, eps, sc.data_ptr(), _ptr(cast_out), _ptr(row_scale), _stream()), 'whmr_layernorm_bwd')
    return dx


def gelu_fwd(pre, out):
    _dev(pre, out)
    assert pre.dtype == out.dtype and pre.is_contiguous() and out.is_contiguous()
    _check(lib().whmr_gelu_fwd(pre.data_ptr(), out.data_ptr(), int(pre.dtype == torch.bfloat16), pre.numel(), _stream()), 'whmr_gelu_fwd')
    return out


def gelu_bwd(pre, dhid, dpre):
    _dev(pre, dhid, dpre)
    assert dhid.dtype in (torch.float32, torch.bfloat16) and pre.is_contiguous() and dhid.is_contiguous() and dpre.is_contiguous()
    _check(lib().whmr_gelu_bwd(pre.data_ptr(), int(pre.dtype == torch.bfloat16), dhid.data_ptr(), int(dhid.dtype == torch.bfloat16),
                               dpre.data_ptr(), int(dpre.dtype == torch.bfloat16), pre.numel(), _stream()), 'whmr_gelu_bwd')
    return dpre


def attention_fwd_train(qkv, out, lse, B, N, H, d, scale):
    _dev(qkv, out, lse)
    assert qkv.dtype == torch.bfloat16 and out.dtype == torch.bfloat16 and lse.dtype == torch.float32
    assert qkv.is_contiguous() and out.is_contiguous() and lse.is_contiguous() and lse.numel() == B * H * N
    _check(lib().whmr_attention_fwd_train(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, N, H, d, scale, _stream()), 'whmr_attention_fwd_train')
    return out


def attention_bwd(qkv, o, dout, lse, dqkv, B, N, H, d, scale):
    _dev(qkv, o, dout, lse, dqkv)
    assert qkv.dtype == torch.bfloat16 and o.dtype == torch.bfloat16 and dqkv.dtype == torch.bfloat16
    assert dout.dtype == torch.float32 and lse.dtype == torch.float32
    for t in (qkv, o, dout, lse, dqkv):
        assert t.is_contiguous()
    _check(lib().whmr_attention_bwd(qkv.data_ptr(), o.data_ptr(), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), B, N, H, d, scale,
                                    _stream()), 'whmr_attention_bwd')
    return dqkv


def attention_bwd_f32(qkv, dout, B, N, H, d, scale):
    """fp32 attention backward: qkv [B*N, 3*H*d], dout [B*N, H*d] -> dqkv like qkv (P recomputed; deterministic)"""
    _dev(qkv, dout)
    qkv, dout = _f32c(qkv.contiguous()), _f32c(dout.contiguous())
    dqkv = torch.empty_like(qkv)
    scratch = torch.empty(B * H * 2 * N * N, dtype=torch.float32, device=qkv.device)
    _check(lib().whmr_attention_bwd_f32(qkv.data_ptr(), dout.data_ptr(), dqkv.data_ptr(), scratch.data_ptr(), B, N, H, d, scale, _stream()),
           'whmr_attention_bwd_f32')
    return dqkv


def tz_conv1(x_nhwc, w, tok):
    """x [B,IH,IW,64] (bf16 / fp32) -- or fp32 [B,IH,IW,128] whose channel halves are ADDED as they are read (the bf16x3 conv0 leaves the W_lo product
    in columns 64..127) -- , w [5,49,64] fp32 -> tok [B,5,OH*OW] fp32 (whmr.py:420 + the reshape at :571)."""
    _dev(x_nhwc, w, tok)
    B, IH, IW, Cc = x_nhwc.shape
    assert Cc in (64, 128) and x_nhwc.is_contiguous() and w.dtype == torch.float32 and w.is_contiguous() and tok.is_contiguous()
    assert Cc == 64 or x_nhwc.dtype == torch.float32
    mode = 2 if Cc == 128 else int(x_nhwc.dtype == torch.bfloat16)
    _check(lib().whmr_tz_conv1(x_nhwc.data_ptr(), mode, w.data_ptr(), tok.data_ptr(), B, IH, IW, _stream()),
           'whmr_tz_conv1')
    return tok


def tz_fold(P, tok, B, OHp, OWp, OH, OW, halves=1, nsplit=1, split_stride=0):
    """tok[b, o, r*OW + s] = sum_{jA, jB} P[(b, r + jA, s + jB), (jA*5 + jB)*5 + o] (+ column 128 + n when halves == 2): the 25-term tail of the composed
    Tz-head convolution Conv2d(256, 5, k25, s6) == conv1(conv0(.)) (whmr.py:418-421,567-571) evaluated as a space-to-depth implicit GEMM."""
    _dev(P, tok)
    assert P.dtype == torch.float32 and P.is_contiguous() and tok.dtype == torch.float32 and tok.is_contiguous()
    ldp = P.shape[-1]
    assert P.numel() >= (nsplit - 1) * split_stride + B * OHp * OWp * ldp and tok.numel() == B * 5 * OH * OW
    _check(lib().whmr_tz_fold(P.data_ptr(), ldp, halves, nsplit, split_stride, tok.data_ptr(), B, OHp, OWp, OH, OW, _stream()), 'whmr_tz_fold')
    return tok


def estimate_translation(S, joints_2d, j0, nj, focal, img_w, img_h):
    _dev(S, joints_2d)
    S, joints_2d = _f32c(S), _f32c(joints_2d)
    B, J = S.shape[:2]
    out = torch.empty(B, 3, dtype=torch.float32, device=S.device)
    _check(lib().whmr_estimate_translation(S.data_ptr(), joints_2d.data_ptr(), B, J, j0, nj, focal, img_w, img_h, out.data_ptr(), _stream()),
           'whmr_estimate_translation')
    return out


# ---- training mode of the deconv pyramid (deconv_train.hip) ----------------------------------------------------------
def _bf(t):
    return int(t.dtype == torch.bfloat16)


def bn_stats(z, gamma, beta, eps, momentum=0.0, running_mean=None, running_var=None):
    """z [M, C] channels-last -> stats [4, C] = mean | invstd | a | b (fp32); running stats updated in place when given."""
    _dev(z, gamma, beta, running_mean, running_var)
    assert z.dim() == 2 and z.is_contiguous() and z.dtype in (torch.float32, torch.bfloat16)
    M, Cc = z.shape
    stats = torch.empty(4, Cc, dtype=torch.float32, device=z.device)
    sc = train_scratch(z.device, 2050 * Cc)
    _check(lib().whmr_bn_stats(z.data_ptr(), _bf(z), M, Cc, _ptr(gamma), _ptr(beta), eps, momentum, _ptr(running_mean), _ptr(running_var),
                               stats.data_ptr(), sc.data_ptr(), _stream()), 'whmr_bn_stats')
    return stats


def bn_sums(z):
    """SyncBatchNorm forward, first half: z [M, C] channels-last -> sums64 [2C + 1] fp64 = sum z | sum z^2 | M of THIS rank (include/whmr_hip.h);
    the caller all-reduces it over the ranks and hands it to ``bn_stats_from_sums``."""
    _dev(z)
    assert z.dim() == 2 and z.is_contiguous() and z.dtype in (torch.float32, torch.bfloat16)
    M, Cc = z.shape
    sums = torch.empty(2 * Cc + 1, dtype=torch.float64, device=z.device)
    sc = train_scratch(z.device, 2050 * Cc)
    _check(lib().whmr_bn_sums(z.data_ptr(), _bf(z), M, Cc, sums.data_ptr(), sc.data_ptr(), _stream()), 'whmr_bn_sums')
    return sums


def bn_stats_from_sums(sums, gamma, beta, eps, momentum=0.0, running_mean=None, running_var=None):
    """sums64 [2C + 1] (summed over the ranks) -> stats [4, C] = mean | invstd | a | b; running stats updated in place when given."""
    _dev(sums, gamma, beta, running_mean, running_var)
    assert sums.dtype == torch.float64 and sums.is_contiguous() and sums.numel() % 2 == 1
    Cc = (sums.numel() - 1) // 2
    stats = torch.empty(4, Cc, dtype=torch.float32, device=sums.device)
    _check(lib().whmr_bn_stats_from_sums(sums.data_ptr(), Cc, _ptr(gamma), _ptr(beta), eps, momentum, _ptr(running_mean), _ptr(running_var),
                                         stats.data_ptr(), _stream()), 'whmr_bn_stats_from_sums')
    return stats


def bn_bwd_sums(z, dy, stats, dgamma, dbeta, accumulate=False):
    """SyncBatchNorm backward, first half: -> sums64 [2C] fp64 = sum g | sum g xhat of THIS rank's rows; dgamma / dbeta get the LOCAL sums."""
    _dev(z, dy, stats, dgamma, dbeta)
    assert z.is_contiguous() and dy.is_contiguous() and z.shape == dy.shape
    M, Cc = z.shape
    sums = torch.empty(2 * Cc, dtype=torch.float64, device=z.device)
    sc = train_scratch(z.device, 2050 * Cc)
    _check(lib().whmr_bn_bwd_sums(z.data_ptr(), _bf(z), dy.data_ptr(), _bf(dy), stats.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), int(accumulate),
                                  M, Cc, sums.data_ptr(), sc.data_ptr(), _stream()), 'whmr_bn_bwd_sums')
    return sums


def bn_bwd_apply(z, dy, stats, sums, count, dz):
    """second half: dz from the globally summed ``sums`` [2C] and the global row count ``count`` (a 1-element fp64 device tensor)."""
    _dev(z, dy, stats, sums, count, dz)
    assert z.is_contiguous() and dy.is_contiguous() and dz.is_contiguous() and z.shape == dy.shape == dz.shape
    assert sums.dtype == torch.float64 and count.dtype == torch.float64 and count.numel() == 1
    M, Cc = z.shape
    sc = train_scratch(z.device, 2050 * Cc)
    _check(lib().whmr_bn_bwd_apply(z.data_ptr(), _bf(z), dy.data_ptr(), _bf(dy), stats.data_ptr(), sums.data_ptr(), count.data_ptr(), dz.data_ptr(),
                                   _bf(dz), M, Cc, sc.data_ptr(), _stream()), 'whmr_bn_bwd_apply')
    return dz


def bn_apply_relu(z, stats, y):
    _dev(z, stats, y)
    assert z.is_contiguous() and y.is_contiguous() and z.shape == y.shape
    M, Cc = z.shape
    _check(lib().whmr_bn_apply_relu(z.data_ptr(), _bf(z), stats.data_ptr(), y.data_ptr(), _bf(y), M, Cc, _stream()), 'whmr_bn_apply_relu')
    return y


def bn_relu_bwd(z, dy, stats, dz, dgamma, dbeta, accumulate=False):
    _dev(z, dy, stats, dz, dgamma, dbeta)
    assert z.is_contiguous() and dy.is_contiguous() and dz.is_contiguous() and z.shape == dy.shape == dz.shape
    M, Cc = z.shape
    sc = train_scratch(z.device, 2050 * Cc)
    _check(lib().whmr_bn_relu_bwd(z.data_ptr(), _bf(z), dy.data_ptr(), _bf(dy), stats.data_ptr(), dz.data_ptr(), _bf(dz), dgamma.data_ptr(),
                                  dbeta.data_ptr(), int(accumulate), M, Cc, sc.data_ptr(), _stream()), 'whmr_bn_relu_bwd')
    return dz


def im2col_t(src_nhwc, OH, OW, KH, KW, S, P, pad_to=64):
    """NHWC map -> transposed column matrix [(ky, kx, c), Mpad], m = (b, oy, ox)."""
    _dev(src_nhwc)
    assert src_nhwc.dim() == 4 and src_nhwc.is_contiguous() and src_nhwc.dtype in (torch.float32, torch.bfloat16)
    B, IH, IW, Cc = src_nhwc.shape
    M = B * OH * OW
    Mpad = (M + pad_to - 1) // pad_to * pad_to
    dst = torch.empty(KH * KW * Cc, Mpad, dtype=src_nhwc.dtype, device=src_nhwc.device)
    _check(lib().whmr_im2col_t(src_nhwc.data_ptr(), dst.data_ptr(), _bf(src_nhwc), B, IH, IW, Cc, OH, OW, KH, KW, S, P, Mpad, _stream()),
           'whmr_im2col_t')
    return dst


def smpl_joints_bwd(model, d_joints49, d_smpl_joints45, d_markers, d_verts, d_posed_joints, d_regd):
    _dev(d_joints49, d_smpl_joints45, d_markers, d_verts, d_posed_joints, d_regd)
    for t in (d_joints49, d_smpl_joints45, d_markers, d_verts, d_posed_joints, d_regd):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous())
    B = d_verts.shape[0]
    _check(lib().whmr_smpl_joints_bwd(C.byref(model), _ptr(d_joints49), _ptr(d_smpl_joints45), _ptr(d_markers), B, d_verts.data_ptr(),
                                      d_posed_joints.data_ptr(), d_regd.data_ptr(), _stream()), 'whmr_smpl_joints_bwd')


def smpl_skin_bwd(model, betas, A, pose_off, d_verts, d_regd, d_vposed, dA_partial):
    _dev(betas, A, pose_off, d_verts, d_regd, d_vposed, dA_partial)
    B = d_verts.shape[0]
    R = 0 if d_regd is None else d_regd.shape[1]
    _check(lib().whmr_smpl_skin_bwd(C.byref(model), betas.data_ptr(), betas.stride(0), A.data_ptr(), pose_off.data_ptr(), d_verts.data_ptr(),
                                    _ptr(d_regd), R, B, d_vposed.data_ptr(), dA_partial.data_ptr(), _stream()), 'whmr_smpl_skin_bwd')


def smpl_chain_bwd(model, rotmat, betas, dA_partial, d_posed_joints, d_pf_beta, d_rotmat, d_betas):
    _dev(rotmat, betas, dA_partial, d_posed_joints, d_pf_beta, d_rotmat, d_betas)
    B = rotmat.shape[0]
    _check(lib().whmr_smpl_chain_bwd(C.byref(model), rotmat.data_ptr(), betas.data_ptr(), betas.stride(0), dA_partial.data_ptr(),
                                     _ptr(d_posed_joints), d_pf_beta.data_ptr(), B, d_rotmat.data_ptr(), d_betas.data_ptr(), _stream()),
           'whmr_smpl_chain_bwd')


MAF_RECORD = 264                # floats per point of the sampler's compact gradient record (whmr_hip.h: whmr_maf_scatter)


def maf_sample_bwd(fmap_nchw, weights, w0, w1, w2, d_out, d_fmap_nchw, XT, DT, pts2d=None, pts3d=None, cam=None, focal=1000.0, res_w=256.0,
                   res_h=256.0, record=None):
    """fmap / d_fmap: logical [B,256,H,W] tensors of any strides (d_fmap fp32 or bf16, accumulated into; may be None).
    ``record`` ([B*P, MAF_RECORD] fp32, instead of d_fmap): the map gradient is left as per-point records for ``maf_scatter``."""
    _dev(fmap_nchw, d_out, d_fmap_nchw, XT, DT, pts2d, pts3d, cam, w0, w1, w2, record)
    B, _, H, W = fmap_nchw.shape
    sb, sc, sy, sx = fmap_nchw.stride()
    P = (pts2d if pts2d is not None else pts3d).shape[1]
    assert d_out.dtype == torch.float32 and d_out.stride(-1) == 1 and XT.shape[0] == 448 and DT.shape[0] == 224
    assert XT.is_contiguous() and DT.is_contiguous() and XT.shape[1] == DT.shape[1] >= B * P
    g = (0, 0, 0, 0) if d_fmap_nchw is None else d_fmap_nchw.stride()
    assert d_fmap_nchw is None or (d_fmap_nchw.dtype in (torch.float32, torch.bfloat16) and d_fmap_nchw.shape == fmap_nchw.shape)
    if record is not None:
        assert d_fmap_nchw is None and record.dtype == torch.float32 and record.is_contiguous() and tuple(record.shape) == (B * P, MAF_RECORD)
        _check(lib().whmr_maf_sample_bwd(fmap_nchw.data_ptr(), _bf(fmap_nchw), sb, sc, sy, sx, H, W, _ptr(pts2d), _ptr(pts3d), _ptr(cam),
                                         cam.stride(0) if cam is not None else 0, focal, res_w, res_h, C.byref(weights), w0.data_ptr(),
                                         w1.data_ptr(), w2.data_ptr(), B, P, d_out.data_ptr(), d_out.stride(0), record.data_ptr(), 2, 0, 0, 0, 0,
                                         XT.data_ptr(), DT.data_ptr(), XT.shape[1], _stream()), 'whmr_maf_sample_bwd')
        return
    _check(lib().whmr_maf_sample_bwd(fmap_nchw.data_ptr(), _bf(fmap_nchw), sb, sc, sy, sx, H, W, _ptr(pts2d), _ptr(pts3d), _ptr(cam),
                                     cam.stride(0) if cam is not None else 0, focal, res_w, res_h, C.byref(weights), w0.data_ptr(),
                                     w1.data_ptr(), w2.data_ptr(), B, P, d_out.data_ptr(), d_out.stride(0), _ptr(d_fmap_nchw),
                                     0 if d_fmap_nchw is None else _bf(d_fmap_nchw), g[0], g[1],
                                     g[2], g[3], XT.data_ptr(), DT.data_ptr(), XT.shape[1], _stream()), 'whmr_maf_sample_bwd')


def maf_scatter(record, d_fmap_nchw, P):
    """d_fmap (logical [B,256,H,W], fp32 of any strides or bf16 channels-last; written by the map's other consumers) += the sampler's records."""
    _dev(record, d_fmap_nchw)
    B, Cc, H, W = d_fmap_nchw.shape
    assert Cc == 256 and record.dtype == torch.float32 and record.is_contiguous() and tuple(record.shape) == (B * P, MAF_RECORD)
    assert d_fmap_nchw.dtype in (torch.float32, torch.bfloat16)
    g = d_fmap_nchw.stride()
    _check(lib().whmr_maf_scatter(record.data_ptr(), B, P, W, d_fmap_nchw.data_ptr(), _bf(d_fmap_nchw), g[0], g[1], g[2], g[3], _stream()),
           'whmr_maf_scatter')


def col2im(dcol, dx_nhwc, OH, OW, KH, KW, S, P):
    """dcol [B*OH*OW, KH*KW*C] (row stride free) -> dx [B,IH,IW,C] (written, not accumulated)."""
    _dev(dcol, dx_nhwc)
    B, IH, IW, Cc = dx_nhwc.shape
    assert dcol.dim() == 2 and dcol.stride(1) == 1 and dcol.shape == (B * OH * OW, KH * KW * Cc) and dx_nhwc.is_contiguous()
    _check(lib().whmr_col2im(dcol.data_ptr(), _bf(dcol), dcol.stride(0), dx_nhwc.data_ptr(), _bf(dx_nhwc), B, IH, IW, Cc, OH, OW, KH, KW, S, P,
                             _stream()), 'whmr_col2im')
    return dx_nhwc


def dense_to_csr(d):
    """Dense [n_out, n_in] device matrix -> (ptr int32 [n_out+1], col int32, val fp32); one host sync (the non-zero count), done once per
    weight version by the callers' caches."""
    sp = d.detach().float().to_sparse_csr()
    return (sp.crow_indices().to(torch.int32).contiguous(), sp.col_indices().to(torch.int32).contiguous(), sp.values().contiguous())


def csr_apply3(csr, x, n_out, out=None, accumulate=False):
    """x [B, n_in, 3] fp32 -> out [B, n_out, 3] = D . x for the CSR triple of D."""
    _dev(x, out)
    ptr, col, val = csr
    x = _f32c(x)
    B, n_in = x.shape[0], x.shape[1]
    if out is None:
        out = torch.empty(B, n_out, 3, dtype=torch.float32, device=x.device)
    _check(lib().whmr_csr_apply3(ptr.data_ptr(), col.data_ptr(), val.data_ptr(), x.data_ptr(), n_in, out.data_ptr(), n_out, B, int(accumulate),
                                 _stream()), 'whmr_csr_apply3')
    return out


# ---- attainable ceilings of this box (csrc/ceilings.hip): measurement aids of bench.py, not on the data path ---------------------------------
def mfma_ceiling(device, seconds=0.15):
    """-> dict(tflops, sclk_mhz): register-fed bf16 MFMA stream on every SIMD (two waves each, random operands), sized from a short calibration launch
    to run ~``seconds``; the shader clock is read inside the kernel (s_memtime per s_memrealtime tick)."""
    sink = torch.zeros(64, dtype=torch.float32, device=device)
    stats = torch.zeros(2, dtype=torch.int64, device=device)
    blocks = 2 * torch.cuda.get_device_properties(device).multi_processor_count

    def run(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _check(lib().whmr_mfma_ceiling(blocks, iters, sink.data_ptr(), stats.data_ptr(), _stream()), 'whmr_mfma_ceiling')
        e1.record()
        torch.cuda.synchronize(device)
        return e0.elapsed_time(e1) * 1e-3
    t = run(20000)
    iters = int(min(max(20000 * seconds / max(t, 1e-6), 20000), 50e6))
    run(iters)                                  # warm: the package clock settles under the load
    t = run(iters)
    st = stats.tolist()
    return {'tflops': blocks * 4 * iters * 8 * 16384.0 / t / 1e12, 'sclk_mhz': st[1] / max(st[0], 1) * 100.0, 'seconds': t}


def hbm_copy_ceiling(device, mbytes=1024, reps=5):
    """-> GB/s moved (read + write) by a streaming copy of ``mbytes`` MiB"""
    n = mbytes * (1 << 20)
    src = torch.empty(n, dtype=torch.uint8, device=device).random_(0, 255)
    dst = torch.empty_like(src)
    _check(lib().whmr_hbm_copy(src.data_ptr(), dst.data_ptr(), n, _stream()), 'whmr_hbm_copy')
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _check(lib().whmr_hbm_copy(src.data_ptr(), dst.data_ptr(), n, _stream()), 'whmr_hbm_copy')
    e1.record()
    torch.cuda.synchronize(device)
    assert torch.equal(src[:4096], dst[:4096]) and torch.equal(src[-4096:], dst[-4096:])
    return 2.0 * n * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


class ClockProbe:
    """Average shader clock while other streams work: ``with ClockProbe(dev) as p: <launch work on the current stream>``; ``p.mhz`` afterwards.
    One wave on a side stream samples s_memtime / s_memrealtime at its start and when the observed stream reaches the end marker (bounded wait)."""

    def __init__(self, device, limit_seconds=2.0):
        self.dev, self.limit = device, limit_seconds
        self.state = torch.zeros(6, dtype=torch.int64, device=device)
        self.side = side_stream(device, 3)
        self.mhz = self.seconds = None
        self.timed_out = False

    def __enter__(self):
        self.side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(self.side):
            _check(lib().whmr_clock_probe_begin(self.state.data_ptr(), float(self.limit), self.side.cuda_stream), 'whmr_clock_probe_begin')
        return self

    def __exit__(self, *exc):
        _check(lib().whmr_clock_probe_end(self.state.data_ptr(), _stream()), 'whmr_clock_probe_end')
        torch.cuda.current_stream(self.dev).wait_stream(self.side)
        torch.cuda.synchronize(self.dev)
        st = self.state.tolist()
        ticks = st[3] - st[1]
        self.seconds = ticks / 1e8
        self.mhz = (st[4] - st[2]) / max(ticks, 1) * 100.0
        self.timed_out = bool(st[5])
        return False
