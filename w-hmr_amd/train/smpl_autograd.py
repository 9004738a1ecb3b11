"""SMPL forward with saved activations + hand-written backward (north_star "forward/backward path": 24-joint kinematic chain and
6890-vertex linear blend skinning).

Reference semantics: autograd through ``pare.models.SMPL`` (smplx lbs) as called at models/whmr.py:132-137 with ``pose2rot=False``
in training (no Gram-Schmidt: whmr.py:129 is eval-only), driven by ``core/trainer.py:410-470``.  Differentiated inputs: betas
[B,10] and rotmats [B,24,3,3]; outputs: vertices [B,6890,3], joints [B,49,3] (+ smpl_joints [B,45,3], markers [B,67,3]).
The angle-axis copy of the pose that goes into ``theta`` (whmr.py:174) carries no gradient here.

Backward = 3 HIP launches + one fp32 GEMM (see smpl_lbs.hip): joints -> vertices / posed joints, skinning backward
(vertex-parallel, per-block partial sums of the 24 skinning-transform gradients), [B,20670] x [posedirs ; shapedirs]^T,
reverse kinematic chain.  Fixed summation order everywhere (no atomics).
"""
import torch

from .. import _lib as L


def _bwd_weights(smpl):
    """[217, 20670] = posedirs (207 rows) stacked on shapedirs viewed as [10, (v, c)]: the weight operand of the d_vposed GEMM."""
    keep = smpl._dev_cache[2]
    if 'bwd_w' not in keep:
        s2 = smpl.shapedirs.reshape(smpl.NUM_VERTS * 3, 10).t()
        keep['bwd_w'] = torch.cat([smpl.posedirs, s2], 0).contiguous()
    return keep['bwd_w']


@torch.no_grad()
def smpl_forward_train(smpl, betas, rotmats, want_smpl_joints=False, want_markers=False):
    """-> (verts, joints49, smpl_joints45 | None, markers | None, saved)."""
    if not betas.is_cuda:
        raise RuntimeError('whmr_amd.SMPL runs on a HIP device only (no CPU fallback)')
    B, dev = betas.shape[0], betas.device
    f32 = dict(dtype=torch.float32, device=dev)
    m = smpl._model()
    betas = betas.float().contiguous()
    rot = rotmats.reshape(B, 216).float().contiguous()
    A = torch.empty(B, 24, 12, **f32)
    pj = torch.empty(B, 24, 3, **f32)
    pf = torch.empty(B, 207, **f32)
    L.smpl_pose_chain(m, rot, betas, False, None, None, A, pj, pf)
    pose_off = torch.empty(B, smpl.NUM_VERTS * 3, **f32)
    L.gemm(pf, smpl._dev_cache[2]['posedirs_t'], pose_off)
    verts = torch.empty(B, smpl.NUM_VERTS, 3, **f32)
    L.smpl_skin(m, betas, pf, A, verts, pose_off)
    joints = torch.empty(B, 49, 3, **f32)
    sj = torch.empty(B, 45, 3, **f32) if want_smpl_joints else None
    mk = torch.empty(B, m.n_markers, 3, **f32) if (want_markers and m.n_markers) else None
    L.smpl_joints(m, verts, pj, joints, sj, mk)
    return verts, joints, sj, mk, (betas, rot, A, pose_off)


@torch.no_grad()
def smpl_backward(smpl, saved, d_verts=None, d_joints=None, d_smpl_joints=None, d_markers=None):
    """Cotangents (each optional) -> (d_betas [B,10], d_rotmats [B,24,3,3])."""
    betas, rot, A, pose_off = saved
    B, dev = betas.shape[0], betas.device
    f32 = dict(dtype=torch.float32, device=dev)
    m = smpl._model()
    c = lambda t: None if t is None else t.float().contiguous()
    dv = torch.zeros(B, smpl.NUM_VERTS, 3, **f32) if d_verts is None else d_verts.float().contiguous().clone()
    d_joints, d_smpl_joints, d_markers = c(d_joints), c(d_smpl_joints), c(d_markers)
    R = 33 if d_smpl_joints is not None else 9
    dpj = torch.empty(B, 24, 3, **f32)
    dregd = torch.empty(B, R, 3, **f32)
    L.smpl_joints_bwd(m, d_joints, d_smpl_joints, d_markers, dv, dpj, dregd)
    dvp = torch.empty(B, smpl.NUM_VERTS * 3, **f32)
    dA = torch.empty(B, (smpl.NUM_VERTS + 127) // 128, 288, **f32)
    L.smpl_skin_bwd(m, betas, A, pose_off, dv, dregd, dvp, dA)
    dpfb = torch.empty(B, 217, **f32)
    L.gemm(dvp, _bwd_weights(smpl), dpfb)
    d_rot = torch.empty(B, 24, 3, 3, **f32)
    d_betas = torch.empty(B, 10, **f32)
    L.smpl_chain_bwd(m, rot, betas, dA, dpj, dpfb, d_rot, d_betas)
    return d_betas, d_rot


class SMPLFn(torch.autograd.Function):
    """verts, joints49, smpl_joints45, markers = SMPLFn.apply(betas, rotmats, smpl_module)."""

    @staticmethod
    def forward(ctx, betas, rotmats, smpl):
        ctx.set_materialize_grads(False)            # outputs no loss reaches arrive as None (smpl_backward takes each cotangent as optional), not as zero-filled tensors
        verts, joints, sj, mk, saved = smpl_forward_train(smpl, betas, rotmats, want_smpl_joints=True, want_markers=True)
        ctx.smpl, ctx.saved = smpl, saved
        if mk is None:
            mk = verts.new_zeros(verts.shape[0], 0, 3)
        return verts, joints, sj, mk

    @staticmethod
    def backward(ctx, d_verts, d_joints, d_sj, d_mk):
        if d_mk is not None and d_mk.numel() == 0:
            d_mk = None
        d_betas, d_rot = smpl_backward(ctx.smpl, ctx.saved, d_verts, d_joints, d_sj, d_mk)
        ctx.saved = None
        return d_betas, d_rot, None
