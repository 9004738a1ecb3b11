"""IUV ground-truth rasteriser (csrc/rasterize.hip) against the CPU oracle (oracle/raster.py) and the analytic known answers."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _world(u, v, z, tz, f=1000.0, c=28.0, s=4.0):
    zz = z + tz
    return [(u - c) * s * zz / f, (v - c) * s * zz / f, z]


def test_single_triangle_depth_order_and_edge_rule(dev):
    from oracle import raster as OR
    from whmr_amd.utils.renderer import IUV_Renderer
    tz = 2 * 1000.0 / (224 * 1.0 + 1e-9)
    tri = [_world(10.0, 10.0, 0.0, tz), _world(30.0, 10.0, 0.0, tz), _world(10.0, 30.0, 0.0, tz)]
    near = [_world(10.0, 10.0, -0.5, tz), _world(30.0, 10.0, -0.5, tz), _world(10.0, 30.0, -0.5, tz)]
    verts = torch.tensor([tri + near], dtype=torch.float32)
    tex = torch.tensor([[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0], [0.5, 0, 0], [0, 0.5, 0], [0, 0, 0.5]])
    cam = torch.tensor([[1.0, 0.0, 0.0]])
    yy, xx = np.mgrid[0:56, 0:56] + 0.5
    expect = (xx > 10) & (yy > 10) & ((xx - 10) + (yy - 10) < 20)
    # the fp32 projection moves the vertices by ~1e-6 px: centres exactly ON the hypotenuse (x + y = 40) may fall on either side on the device
    # (the exact strict-interior rule is pinned on the float64 oracle, tests/test_oracle_cpu.py); every other pixel is decided
    on_edge = (xx - 10) + (yy - 10) == 20
    for faces in ([[0, 1, 2], [3, 4, 5]], [[3, 4, 5], [0, 1, 2]]):
        r = IUV_Renderer(orig_size=(224, 224), output_size=(56, 56), dp=dict(vert_mapping=None, faces=torch.tensor(faces), textures_vts=tex))
        out, fid = r.verts2iuvimg(verts.to(dev), cam.to(dev), want_faces=True)
        fid = fid.cpu().numpy()[0]
        assert np.array_equal((fid >= 0)[~on_edge], expect[~on_edge])
        assert (fid[expect] == faces.index([3, 4, 5])).all()                            # the nearer triangle wins whatever the face order
        ref, rfid = OR.rasterize(verts.numpy(), faces, tex.numpy(), cam.numpy(), 1000.0, (224, 224), (56, 56))
        assert np.array_equal(fid[~on_edge], rfid[0][~on_edge])
        assert np.abs(out.cpu().numpy() - ref)[0][:, expect].max() < 1e-5
        assert np.allclose(out.cpu().numpy()[0].sum(0)[expect], 0.5, atol=1e-5)
    # equal depth: the smaller face index wins (deterministic z-buffer key)
    same = torch.tensor([tri + tri], dtype=torch.float32)
    r = IUV_Renderer(orig_size=(224, 224), output_size=(56, 56), dp=dict(vert_mapping=None, faces=torch.tensor([[3, 4, 5], [0, 1, 2]]), textures_vts=tex))
    _, fid = r.verts2iuvimg(same.to(dev), cam.to(dev), want_faces=True)
    assert (fid.cpu().numpy()[0][expect] == 0).all()
    # shared diagonal of a square: centres on the edge belong to neither triangle
    sq = torch.tensor([[_world(10.0, 10.0, 0.0, tz), _world(30.0, 10.0, 0.0, tz), _world(30.0, 30.0, 0.0, tz), _world(10.0, 30.0, 0.0, tz)]], dtype=torch.float32)
    r = IUV_Renderer(orig_size=(224, 224), output_size=(56, 56), dp=dict(vert_mapping=None, faces=torch.tensor([[0, 1, 2], [0, 2, 3]]), textures_vts=torch.ones(4, 3)))
    o, fid = r.verts2iuvimg(sq.to(dev), cam.to(dev), want_faces=True)
    inside = (xx > 10) & (xx < 30) & (yy > 10) & (yy < 30)
    got = fid.cpu().numpy()[0]
    assert np.array_equal((got >= 0)[xx != yy], inside[xx != yy])
    assert (got[inside & (xx > yy)] == 0).all() and (got[inside & (xx < yy)] == 1).all()   # each half belongs to its own triangle
    with pytest.raises(RuntimeError):
        r.verts2iuvimg(sq, cam)


def test_iuv_rasterizer_matches_oracle_on_a_posed_mesh(dev, assets):
    """trainer shape (core/trainer.py:442-464): the synthetic SMPL mesh (6890 vertices, 13776 faces) posed by the oracle, DensePose-style vertex
    duplication map, 256 x 256 frame -> 128 x 128 IUV, B = 3 incl. an off-centre and a far camera; + the target maps of iuv_img2map"""
    from oracle import raster as OR
    from oracle import smpl as OS
    from whmr_amd.utils.iuvmap import iuv_img2map
    from whmr_amd.utils.renderer import IUV_Renderer
    g = torch.Generator().manual_seed(2)
    B = 3
    V = 6890
    rot = torch.eye(3).expand(B, 24, 3, 3).clone()
    verts, _ = OS.smpl_forward(torch.randn(B, 10, generator=g) * 0.5, rot, assets['smpl'])
    # the synthetic SMPL has no real connectivity: triangle soup of 13776 small faces from two spatial orderings of the rest vertices (neighbours
    # in a coarse x / y (resp. z / y) cell order become triangles) -- many overlaps in depth, like a posed body seen from one side
    vt = assets['smpl']['v_template']
    o1 = torch.argsort((vt[:, 1] * 20).floor() * 1000 + vt[:, 0] * 10)
    o2 = torch.argsort((vt[:, 1] * 20).floor() * 1000 + vt[:, 2] * 10)
    faces = torch.cat([torch.stack([o1[:-2], o1[1:-1], o1[2:]], 1), torch.stack([o2[:-2], o2[2:], o2[1:-1]], 1)])[:13776].to(torch.int32).contiguous()
    vmap = torch.cat([torch.arange(V), torch.randint(0, V, (939,), generator=g)])        # 7829 DensePose vertices: the SMPL ones + seam duplicates
    tex = torch.stack([torch.randint(1, 25, (vmap.numel(),), generator=g).float() / 24, torch.rand(vmap.numel(), generator=g),
                       torch.rand(vmap.numel(), generator=g)], 1)
    cam = torch.tensor([[0.9, 0.0, 0.0], [0.6, 0.2, -0.1], [1.4, -0.05, 0.1]])
    r = IUV_Renderer(orig_size=(256, 256), output_size=(128, 128), dp=dict(vert_mapping=vmap, faces=faces, textures_vts=tex))
    out, fid = r.verts2iuvimg(verts.to(dev), cam.to(dev), want_faces=True)
    assert out.shape == (B, 3, 128, 128)
    ref, rfid = OR.rasterize(verts.numpy(), faces.numpy(), tex.numpy(), cam.numpy(), 1000.0, (256, 256), (128, 128), vmap=vmap.numpy())
    fid, o = fid.cpu().numpy(), out.cpu().numpy()
    covered = (rfid >= 0).mean()
    assert covered > 0.02                                                               # the mesh is in view
    # fp32 (device) vs float64 (oracle): a pixel centre within rounding of an edge / two faces within rounding in depth may resolve differently
    differ = (fid != rfid).mean()
    print('covered %.3f of the pixels; face index differs on %.2e of them' % (covered, differ))
    assert differ < 2e-3
    agree = (fid == rfid) & (rfid >= 0)
    err = np.abs(o - ref)[np.broadcast_to(agree[:, None], o.shape)]
    assert err.max() < 1e-3 and np.median(err) < 1e-6                                  # sliver triangles amplify the fp32 rounding of the barycentrics
    assert not o[np.broadcast_to((fid < 0)[:, None], o.shape)].any()
    # deterministic: a second run gives identical bits
    out2 = r.verts2iuvimg(verts.to(dev), cam.to(dev))
    assert torch.equal(out, out2)
    # target maps (utils/iuvmap.py:67-110) on the vitpose crop of the image (trainer.py:454-455)
    crop = out[:, :, :, 16:-16]
    U, Vm, I, A = iuv_img2map(crop)
    rU, rV, rI, rA = OR.iuv_img2map(crop.cpu().numpy().astype(np.float64))
    assert I.shape == (B, 25, 128, 96) and A.shape == (B, 15, 128, 96)
    assert np.array_equal(I.cpu().numpy(), rI) and np.array_equal(A.cpu().numpy(), rA)
    assert np.allclose(U.cpu().numpy(), rU) and np.allclose(Vm.cpu().numpy(), rV)
    assert torch.equal(I.sum(1), torch.ones_like(I[:, 0]))                              # exactly one part (or background) per pixel


def test_aux_supervision_targets_and_losses(dev, assets):
    """core/trainer.py:442-482: IUV image of the fitted mesh -> vitpose crop -> target maps -> body_uv_losses on the dp_head outputs; the device
    pipeline (HIP rasteriser + tensor arithmetic) against the CPU oracle's image fed through the same loss formulas on the host"""
    from oracle import raster as OR
    from oracle import smpl as OS
    from oracle import synth
    from whmr_amd.train.aux_supervision import aux_supervision_loss, gt_camera_from_translation, render_iuv_targets
    from whmr_amd.utils.renderer import IUV_Renderer
    import torch.nn.functional as F
    B = 2
    tables = synth.make_densepose_tables(0, assets)
    verts, _ = OS.smpl_forward(torch.zeros(B, 10), torch.eye(3).expand(B, 24, 3, 3).clone(), assets['smpl'])
    cam_t = torch.tensor([[0.05, -0.02, 45.0], [-0.1, 0.1, 60.0]])
    cam = gt_camera_from_translation(cam_t, focal_length=5000.0, img_res=256)
    assert torch.allclose(cam[:, 0], (2 * 5000.0 / 256) / cam_t[:, 2]) and torch.equal(cam[:, 1:], cam_t[:, :2])
    r = IUV_Renderer(orig_size=(256, 256), output_size=(128, 128), dp=tables)
    img, uvia = render_iuv_targets(r, verts.to(dev), cam.to(dev))
    assert img.shape == (B, 3, 128, 96)
    ref, _ = OR.rasterize(verts.numpy(), tables['faces'].numpy(), tables['textures_vts'].numpy(), cam.numpy(), 1000.0, (256, 256), (128, 128),
                          vmap=tables['vert_mapping'].numpy())
    ref = ref[:, :, :, 16:-16]
    assert (ref[:, 0] > 0).mean() > 0.02
    rU, rV, rI, rA = (torch.from_numpy(a).float() for a in OR.iuv_img2map(ref))
    assert (uvia[2].cpu() != rI).float().mean() < 1e-3                              # a handful of edge pixels may resolve differently in fp32
    g = torch.Generator().manual_seed(1)
    dp = {k: torch.randn(B, c, 128, 96, generator=g) for k, c in (('predict_u', 25), ('predict_v', 25), ('predict_uv_index', 25), ('predict_ann_index', 15))}
    dpd = {k: v.to(dev).requires_grad_(True) for k, v in dp.items()}
    loss = aux_supervision_loss([dpd], uvia)
    loss.backward()
    # the same formulas on the host with the oracle's maps (core/trainer.py:273-297)
    idx = F.cross_entropy(dp['predict_uv_index'].permute(0, 2, 3, 1).reshape(-1, 25), rI.argmax(1).view(-1))
    ann = F.cross_entropy(dp['predict_ann_index'].permute(0, 2, 3, 1).reshape(-1, 15), rA.argmax(1).view(-1))
    fg = rI > 0
    lu = F.smooth_l1_loss(dp['predict_u'][fg], rU[fg], reduction='sum') / B * 0.125
    lv = F.smooth_l1_loss(dp['predict_v'][fg], rV[fg], reduction='sum') / B * 0.125
    ref_loss = (idx + ann + lu + lv).item()
    assert abs(loss.item() - ref_loss) < 2e-3 * abs(ref_loss), (loss.item(), ref_loss)
    assert all(v.grad is not None and torch.isfinite(v.grad).all() for v in dpd.values())


def _iuv_loss_reference(logits, img, w):
    """core/trainer.py:273-297 on utils/iuvmap.py:67-110 targets (the oracle's maps), float64 on the host: the reference's boolean-index form"""
    import torch.nn.functional as F
    from oracle import raster as OR
    U, V, I, A = (torch.from_numpy(a) for a in OR.iuv_img2map(img.double().numpy()))
    y = logits.double().requires_grad_(True)
    u, v, idx, ann = (t.permute(0, 3, 1, 2) for t in torch.split(y, [25, 25, 25, 15], dim=-1))
    B = y.shape[0]
    li = F.cross_entropy(idx.permute(0, 2, 3, 1).reshape(-1, 25), I.argmax(1).view(-1))
    la = F.cross_entropy(ann.permute(0, 2, 3, 1).reshape(-1, 15), A.argmax(1).view(-1))
    fg = I > 0
    lu = F.smooth_l1_loss(u[fg], U[fg], reduction='sum') / B * w
    lv = F.smooth_l1_loss(v[fg], V[fg], reduction='sum') / B * w
    return y, torch.stack([lu, lv, li, la])


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_iuv_losses_fused_kernel_matches_the_map_form(dev, dtype):
    """csrc/iuv_loss.hip (whmr_iuv_losses / whmr_iuv_losses_bwd) against body_uv_losses' formulas on iuv_img2map targets in float64: every part
    id incl. background, a value outside 0..24 (no indicator: counts as class 0, no U / V term), |u_pred - U| on both sides of the smooth-L1 knee,
    a cropped (strided) image view, a ragged last block (P % 128 != 0), non-uniform upstream gradients; bf16 logits in the padded [P, 128] layout."""
    from whmr_amd import _lib as L
    from whmr_amd.train.aux_supervision import IUVLossFn
    g = torch.Generator().manual_seed(5)
    B, H, W, w = 3, 13, 11, 0.125
    part = torch.randint(0, 25, (B, H, W + 8), generator=g).float()
    part[0, 0, 4:9] = torch.tensor([25.0, 0.0, 24.0, 26.0, 12.0])                         # 25, 26: outside the indicator range
    full = torch.stack([part / 24.0, torch.rand(B, H, W + 8, generator=g), torch.rand(B, H, W + 8, generator=g)], 1)
    img = full[:, :, :, 4:-4]                                                            # strided view, like the vitpose crop
    logits = torch.randn(B, H, W, 90, generator=g) * 1.5
    if dtype == 'bf16':
        logits = logits.bfloat16().float()
        buf = torch.zeros(B * H * W, 128, dtype=torch.bfloat16, device=dev)
        buf[:, :90] = logits.view(-1, 90).to(dev)
        y = buf.view(B, H, W, 128)[..., :90]
    else:
        y = logits.to(dev)
    y = y.detach().requires_grad_(True)
    imgd = full.to(dev)[:, :, :, 4:-4]
    assert not imgd.is_contiguous()
    ref_y, ref = _iuv_loss_reference(logits, img, w)
    out = IUVLossFn.apply(y, imgd, w)
    assert out.shape == (4,) and torch.allclose(out.cpu().double(), ref.detach(), rtol=2e-6, atol=0)
    up = torch.tensor([2.0, 0.5, 3.0, 1.5])
    (ref * up.double()).sum().backward()
    (out * up.to(dev)).sum().backward()
    got, want = y.grad.float().cpu().double(), ref_y.grad
    scale = want.abs().max()
    tol = 2e-6 if dtype == 'fp32' else 2.0 ** -8                                        # bf16 gradient: one rounding of each entry
    assert ((got - want).abs() <= tol * want.abs() + 1e-7 * scale).all(), (got - want).abs().max() / scale
    # the raw kernel output: padded columns are zero, and the same bits on a second run (fixed-order sums)
    dyp = L.iuv_losses_bwd(y.detach(), imgd, w, up.to(dev), 128)
    assert dyp.shape == (B * H * W, 128) and not dyp[:, 90:].any() and torch.equal(dyp[:, :90].reshape(B, H, W, 90), y.grad)
    assert torch.equal(L.iuv_losses(y.detach(), imgd, w), out.detach())
    with pytest.raises(AssertionError):
        L.iuv_losses(y.detach()[..., :89], imgd, w)


def test_iuv_losses_fused_path_through_the_head_convolution(dev):
    """IUV head conv (ConvNHWCFn, one implicit GEMM over 90 channels) -> losses: the fused node (its padded gradient handed to the convolution's
    backward as is) against the map form (lazy NCHW fp32 views -> body_uv_losses through torch autograd) on the same weights and features."""
    from whmr_amd.train import heads_autograd as HA
    from whmr_amd.train.aux_supervision import IUVHeadOutput, aux_supervision_loss
    from whmr_amd.utils.iuvmap import iuv_img2map
    g = torch.Generator().manual_seed(7)
    B, H, W, Cin = 2, 16, 12, 256
    x = (torch.randn(B, H, W, Cin, generator=g) * 0.5).to(dev).bfloat16()
    part = torch.randint(0, 25, (B, H, W), generator=g).float()
    img = torch.stack([part / 24.0, torch.rand(B, H, W, generator=g), torch.rand(B, H, W, generator=g)], 1).to(dev)
    res, hits = [], []
    orig = HA._padded_base
    HA._padded_base = lambda *a_: hits.append(orig(*a_)) or hits[-1]                     # records whether the padded operand was found behind the view
    for fused in (True, False):
        wt = (torch.randn(90, Cin, 3, 3, generator=torch.Generator().manual_seed(9)) * 0.02).to(dev).requires_grad_(True)
        bs = torch.zeros(90, device=dev).requires_grad_(True)
        xin = x.clone().requires_grad_(True)
        d = IUVHeadOutput(HA.ConvNHWCFn.apply(xin, wt, 1, torch.bfloat16, 1, bs))
        assert set(d) == set(IUVHeadOutput.KEYS) and dict.__len__(d) == 0              # nothing materialised yet
        loss = aux_supervision_loss([d], None if fused else iuv_img2map(img), iuv_image_gt=img if fused else None)
        assert dict.__len__(d) == (0 if fused else 4)
        loss.backward()
        assert (hits[-1] is not None) == fused                                           # fused: the convolution took the loss node's padded buffer as is
        res.append((loss.item(), wt.grad.clone(), bs.grad.clone(), xin.grad.float().clone()))
    HA._padded_base = orig
    assert d['predict_u'].shape == (B, 25, H, W) and d['predict_ann_index'].shape == (B, 15, H, W) and d['predict_u'].dtype == torch.float32
    (lf, wf, bf, xf), (lm, wm, bm, xm) = res
    assert abs(lf - lm) < 1e-5 * abs(lm)
    for a, b in ((wf, wm), (bf, bm), (xf, xm)):                                          # same bf16 operand roundings up to the order of one sum
        assert (a - b).abs().max() <= 2e-2 * b.abs().max() and (a - b).norm() <= 5e-3 * b.norm()
