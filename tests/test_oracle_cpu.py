"""CPU suite: the oracle against the committed reference fixtures + analytic known answers (no GPU needed)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN


def _rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def test_synthetic_weights_are_reproducible(state_dict):
    """the fixtures were produced with these weights: any drift of the generator would silently unpin parity"""
    g = np.load(os.path.join(GOLDEN, 'whmr_b2.npz'))
    for k, s in zip(g['weight_keys'], g['weight_sums']):
        assert abs(state_dict[str(k)].double().sum().item() - s) <= 1e-9 * max(1.0, abs(s)), k


def test_oracle_whmr_forward_matches_reference_fixture(assets, state_dict):
    from oracle import whmr as OW
    g = np.load(os.path.join(GOLDEN, 'whmr_b2.npz'))
    t = lambda k: torch.from_numpy(g['in_' + k])
    taps = {}
    with torch.no_grad():
        out = OW.whmr_forward(state_dict, assets, t('x'), t('center'), t('scale'), t('bbox_height'), t('orig_shape'),
                              t('bbox_info'), full_x=t('full_x'), taps=taps)
    for k, v in out.items():
        assert _rel(v, g['out_' + k]) < 1e-5, k
    assert _rel(taps['s_feat0'], g['s_feat']) < 1e-5
    assert _rel(taps['Tz'], g['Tz']) < 1e-5
    for i in range(3):
        assert _rel(taps['ref_feature'][i], g['ref_feature%d' % i]) < 1e-5
        fm = taps['fmaps'][i]
        assert _rel(fm.reshape(-1)[torch.from_numpy(g['fmap%d_idx' % i])], g['fmap%d_val' % i]) < 1e-5
        assert abs(fm.double().sum().item() - g['fmap%d_sum' % i][0]) < 1e-3 * g['fmap%d_sum' % i][1]


def test_oracle_vit224_matches_reference_fixture():
    from oracle import synth
    from oracle.vit import vit_forward
    g = np.load(os.path.join(GOLDEN, 'vit224_b2.npz'))
    with torch.no_grad():
        out = vit_forward(synth.make_vit_state(1, (224, 224)), torch.from_numpy(g['x']))
    assert out.shape == (2, 768, 14, 14)
    assert _rel(out, g['s_feat']) < 1e-5


def test_oracle_geometry_matches_reference_fixture():
    from oracle import geometry as OG
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, 'geometry.npz')).items()}
    assert _rel(OG.rotation_matrix_to_angle_axis(g['in_R']), g['out_aa']) < 1e-6
    assert _rel(OG.batch_rodrigues(g['in_aa_in']), g['out_rod']) < 1e-6
    assert _rel(OG.rot6d_to_rotmat(g['in_r6']), g['out_r6_to_R']) < 1e-6
    assert _rel(OG.unbiased_gram_schmidt(g['in_m33']), g['out_gs']) < 1e-6
    assert _rel(OG.projection(g['in_pts'], g['in_cam']), g['out_proj']) < 1e-6
    assert _rel(OG.perspective_projection(g['in_pts'], torch.eye(3).unsqueeze(0), g['in_tr'], g['in_fl'], g['in_cc']),
                g['out_persp']) < 1e-6
    et = OG.estimate_translation(g['in_et_S'], g['in_et_j2d'], 5000., (224., 224.))
    assert _rel(et, g['out_est_trans']) < 1e-6
    # known answer: exact projections with unit confidence recover the translation that produced them
    S = g['in_et_S'][:2].clone()
    t = torch.tensor([[0.2, -0.1, 5.0], [-0.3, 0.4, 8.0]])
    p = S + t[:, None]
    j2d = torch.cat([5000. * p[..., :2] / p[..., 2:] + 112., torch.ones(2, 49, 1)], -1)
    assert torch.allclose(OG.estimate_translation(S, j2d, 5000., (224., 224.)), t, atol=2e-3)


def test_geometry_known_answers():
    from oracle import geometry as OG
    gen = torch.Generator().manual_seed(0)
    R = OG.rot6d_to_rotmat(torch.randn(32, 6, generator=gen))
    eye = torch.eye(3).expand(32, 3, 3)
    assert torch.allclose(R @ R.transpose(1, 2), eye, atol=1e-5) and torch.allclose(torch.linalg.det(R), torch.ones(32), atol=1e-5)
    G = OG.unbiased_gram_schmidt(torch.randn(4, 8, 3, 3, generator=gen)).reshape(-1, 3, 3)
    assert torch.allclose(G @ G.transpose(1, 2), torch.eye(3).expand(32, 3, 3), atol=1e-5)
    aa = torch.randn(32, 3, generator=gen) * 0.8
    assert torch.allclose(OG.rotation_matrix_to_angle_axis(OG.batch_rodrigues(aa)), aa, atol=1e-4)   # round trip
    assert torch.allclose(OG.batch_rodrigues(torch.zeros(1, 3)), torch.eye(3).unsqueeze(0), atol=1e-6)


def _rx(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])


def _ry(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def _rz(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def _euler_cases():
    return [(0.37, 0.0, 0.0), (0.0, 0.0, -0.52), (0.0, 0.41, 0.0), (0.37, 0.0, -0.52), (-0.6, 0.0, 0.6), (0.3, -0.2, 0.5)]


def test_batch_euler2matrix_composition_order():
    """pare.utils.geometry.batch_euler2matrix (3P, = DECA's euler_to_quaternion o quaternion_to_rotation_matrix): R = Rx(x) . Ry(y) . Rz(z).
    The reference calls it with [pitch, 0, roll] (models/whmr.py:521-522): pitch-only = Rx, roll-only = Rz, both = Rx(pitch) . Rz(roll) --
    NOT Rz . Rx (the two differ in the sign of q_y = -+ sx sz)."""
    from oracle import geometry as OG
    e = torch.tensor(_euler_cases(), dtype=torch.float64)
    R = OG.batch_euler2matrix(e).numpy()
    for (x, y, z), r in zip(_euler_cases(), R):
        assert np.allclose(r, _rx(x) @ _ry(y) @ _rz(z), atol=1e-12), (x, y, z)
    both = OG.batch_euler2matrix(torch.tensor([[0.37, 0.0, -0.52]], dtype=torch.float64))[0].numpy()
    assert np.allclose(both, _rx(0.37) @ _rz(-0.52), atol=1e-12)
    assert np.abs(both - _rz(-0.52) @ _rx(0.37)).max() > 1e-2            # the other order is a different matrix
    # the quaternion itself, component by component (the published formula)
    h = e[3] / 2
    cx, cy, cz, sx, sy, sz = torch.cos(h[0]), torch.cos(h[1]), torch.cos(h[2]), torch.sin(h[0]), torch.sin(h[1]), torch.sin(h[2])
    q = torch.stack([cx * cy * cz - sx * sy * sz, cx * sy * sz + cy * cz * sx, cx * cz * sy - sx * cy * sz, cx * cy * sz + sx * cz * sy])
    assert torch.allclose(OG.quat_to_rotmat(q[None]), torch.from_numpy(both)[None], atol=1e-12)


def test_product_euler2matrix_matches_oracle():
    """the device-side post-processing of cam_model uses the same tensor expression (pure torch, runs on CPU)"""
    from oracle import geometry as OG
    from whmr_amd.models.cam_model import batch_euler2matrix
    e = torch.tensor(_euler_cases(), dtype=torch.float32)
    assert torch.allclose(batch_euler2matrix(e), OG.batch_euler2matrix(e), atol=1e-6)
    assert torch.allclose(batch_euler2matrix(e[3:4])[0].double(), torch.from_numpy(_rx(0.37) @ _rz(-0.52)), atol=1e-6)


def test_smpl_known_answers(assets):
    """pins the un-vendored SMPL arithmetic analytically: identity pose, rigid root rotation, single-joint rotation"""
    from oracle import geometry as OG
    from oracle import smpl as OS
    s = assets['smpl']
    betas = torch.tensor([[0.4, -0.7, 0.2, 0, 0, 0.1, 0, 0, -0.3, 0.5]])
    eye = torch.eye(3).expand(1, 24, 3, 3).clone()
    v_shaped = s['v_template'] + torch.einsum('l,mkl->mk', betas[0], s['shapedirs'])
    v, j = OS.lbs(betas, eye, s)
    J = s['J_regressor'] @ v_shaped
    assert _rel(v[0], v_shaped) < 1e-6 and _rel(j[0], J) < 1e-6
    R = OG.batch_rodrigues(torch.tensor([[0.2, 0.9, -0.4]]))[0]
    rot = eye.clone()
    rot[0, 0] = R
    v, j = OS.lbs(betas, rot, s)
    assert _rel(v[0], (v_shaped - J[0]) @ R.t() + J[0]) < 1e-5
    assert _rel(j[0], (J - J[0]) @ R.t() + J[0]) < 1e-5
    # rotate a leaf joint (23): joints not below it stay put
    rot = eye.clone()
    rot[0, 23] = R
    _, j = OS.lbs(betas, rot, s)
    assert _rel(j[0], J) < 1e-6
    verts, j49 = OS.smpl_forward(betas, eye, s)
    assert j49.shape == (1, 49, 3)
    assert torch.equal(j49[0, 0], verts[0, 332])                      # 'OP Nose' = picked vertex 332
    assert _rel(j49[0, 8], J[0]) < 1e-6                               # 'OP MidHip' = SMPL joint 0


def test_maf_known_answers(state_dict):
    """grid_sample at integer texel coordinates returns the exact texel; far outside the map -> zeros (padding)"""
    from oracle import whmr as OW
    fmap = torch.randn(1, 256, 5, 4, generator=torch.Generator().manual_seed(1))
    pts = torch.tensor([[[-1.0, -1.0], [1.0, 1.0], [-1.0 + 2 * 2 / 3, -1.0 + 2 * 3 / 4], [3.0, 3.0]]])
    _, pf = OW.maf_sampling(state_dict, pts, fmap, 'maf_extractor.0.')
    assert torch.allclose(pf[0, :, 0], fmap[0, :, 0, 0], atol=1e-6) and torch.allclose(pf[0, :, 1], fmap[0, :, 4, 3], atol=1e-6)
    assert torch.allclose(pf[0, :, 2], fmap[0, :, 3, 2], atol=1e-5)
    assert (pf[0, :, 3] == 0).all()


def test_oracle_config1_hmr_matches_reference_fixture(assets):
    """BASELINE config #1 (the reference's own CPU-runnable case): pose_resnet/HMR R50 + regressor + SMPL, 1 x 224 x 224"""
    from oracle import smpl as OS
    from oracle import synth
    from oracle.hmr import hmr_forward, pose_resnet_global
    g = np.load(os.path.join(GOLDEN, 'hmr_b1.npz'))
    sd = synth.make_hmr_state(0, assets)
    x = torch.from_numpy(g['x'])
    with torch.no_grad():
        rot, shape, cam = hmr_forward(sd, x)
        f, gf = pose_resnet_global(sd, x)
    assert f.shape == (1, 2048, 7, 7) and gf.shape == (1, 2048)
    assert _rel(rot, g['rotmat']) < 1e-5 and _rel(shape, g['shape']) < 1e-5 and _rel(cam, g['cam']) < 1e-5
    assert _rel(OS.smpl_forward(shape, rot, assets['smpl'])[0], g['verts']) < 1e-5


def test_crop_oracle_known_answers():
    """oracle/crop.py (cv2 restatement, parity unpinned): identity box reproduces the frame; integer / half-pixel shifts are exact."""
    import numpy as np
    from oracle import crop as OC
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(64, 64, 3), dtype=np.uint8)
    patch, trans = OC.generate_patch_image_cv(img, 32.0, 32.0, 64.0, 64.0, 64, 64, False, 1.0, 0)
    assert np.allclose(trans, [[1, 0, 0], [0, 1, 0]], atol=1e-12) and np.array_equal(patch, img)
    shifted, _ = OC.generate_patch_image_cv(img, 35.0, 30.0, 64.0, 64.0, 64, 64, False, 1.0, 0)      # patch(x, y) = img(x + 3, y - 2)
    assert np.array_equal(shifted[2:, :61], img[:62, 3:]) and not shifted[:2].any() and not shifted[:, 61:].any()
    half, _ = OC.generate_patch_image_cv(img, 32.5, 32.0, 64.0, 64.0, 64, 64, False, 1.0, 0)         # half-pixel: mean of two neighbours
    ref = (img[:, :-1].astype(np.int64) + img[:, 1:] + 1) >> 1
    assert np.array_equal(half[:, :63], ref.astype(np.uint8))
    t = OC.to_tensor_normalize(img)
    assert t.shape == (3, 64, 64) and t.dtype == np.float32
    assert np.allclose(t[0], (img[..., 0] / 255.0 - 0.485) / 0.229, atol=1e-6)


def test_crop_oracle_against_an_independent_bilinear_warp():
    """cv2 is not installed, so oracle/crop.py cannot be pinned bit for bit; this pins its GEOMETRY and interpolation against an independent
    implementation: scipy.ndimage.map_coordinates (order 1, zero padding blended in at the frame edge: 'grid-constant') evaluated at the exact inverse-affine coordinates of every output
    pixel.  OpenCV's scheme quantises coordinates to 1/32 pixel and the weights to 5 bits, so on a smooth image the two agree to a couple of grey
    levels -- a wrong pixel-centre convention, transposed axes, a flipped inverse or a border rule would be off by tens."""
    import numpy as np
    from scipy import ndimage
    from oracle import crop as OC
    yy, xx = np.mgrid[0:240, 0:320].astype(np.float64)
    img = np.stack([127.5 + 100 * np.sin(xx / 23.0) * np.cos(yy / 31.0), 127.5 + 90 * np.cos(xx / 17.0 + yy / 41.0), 60 + 0.5 * xx + 0.2 * yy], -1)
    img = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    for (cx, cy, bw, bh, pw, ph, scale) in ((160.3, 120.7, 150.0, 150.0, 224, 224, 1.2), (40.0, 200.0, 120.0, 160.0, 192, 256, 1.0), (300.5, 20.25, 90.0, 90.0, 64, 64, 1.3)):
        patch, trans = OC.generate_patch_image_cv(img, cx, cy, bw, bh, pw, ph, False, scale, 0)
        inv = OC.invert_affine(trans)
        oy, ox = np.mgrid[0:ph, 0:pw].astype(np.float64)
        sx = inv[0, 0] * ox + inv[0, 1] * oy + inv[0, 2]
        sy = inv[1, 0] * ox + inv[1, 1] * oy + inv[1, 2]
        ref = np.stack([ndimage.map_coordinates(img[..., c].astype(np.float64), [sy, sx], order=1, mode='grid-constant', cval=0.0) for c in range(3)], -1)
        diff = np.abs(patch.astype(np.float64) - ref)
        assert diff.max() <= 4.0 and (diff > 2.0).mean() < 2e-3, (diff.max(), (diff > 2.0).mean())
        assert patch.shape == (ph, pw, 3)
        outside = (sx < -1) | (sx > 320) | (sy < -1) | (sy > 240)
        assert not patch[outside].any()                                                   # BORDER_CONSTANT: nothing outside the frame
        assert outside.any() or scale < 1.25


@pytest.mark.parametrize('droppath', [False, True])
def test_train_oracle_matches_reference_fixture(assets, state_dict, droppath):
    """oracle/train.py (training-mode forward + torch autograd) against tests/golden/whmr_train_b2.npz, which holds the imported
    reference's loss, per-parameter gradient (norm, sum) pairs, a few full gradients and the BatchNorm running statistics after the step
    (tests/golden/make_golden_train.py).  TRAIN.STAGE 2 (the configs/pymaf_config.yaml default); ``droppath``: the run with stochastic depth
    (vit.py:132-139,233, drop_path_rate 0.3) on the keep masks the fixture stores -- the reference consumed the same masks."""
    import os
    import numpy as np
    from oracle import synth
    from oracle import train as OT
    from conftest import GOLDEN
    fx = np.load(os.path.join(GOLDEN, 'whmr_train_b2.npz'))
    keys = [str(k) for k in fx['grad_keys']]
    inp = synth.make_inputs(2, 0)
    p = {k: (v.clone().requires_grad_(True) if k in keys else v) for k, v in state_dict.items()}
    stats, dp, gout = {}, [], []
    tag = 'stage2_droppath' if droppath else 'stage2'
    masks = torch.from_numpy(fx['drop_masks']) if droppath else None
    # global_output (whmr.py:630-654) with the camera rotation the reference's cam_model produced for the fixture's full image
    outs = OT.whmr_forward_train(p, assets, inp['x'], inp['center'], inp['scale'], inp['bbox_height'], inp['orig_shape'], inp['bbox_info'],
                                 stage=2, stats=stats, dp_out=dp, drop_masks=masks, drop_path_rate=0.3 if droppath else 0.0,
                                 global_out=gout, cam_rotmat=torch.from_numpy(fx['cam_rotmat']))
    loss, loss_dp, loss_g = OT.cotangent_loss(outs), OT.dp_cotangent_loss(dp[0]), OT.global_cotangent_loss(gout[0])
    (loss + loss_dp + loss_g).backward()
    assert abs(loss.item() - float(fx['loss_' + tag])) < 1e-5 and abs(loss_dp.item() - float(fx['loss_dp_' + tag])) < 1e-5
    assert abs(loss_g.item() - float(fx['loss_global_' + tag])) < 1e-5
    assert any(k.startswith('global_orient.') for k in keys)                         # the head's gradients are part of the pinned set (round 5)
    if droppath:
        assert abs(float(fx['loss_stage2_droppath']) - float(fx['loss_stage2'])) > 1e-4       # the masks really change the function
    ns = fx['grad_norm_sum_' + tag]
    for i, k in enumerate(keys):
        g = p[k].grad.double()
        if ns[i, 0] < 1e-8:
            assert g.norm().item() < 1e-6, k
            continue
        assert abs(g.norm().item() - ns[i, 0]) < 2e-4 * ns[i, 0], k
    for name in fx.files:
        if name.startswith('grad_%s/' % tag):
            k = name.split('/', 1)[1]
            ref = torch.from_numpy(fx[name])
            assert ((p[k].grad - ref).abs().max() / ref.abs().max()).item() < 5e-4, k
        if name.startswith('stat_%s/' % tag):
            k = name.split('/', 1)[1]
            ref = torch.from_numpy(fx[name])
            assert ((stats[k] - ref).abs().max() / ref.abs().max().clamp_min(1e-12)).item() < 2e-5, k


def test_raster_oracle_known_answers():
    """oracle/raster.py (pytorch3d restatement, parity unpinned): camera formula, single-triangle coverage with the strict-interior rule,
    barycentric texture interpolation (perspective-correct), depth order, shared edges"""
    from oracle import raster as OR
    # camera: a vertex on the optical axis lands on the image centre + the (px - W/2) offset the reference's K scaling introduces
    K = OR.camera_K(1000.0, (256, 256))
    assert np.allclose(K, (1000 * 256 / 224, 1000 * 256 / 224, 128 * 256 / 224, 128 * 256 / 224))
    cam = np.array([[0.9, 0.0, 0.0]])
    tz = 2 * 1000.0 / (256 * 0.9 + 1e-9)
    scr = OR.project(np.zeros((1, 1, 3)), cam, K, 1000.0, (256, 256), (128, 128))
    assert np.allclose(scr[0, 0], [64 + 0.5 * (K[2] - 128), 64 + 0.5 * (K[3] - 128), tz])
    assert np.allclose(OR.project(np.zeros((1, 1, 3)), cam, OR.camera_K(1000.0, (224, 224)), 1000.0, (224, 224), (56, 56))[0, 0, :2], [28, 28])
    # single fronto-parallel triangle built in screen space: choose vertices whose projection is known (orig 224 -> K unscaled, out 56)
    f, tz = 1000.0, 2 * 1000.0 / (224 * 1.0 + 1e-9)

    def world(u, v, z):                     # inverse of project() for cam = (1, 0, 0)
        zz = z + tz
        return [(u - 28) * 4 * zz / f, (v - 28) * 4 * zz / f, z]
    tri = np.array([[world(10.0, 10.0, 0.0), world(30.0, 10.0, 0.0), world(10.0, 30.0, 0.0)]])
    tex = np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]])
    out, fid = OR.rasterize(tri, [[0, 1, 2]], tex, np.array([[1.0, 0, 0]]), 1000.0, (224, 224), (56, 56))
    yy, xx = np.mgrid[0:56, 0:56] + 0.5
    expect = (xx > 10) & (yy > 10) & ((xx - 10) + (yy - 10) < 20)           # strictly inside: centres ON the hypotenuse x + y = 40 are not covered
    assert np.array_equal(fid[0] >= 0, expect)
    assert not (fid[0][(xx - 10) + (yy - 10) == 20] >= 0).any()
    # barycentric interpolation = the affine coordinates of the pixel centre (equal depths: perspective correction is the identity)
    j, i = 15, 12
    assert np.allclose(out[0, :, j, i], [1 - (i + 0.5 - 10) / 20 - (j + 0.5 - 10) / 20, (i + 0.5 - 10) / 20, (j + 0.5 - 10) / 20])
    assert np.allclose(out[0].sum(0)[fid[0] >= 0], 1.0) and not out[0][:, fid[0] < 0].any()
    # depth order: a nearer triangle over the same pixels wins regardless of the face order; equal depth -> the smaller face index
    near = np.array([world(10.0, 10.0, -0.5), world(30.0, 10.0, -0.5), world(10.0, 30.0, -0.5)])
    v2 = np.concatenate([tri[0], near])[None]
    tex2 = np.concatenate([tex, 0.5 * tex])
    for faces in ([[0, 1, 2], [3, 4, 5]], [[3, 4, 5], [0, 1, 2]]):
        o2, f2 = OR.rasterize(v2, faces, tex2, np.array([[1.0, 0, 0]]), 1000.0, (224, 224), (56, 56))
        nearest = faces.index([3, 4, 5])
        assert (f2[0][expect] == nearest).all() and np.allclose(o2[0].sum(0)[expect], 0.5)
    o3, f3 = OR.rasterize(np.concatenate([tri[0], tri[0]])[None], [[0, 1, 2], [3, 4, 5]], tex2, np.array([[1.0, 0, 0]]), 1000.0, (224, 224), (56, 56))
    assert (f3[0][expect] == 0).all()
    # two triangles sharing the diagonal of a square: pixel centres exactly on the shared edge belong to neither (pytorch3d blur 0)
    sq = np.array([[world(10.0, 10.0, 0.0), world(30.0, 10.0, 0.0), world(30.0, 30.0, 0.0), world(10.0, 30.0, 0.0)]])
    o4, f4 = OR.rasterize(sq, [[0, 1, 2], [0, 2, 3]], np.ones((4, 3)), np.array([[1.0, 0, 0]]), 1000.0, (224, 224), (56, 56))
    inside_sq = (xx > 10) & (xx < 30) & (yy > 10) & (yy < 30)
    assert np.array_equal(f4[0] >= 0, inside_sq & (xx != yy))
    # perspective-correct interpolation: a triangle tilted in depth -- the value at a pixel is the texture at the 3-D point the ray hits
    tilt = np.array([[world(10.0, 10.0, 0.0), world(30.0, 10.0, 3.0), world(10.0, 30.0, 0.0)]])
    o5, f5 = OR.rasterize(tilt, [[0, 1, 2]], tex, np.array([[1.0, 0, 0]]), 1000.0, (224, 224), (56, 56))
    z0, z1 = tz, 3.0 + tz
    w1 = (20.5 - 10) / 20                                                 # screen-space weight of vertex 1 at pixel centre (20.5, 10.5)
    w2 = 0.5 / 20
    w0 = 1 - w1 - w2
    b1 = (w1 / z1) / (w0 / z0 + w1 / z1 + w2 / z0)
    assert abs(o5[0, 1, 10, 20] - b1) < 1e-12 and b1 < w1
    # iuv_img2map: indicator maps and the 15 annotation groups
    uv = np.zeros((1, 3, 2, 2))
    uv[0, :, 0, 0] = [7 / 24, 0.3, 0.6]
    uv[0, :, 1, 1] = [24 / 24, 0.1, 0.2]
    U, V, I, A = OR.iuv_img2map(uv)
    assert I[0, 7, 0, 0] == 1 and I[0, 0, 0, 1] == 1 and I[0, 24, 1, 1] == 1 and I.sum() == 4
    assert U[0, 7, 0, 0] == 0.3 and V[0, 24, 1, 1] == 0.2 and A[0, 6, 0, 0] == 1 and A[0, 14, 1, 1] == 1 and A.sum() == 4
