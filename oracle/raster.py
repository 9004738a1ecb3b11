"""Oracle restatement of the IUV ground-truth rasteriser.  TEST INFRASTRUCTURE ONLY.

What it follows: utils/renderer.py:296-446 of the reference (IUV_Renderer: camera matrices :362-411, verts2iuvimg :412-433), which delegates the
actual rasterisation to pytorch3d (MeshRasterizer faces_per_pixel=1, blur_radius=0; HardFlatShader over TexturesVertex with AmbientLights).
pytorch3d is a third-party dependency ABSENT from /root/reference (environment.yml pins pytorch3d 0.7.x): its published naive rasterisation
algorithm (pytorch3d/csrc/rasterize_meshes: pixel centres, strict-interior test at blur 0, kEpsilon area cut, perspective-correct barycentrics,
nearest face) and its screen <-> NDC conventions are restated here in float64 numpy -- PARITY UNPINNED against the real library; pinned by
the analytic known answers of tests/test_oracle_cpu.py (single triangle, depth order, shared edges, camera formula).
"""
import numpy as np


def camera_K(focal, orig_size):
    """renderer.py:362-380: K = [[f, 0, w/2], [0, f, h/2]], all four entries scaled by orig / 224 when orig_size[0] != 224."""
    fx = fy = float(focal)
    px, py = orig_size[1] / 2.0, orig_size[0] / 2.0
    if orig_size[0] != 224:
        sw, sh = orig_size[1] / 224.0, orig_size[0] / 224.0
        fx, px, fy, py = fx * sw, px * sw, fy * sh, py * sh
    return fx, fy, px, py


def project(verts, cam, K, focal, orig_size, out_size):
    """renderer.py:396-411 + pytorch3d PerspectiveCameras(in_ndc=False, R = diag(-1,-1,1), T = (-cam1, -cam2, 2 f / (orig_h cam0 + 1e-9))):
    X_view = (-(x + cam1), -(y + cam2), z + tz); pytorch3d's screen->NDC flip (+X left, +Y up) cancels the two minus signs, so in pixels of the
    H x W output image  u = W/2 + (W/orig_w) (fx (x + cam1) / z' + px - orig_w/2),  v likewise."""
    fx, fy, px, py = K
    H, W = out_size
    tz = 2.0 * focal / (orig_size[0] * cam[:, 0] + 1e-9)
    z = verts[..., 2] + tz[:, None]
    u = 0.5 * W + (W / orig_size[1]) * (fx * (verts[..., 0] + cam[:, 1:2]) / z + px - 0.5 * orig_size[1])
    v = 0.5 * H + (H / orig_size[0]) * (fy * (verts[..., 1] + cam[:, 2:3]) / z + py - 0.5 * orig_size[0])
    return np.stack([u, v, z], -1)


def rasterize(verts, faces, tex, cam, focal=1000.0, orig_size=(224, 224), out_size=(56, 56), vmap=None):
    """-> (iuv [B, 3, H, W], face index [B, H, W] (-1 = background)), float64."""
    verts, cam, tex = np.asarray(verts, np.float64), np.asarray(cam, np.float64), np.asarray(tex, np.float64)
    faces = np.asarray(faces, np.int64)
    if vmap is not None:
        verts = verts[:, np.asarray(vmap, np.int64)]
    B = verts.shape[0]
    H, W = out_size
    scr = project(verts, cam, camera_K(focal, orig_size), focal, orig_size, out_size)
    out = np.zeros((B, 3, H, W))
    fid = -np.ones((B, H, W), np.int64)
    zbuf = np.full((B, H, W), np.inf)
    eps = 1e-8 * 0.25 * H * W
    for b in range(B):
        for f, (i0, i1, i2) in enumerate(faces):
            v0, v1, v2 = scr[b, i0], scr[b, i1], scr[b, i2]
            xs, ys = (v0[0], v1[0], v2[0]), (v0[1], v1[1], v2[1])
            if not (max(xs) >= 0 and max(ys) >= 0 and min(xs) <= W and min(ys) <= H):
                continue
            x0, x1 = max(int(np.floor(min(xs) - 0.5)), 0), min(int(np.ceil(max(xs) - 0.5)), W - 1)
            y0, y1 = max(int(np.floor(min(ys) - 0.5)), 0), min(int(np.ceil(max(ys) - 0.5)), H - 1)
            if x1 < x0 or y1 < y0:
                continue
            area = (v2[0] - v0[0]) * (v1[1] - v0[1]) - (v2[1] - v0[1]) * (v1[0] - v0[0])
            if abs(area) <= eps:
                continue
            px, py = np.meshgrid(np.arange(x0, x1 + 1) + 0.5, np.arange(y0, y1 + 1) + 0.5)

            def edge(a, c):
                return (px - a[0]) * (c[1] - a[1]) - (py - a[1]) * (c[0] - a[0])
            w0, w1, w2 = edge(v1, v2) / area, edge(v2, v0) / area, edge(v0, v1) / area
            inside = (w0 > 0) & (w1 > 0) & (w2 > 0)
            if not inside.any():
                continue
            q0, q1, q2 = w0 / v0[2], w1 / v1[2], w2 / v2[2]
            den = q0 + q1 + q2
            with np.errstate(divide='ignore', invalid='ignore'):
                b0, b1, b2 = q0 / den, q1 / den, q2 / den
                pz = b0 * v0[2] + b1 * v1[2] + b2 * v2[2]
            zb = zbuf[b, y0:y1 + 1, x0:x1 + 1]
            take = inside & (pz >= 0) & ((pz < zb) | ((pz == zb) & (f < fid[b, y0:y1 + 1, x0:x1 + 1])))
            if not take.any():
                continue
            zb[take] = pz[take]
            fid[b, y0:y1 + 1, x0:x1 + 1][take] = f
            for c in range(3):
                val = b0 * tex[i0, c] + b1 * tex[i1, c] + b2 * tex[i2, c]
                out[b, c, y0:y1 + 1, x0:x1 + 1][take] = val[take]
    return out, fid


def iuv_img2map(uv):
    """utils/iuvmap.py:67-110 (uv_rois=None): part indicator maps, U / V masked per part, 15 annotation groups"""
    index2mask = ((0,), (1, 2), (3,), (4,), (5,), (6,), (7, 9), (8, 10), (11, 13), (12, 14), (15, 17), (16, 18), (19, 21), (20, 22), (23, 24))
    part = np.round(np.asarray(uv)[:, 0] * 24)
    idx = np.stack([(part == i).astype(np.float64) for i in range(25)], 1)
    return idx * uv[:, 1:2], idx * uv[:, 2:3], idx, np.stack([sum(idx[:, j] for j in g) for g in index2mask], 1)
