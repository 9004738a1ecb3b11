// Small rotation / projection kernels behind whmr_amd.utils.geometry (same function names as utils/geometry.py).
// One thread per rotation / point; these are latency-trivial and exist so that no geometry op leaves the device
// or falls back to a framework op.  Reference lines are cited in geometry_dev.h and per kernel below.
#include "geometry_dev.h"

// mode 0: rot6d -> R (geometry.py:243-257), in stride 6
// mode 1: unbiased Gram-Schmidt (geometry.py:260-272), in stride 9
// mode 2: Rodrigues aa -> R (geometry.py:14-27), in stride 3
__global__ void rot_to_mat_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int mode) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float o[9];
    if (mode == 0) { float x[6]; for (int k = 0; k < 6; ++k) x[k] = in[6 * i + k]; rot6d_to_rotmat9(x, o); }
    else if (mode == 1) { float m[9]; for (int k = 0; k < 9; ++k) m[k] = in[9 * i + k]; gram_schmidt9(m, o); }
    else { float t[3] = {in[3 * i], in[3 * i + 1], in[3 * i + 2]}; rodrigues9(t, o); }
    for (int k = 0; k < 9; ++k) out[9 * i + k] = o[k];
}

// rotation_matrix_to_angle_axis (geometry.py:54-83)
__global__ void mat_to_aa_kernel(const float* __restrict__ in, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float m[9], o[3];
    for (int k = 0; k < 9; ++k) m[k] = in[9 * i + k];
    rotmat_to_aa3(m, o);
    out[3 * i] = o[0]; out[3 * i + 1] = o[1]; out[3 * i + 2] = o[2];
}

// perspective_projection (geometry.py:310-341): p' = R p + t; p'/z; K p'.  rot may be null (identity) or have batch 1
// (rot_bstride = 0, whmr.py:158-160); focal is per batch (focal_bstride = 1) or a single scalar (0).
// post_scale / post_shift: out = proj * post_scale[b,:] + post_shift  -- covers projection()'s /(IMG_RES/2)
// (geometry.py:303-304) and the regressor's kp/center - 1 (whmr.py:173).
__global__ void perspective_kernel(const float* __restrict__ pts, const float* __restrict__ rot, int rot_bstride,
                                   const float* __restrict__ trans, const float* __restrict__ focal, int focal_bstride,
                                   const float* __restrict__ center, const float* __restrict__ post_div, float post_shift,
                                   float* __restrict__ out, int B, int P) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * P) return;
    const int b = i / P;
    float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
    if (rot) {
        const float* R = rot + (size_t)b * rot_bstride;
        const float nx = R[0] * x + R[1] * y + R[2] * z, ny = R[3] * x + R[4] * y + R[5] * z, nz = R[6] * x + R[7] * y + R[8] * z;
        x = nx; y = ny; z = nz;
    }
    x += trans[3 * b]; y += trans[3 * b + 1]; z += trans[3 * b + 2];
    const float f = focal[b * focal_bstride];
    const float cx = center ? center[2 * b] : 0.f, cy = center ? center[2 * b + 1] : 0.f;
    float u = f * (x / z) + cx, v = f * (y / z) + cy;
    if (post_div) { u = u / post_div[2 * b] + post_shift; v = v / post_div[2 * b + 1] + post_shift; }
    out[2 * i] = u; out[2 * i + 1] = v;
}

// projection (geometry.py:289-307): weak-perspective camera (s, tx, ty) -> t = [tx, ty, 2*1000/(256*s + 1e-9)], f = 1000,
// centre 0, then / (256/2).  img_res / focal are passed so cfg.IMG_RES / constants.FOCAL_LENGTH stay configurable.
__global__ void weak_projection_kernel(const float* __restrict__ pts, const float* __restrict__ cam, float* __restrict__ out,
                                       int B, int P, float focal, float res_w, float res_h) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * P) return;
    const int b = i / P;
    const float s = cam[3 * b], tx = cam[3 * b + 1], ty = cam[3 * b + 2];
    const float tz = 2 * focal / (res_h * s + 1e-9f);
    const float x = pts[3 * i] + tx, y = pts[3 * i + 1] + ty, z = pts[3 * i + 2] + tz;
    out[2 * i] = (focal * (x / z)) / (res_w / 2.f);
    out[2 * i + 1] = (focal * (y / z)) / (res_h / 2.f);
}

extern "C" int whmr_rot_to_mat(const float* in, float* out, int n, int mode, void* stream) {
    if (n <= 0 || mode < 0 || mode > 2) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(rot_to_mat_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, in, out, n, mode);
    WHMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int whmr_mat_to_aa(const float* in, float* out, int n, void* stream) {
    if (n <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(mat_to_aa_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, in, out, n);
    WHMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int whmr_perspective(const float* pts, const float* rot, int rot_bstride, const float* trans, const float* focal,
                                int focal_bstride, const float* center, const float* post_div, float post_shift, float* out,
                                int B, int P, void* stream) {
    if (B <= 0 || P <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(perspective_kernel, dim3((B * P + 63) / 64), dim3(64), 0, (hipStream_t)stream, pts, rot, rot_bstride,
                       trans, focal, focal_bstride, center, post_div, post_shift, out, B, P);
    WHMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int whmr_weak_projection(const float* pts, const float* cam, float* out, int B, int P, float focal, float res_w,
                                    float res_h, void* stream) {
    if (B <= 0 || P <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(weak_projection_kernel, dim3((B * P + 63) / 64), dim3(64), 0, (hipStream_t)stream, pts, cam, out, B, P,
                       focal, res_w, res_h);
    WHMR_CHECK_LAUNCH();
    return 0;
}
