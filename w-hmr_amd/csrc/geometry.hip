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

// estimate_translation (utils/geometry.py:344-408; called per batch at core/trainer.py:435 with a .cpu() round trip and a numpy
// loop over the samples -- SURVEY 8f N3): weighted least squares for the camera translation that best re-projects the 3-D joints
// S onto the 2-D joints, one thread per sample, float64 like numpy: rows (F w, 0, (Ox - x) w | ((x - Ox) Z - F X) w) and
// (0, F w, (Oy - y) w | ((y - Oy) Z - F Y) w) with w = sqrtf(conf); A = Q^T Q, b = Q^T c, 3x3 solve with partial pivoting.
__global__ __launch_bounds__(64) void estimate_translation_kernel(const float* __restrict__ S, const float* __restrict__ j2d, int B, int J, int j0,
                                                                  int nj, double F, double Ox, double Oy, float* __restrict__ out) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    double A[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};           // augmented [A | b]
    for (int j = j0; j < j0 + nj; ++j) {
        const float* s = S + ((size_t)b * J + j) * 3;
        const float* p = j2d + ((size_t)b * J + j) * 3;
        const double w = (double)sqrtf(p[2]);
        const double x = p[0], y = p[1], X = s[0], Y = s[1], Z = s[2];
        const double q0 = F * w, q2x = (Ox - x) * w, q2y = (Oy - y) * w;
        const double cx = ((x - Ox) * Z - F * X) * w, cy = ((y - Oy) * Z - F * Y) * w;
        A[0][0] += q0 * q0; A[0][2] += q0 * q2x; A[0][3] += q0 * cx;
        A[1][1] += q0 * q0; A[1][2] += q0 * q2y; A[1][3] += q0 * cy;
        A[2][2] += q2x * q2x + q2y * q2y; A[2][3] += q2x * cx + q2y * cy;
    }
    A[2][0] = A[0][2]; A[2][1] = A[1][2];
    // Gaussian elimination with partial pivoting (what LAPACK's dgesv behind np.linalg.solve does)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int piv = c;
        for (int r = c + 1; r < 3; ++r)
            if (fabs(A[r][c]) > fabs(A[piv][c])) piv = r;
        if (piv != c)
            for (int k = 0; k < 4; ++k) { const double t = A[c][k]; A[c][k] = A[piv][k]; A[piv][k] = t; }
        for (int r = c + 1; r < 3; ++r) {
            const double f = A[r][c] / A[c][c];
            for (int k = c; k < 4; ++k) A[r][k] -= f * A[c][k];
        }
    }
    double t[3];
    for (int r = 2; r >= 0; --r) {
        double v = A[r][3];
        for (int k = r + 1; k < 3; ++k) v -= A[r][k] * t[k];
        t[r] = v / A[r][r];
    }
    out[b * 3 + 0] = (float)t[0]; out[b * 3 + 1] = (float)t[1]; out[b * 3 + 2] = (float)t[2];
}

extern "C" int whmr_estimate_translation(const float* S, const float* joints_2d, int B, int J, int j0, int nj, float focal, float img_w,
                                         float img_h, float* out, void* stream) {
    if (B <= 0 || nj <= 0 || j0 < 0 || j0 + nj > J) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(estimate_translation_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, S, joints_2d, B, J, j0, nj,
                       (double)focal, (double)img_w / 2.0, (double)img_h / 2.0, out);
    WHMR_CHECK_LAUNCH();
    return 0;
}
