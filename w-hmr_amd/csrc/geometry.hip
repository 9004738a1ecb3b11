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

// backward of whmr_mat_to_aa: d_in [n, 9] = (d aa / d R)^T d_out [n, 3] (geometry_dev.h rotmat_to_aa3_bwd)
__global__ void mat_to_aa_bwd_kernel(const float* __restrict__ in, const float* __restrict__ d_out, float* __restrict__ d_in, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float R[9], g[3], dR[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = in[9 * i + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) g[k] = d_out[3 * i + k];
    rotmat_to_aa3_bwd(R, g, dR);
#pragma unroll
    for (int k = 0; k < 9; ++k) d_in[9 * i + k] = dR[k];
}

extern "C" int whmr_mat_to_aa_bwd(const float* in, const float* d_out, float* d_in, int n, void* stream) {
    if (n <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(mat_to_aa_bwd_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, in, d_out, d_in, n);
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

// ---- forward glue: the O(batch) arithmetic around the kernels of WHMR.forward, each a single launch (was ~170 framework element-wise launches
// per forward: softmax / arange / cos / sin / stack / cat ... -- 10 % of the kernel time of a batch-64 forward and most of a batch-1 forward).

// Camera-calibration head post-processing (whmr.py:513-522 over utils/cam_utils.py:121-145 and pare's softargmax1d / batch_euler2matrix):
// logits [Bf, 3 D] = [vfov | pitch | roll] bins -> soft-argmax (expected bin under the softmax, mapped to [-1, 1]) -> angle in its range ->
// cam_rotmat = euler2matrix([pitch, 0, roll]), render_rotmat = euler2matrix([-pitch, 0, roll]) (q = qx qy qz -> R).  One workgroup of D threads
// per frame; Bf == 1 broadcasts the frame's matrices to all B crops (demo/tester.py:161 replicates the frame instead).
__device__ __forceinline__ void euler_xz_to_mat(float px, float rz, float* R) {
    // pare / DECA batch_euler2matrix with y = 0: quaternion (w, x, y, z) of qx . qy . qz, normalised, then the standard q -> R
    const float cx = cosf(0.5f * px), sx = sinf(0.5f * px), cz = cosf(0.5f * rz), sz = sinf(0.5f * rz);
    float w = cx * cz, x = cz * sx, y = -sx * sz, z = cx * sz;          // cy = 1, sy = 0
    const float n = sqrtf(w * w + x * x + y * y + z * z);
    w /= n; x /= n; y /= n; z /= n;
    const float w2 = w * w, x2 = x * x, y2 = y * y, z2 = z * z, wx = w * x, wy = w * y, wz = w * z, xy = x * y, xz = x * z, yz = y * z;
    R[0] = w2 + x2 - y2 - z2; R[1] = 2 * xy - 2 * wz; R[2] = 2 * wy + 2 * xz;
    R[3] = 2 * wz + 2 * xy; R[4] = w2 - x2 + y2 - z2; R[5] = 2 * yz - 2 * wx;
    R[6] = 2 * xz - 2 * wy; R[7] = 2 * wx + 2 * yz; R[8] = w2 - x2 - y2 + z2;
}

__global__ __launch_bounds__(256) void cam_head_kernel(const float* __restrict__ logits, int ld, int D, float p_lo, float p_hi, float r_lo, float r_hi,
                                                       int Bf, int B, float* __restrict__ cam_rotmat, float* __restrict__ render_rotmat) {
    __shared__ float red[2][3][4];
    __shared__ float sR[18];
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float v[2], m[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {                                        // a = 0 pitch bins (columns D..2D), 1 roll bins (2D..3D)
        v[a] = tid < D ? logits[(size_t)f * ld + (a + 1) * D + tid] : -INFINITY;
        m[a] = wave_max(v[a]);
        if (lane == 0) red[a][0][wave] = m[a];
    }
    __syncthreads();
    float e[2], s[2], t[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const float mx = fmaxf(fmaxf(red[a][0][0], red[a][0][1]), fmaxf(red[a][0][2], red[a][0][3]));
        e[a] = tid < D ? expf(v[a] - mx) : 0.f;
        s[a] = wave_sum(e[a]);
        t[a] = wave_sum(e[a] * (float)tid);
        if (lane == 0) { red[a][1][wave] = s[a]; red[a][2][wave] = t[a]; }
    }
    __syncthreads();
    if (tid == 0) {
        float ang[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const float sum = (red[a][1][0] + red[a][1][1]) + (red[a][1][2] + red[a][1][3]);
            const float idx = ((red[a][2][0] + red[a][2][1]) + (red[a][2][2] + red[a][2][3])) / sum;   // E[bin] under softmax
            const float soft = idx / (float)(D - 1) * 2.f - 1.f;
            const float lo = a == 0 ? p_lo : r_lo, hi = a == 0 ? p_hi : r_hi;
            ang[a] = (hi - lo) * ((soft + 1.f) / 2.f) + lo;
        }
        euler_xz_to_mat(ang[0], ang[1], sR);
        euler_xz_to_mat(-ang[0], ang[1], sR + 9);
    }
    __syncthreads();
    const int b0 = Bf == 1 ? 0 : f, nb = Bf == 1 ? B : 1;
    for (int i = tid; i < nb * 9; i += 256) {
        cam_rotmat[(size_t)b0 * 9 + i] = sR[i % 9];
        render_rotmat[(size_t)b0 * 9 + i] = sR[9 + i % 9];
    }
}

extern "C" int whmr_cam_head(const float* logits, int ld, int D, float pitch_lo, float pitch_hi, float roll_lo, float roll_hi, int Bf, int B,
                             float* cam_rotmat, float* render_rotmat, void* stream) {
    if (Bf <= 0 || B <= 0 || D <= 0 || D > 256 || ld < 3 * D || (Bf != 1 && Bf != B)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(cam_head_kernel, dim3(Bf), dim3(256), 0, (hipStream_t)stream, logits, ld, D, pitch_lo, pitch_hi, roll_lo, roll_hi, Bf, B,
                       cam_rotmat, render_rotmat);
    WHMR_CHECK_LAUNCH();
    return 0;
}

// Input state of the global-orientation head (whmr.py:295-297): xc[b, F .. F+15) = [rot6d(cam_rotmat[b]) (first two columns, row-major: geometry.py:275-286)
// | local_orient[b] (the root joint's 3x3, the first 9 floats of the stage's rotmat row)] in one launch.
__global__ void orient_state_kernel(const float* __restrict__ cam_rotmat, const float* __restrict__ rotmat, long ld_rot, float* __restrict__ xc, long ld,
                                    int F, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 15) return;
    const int b = i / 15, e = i % 15;
    float v;
    if (e < 6) v = cam_rotmat[(size_t)b * 9 + (e >> 1) * 3 + (e & 1)];
    else v = rotmat[(size_t)b * ld_rot + e - 6];
    xc[(size_t)b * ld + F + e] = v;
}

extern "C" int whmr_orient_state(const float* cam_rotmat, const float* rotmat, long ld_rot, float* xc, long ld, int F, int B, void* stream) {
    if (B <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(orient_state_kernel, dim3((B * 15 + 255) / 256), dim3(256), 0, (hipStream_t)stream, cam_rotmat, rotmat, ld_rot, xc, ld, F, B);
    WHMR_CHECK_LAUNCH();
    return 0;
}

// Tail of the global-orientation head (whmr.py:301-305,630-640): r [B, 9] (the head's residual output) -> unbiased Gram-Schmidt -> g_rot; its
// angle-axis; global_pose [B, 72] = [aa(g_rot) | pose_aa[:, 3:]], global_rotmat [B, 24, 9] = [g_rot | rotmat[:, 1:]] in one launch.
__global__ __launch_bounds__(256) void orient_tail_kernel(const float* __restrict__ r, const float* __restrict__ pose_aa, const float* __restrict__ rotmat,
                                                          float* __restrict__ g_pose, float* __restrict__ g_rotmat) {
    __shared__ float sO[9], sA[3];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        float m[9], o[9], a3[3];
        for (int k = 0; k < 9; ++k) m[k] = r[(size_t)b * 9 + k];
        gram_schmidt9(m, o);
        rotmat_to_aa3(o, a3);
        for (int k = 0; k < 9; ++k) sO[k] = o[k];
        for (int k = 0; k < 3; ++k) sA[k] = a3[k];
    }
    __syncthreads();
    if (tid < 72) g_pose[(size_t)b * 72 + tid] = tid < 3 ? sA[tid] : pose_aa[(size_t)b * 72 + tid];
    if (tid < 216) g_rotmat[(size_t)b * 216 + tid] = tid < 9 ? sO[tid] : rotmat[(size_t)b * 216 + tid];
}

extern "C" int whmr_orient_tail(const float* r, const float* pose_aa, const float* rotmat, float* g_pose, float* g_rotmat, int B, void* stream) {
    if (B <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(orient_tail_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, r, pose_aa, rotmat, g_pose, g_rotmat);
    WHMR_CHECK_LAUNCH();
    return 0;
}
