// Device-side rotation helpers shared by geometry.hip and smpl_lbs.hip.
// Each function follows the reference arithmetic of utils/geometry.py (lines cited per function).
#pragma once
#include "common.h"

struct v3 { float x, y, z; };

__device__ __forceinline__ v3 cross3(v3 a, v3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ float dot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// F.normalize(p=2, eps=1e-12): v / max(||v||, eps)
__device__ __forceinline__ v3 normalize3(v3 a) {
    const float n = fmaxf(sqrtf(dot3(a, a)), 1e-12f);
    return {a.x / n, a.y / n, a.z / n};
}

// utils/geometry.py:260-272.  m = row-major 3x3 (m[3*i+j]); columns are t1,t2,t3; output columns r1,r2,r3.
__device__ __forceinline__ void gram_schmidt9(const float* m, float* o) {
    const v3 t1 = {m[0], m[3], m[6]}, t2 = {m[1], m[4], m[7]}, t3 = {m[2], m[5], m[8]};
    v3 c = cross3(t2, t3);
    const v3 r1 = normalize3({(c.x + t1.x) / 2.f, (c.y + t1.y) / 2.f, (c.z + t1.z) / 2.f});
    c = cross3(t3, r1);
    const v3 r2_ = {(c.x + t2.x) / 2.f, (c.y + t2.y) / 2.f, (c.z + t2.z) / 2.f};
    const float d = dot3(r2_, r1);
    const v3 r2 = normalize3({r2_.x - d * r1.x, r2_.y - d * r1.y, r2_.z - d * r1.z});
    const v3 r3 = cross3(r1, r2);
    o[0] = r1.x; o[1] = r2.x; o[2] = r3.x;
    o[3] = r1.y; o[4] = r2.y; o[5] = r3.y;
    o[6] = r1.z; o[7] = r2.z; o[8] = r3.z;
}

// utils/geometry.py:243-257.  x = 6 values viewed (3,2): a1 = x[0,2,4], a2 = x[1,3,5].
__device__ __forceinline__ void rot6d_to_rotmat9(const float* x, float* o) {
    const v3 a1 = {x[0], x[2], x[4]}, a2 = {x[1], x[3], x[5]};
    const v3 b1 = normalize3(a1);
    const float d = dot3(b1, a2);
    const v3 b2 = normalize3({a2.x - d * b1.x, a2.y - d * b1.y, a2.z - d * b1.z});
    const v3 b3 = cross3(b1, b2);
    o[0] = b1.x; o[1] = b2.x; o[2] = b3.x;
    o[3] = b1.y; o[4] = b2.y; o[5] = b3.y;
    o[6] = b1.z; o[7] = b2.z; o[8] = b3.z;
}

// utils/geometry.py:30-51 (w,x,y,z), re-normalised.
__device__ __forceinline__ void quat_to_rotmat9(float w, float x, float y, float z, float* o) {
    const float n = sqrtf(w * w + x * x + y * y + z * z);
    w /= n; x /= n; y /= n; z /= n;
    const float w2 = w * w, x2 = x * x, y2 = y * y, z2 = z * z;
    const float wx = w * x, wy = w * y, wz = w * z, xy = x * y, xz = x * z, yz = y * z;
    o[0] = w2 + x2 - y2 - z2; o[1] = 2 * xy - 2 * wz;    o[2] = 2 * wy + 2 * xz;
    o[3] = 2 * wz + 2 * xy;    o[4] = w2 - x2 + y2 - z2; o[5] = 2 * yz - 2 * wx;
    o[6] = 2 * xz - 2 * wy;    o[7] = 2 * wx + 2 * yz;    o[8] = w2 - x2 - y2 + z2;
}

// utils/geometry.py:14-27: angle = ||theta + 1e-8||, axis = theta / angle (unshifted theta).
__device__ __forceinline__ void rodrigues9(const float* t, float* o) {
    const float a0 = t[0] + 1e-8f, a1 = t[1] + 1e-8f, a2 = t[2] + 1e-8f;
    const float angle = sqrtf(a0 * a0 + a1 * a1 + a2 * a2);
    const float nx = t[0] / angle, ny = t[1] / angle, nz = t[2] / angle;
    const float h = angle * 0.5f, c = cosf(h), s = sinf(h);
    quat_to_rotmat9(c, s * nx, s * ny, s * nz, o);
}

// utils/geometry.py:54-83 + :160-240 + :86-136: R -> quaternion (4 masked cases, eps 1e-6 on R[2][2]) -> angle-axis, NaN -> 0.
// R row-major; the reference indexes the transpose rmat_t[a][b] = R[b][a].
__device__ __forceinline__ void rotmat_to_aa3(const float* R, float* o) {
#define RT(a, b) R[3 * (b) + (a)]
    const float d0 = RT(0, 0), d1 = RT(1, 1), d2 = RT(2, 2);
    const bool m2 = d2 < 1e-6f, m01 = d0 > d1, m0n1 = d0 < -d1;
    float q0, q1, q2, q3, t;
    if (m2 && m01) {
        t = 1 + d0 - d1 - d2;
        q0 = RT(1, 2) - RT(2, 1); q1 = t; q2 = RT(0, 1) + RT(1, 0); q3 = RT(2, 0) + RT(0, 2);
    } else if (m2 && !m01) {
        t = 1 - d0 + d1 - d2;
        q0 = RT(2, 0) - RT(0, 2); q1 = RT(0, 1) + RT(1, 0); q2 = t; q3 = RT(1, 2) + RT(2, 1);
    } else if (!m2 && m0n1) {
        t = 1 - d0 - d1 + d2;
        q0 = RT(0, 1) - RT(1, 0); q1 = RT(2, 0) + RT(0, 2); q2 = RT(1, 2) + RT(2, 1); q3 = t;
    } else {
        t = 1 + d0 + d1 + d2;
        q0 = t; q1 = RT(1, 2) - RT(2, 1); q2 = RT(2, 0) - RT(0, 2); q3 = RT(0, 1) - RT(1, 0);
    }
#undef RT
    const float st = sqrtf(t);
    q0 = q0 / st * 0.5f; q1 = q1 / st * 0.5f; q2 = q2 / st * 0.5f; q3 = q3 / st * 0.5f;
    const float s2 = q1 * q1 + q2 * q2 + q3 * q3;
    const float s = sqrtf(s2);
    const float two_theta = 2.0f * (q0 < 0.0f ? atan2f(-s, -q0) : atan2f(s, q0));
    const float k = s2 > 0.0f ? two_theta / s : 2.0f;
    float a0 = q1 * k, a1 = q2 * k, a2 = q3 * k;
    o[0] = (a0 != a0) ? 0.f : a0;
    o[1] = (a1 != a1) ? 0.f : a1;
    o[2] = (a2 != a2) ? 0.f : a2;
}

// Reverse mode of rotmat_to_aa3 (what torch autograd computes through utils/geometry.py:54-83,86-136,160-240 in the reference's training graph:
// global_pose / theta carry rotation_matrix_to_angle_axis of rotmats that have a graph, whmr.py:174,632-633): dR [9] = (d aa / d R)^T d_aa.
// Same branch selection as the forward (the reference multiplies the four candidate quaternions by 0 / 1 masks: only the selected one has a
// gradient); the forward's `aa[isnan(aa)] = 0` is an in-place masked write, so a NaN component passes no gradient.  One deliberate difference:
// at sin^2(theta) == 0 exactly (R = I bit for bit) torch's where(sin2 > 0, two_theta / sin, 2) still back-propagates through the unselected
// branch's sqrt(0) and yields NaN; here the selected branch k = 2 is differentiated (d q_i = 2 d aa_i), a finite value.
__device__ __forceinline__ void rotmat_to_aa3_bwd(const float* R, const float* d_o, float* dR) {
#define RT(a, b) R[3 * (b) + (a)]
#define DRT(a, b) dR[3 * (b) + (a)]
    const float d0 = RT(0, 0), d1 = RT(1, 1), d2 = RT(2, 2);
    const bool m2 = d2 < 1e-6f, m01 = d0 > d1, m0n1 = d0 < -d1;
    const int br = (m2 && m01) ? 0 : (m2 && !m01) ? 1 : (!m2 && m0n1) ? 2 : 3;
    float n0, n1, n2, n3, t;
    if (br == 0) { t = 1 + d0 - d1 - d2; n0 = RT(1, 2) - RT(2, 1); n1 = t; n2 = RT(0, 1) + RT(1, 0); n3 = RT(2, 0) + RT(0, 2); }
    else if (br == 1) { t = 1 - d0 + d1 - d2; n0 = RT(2, 0) - RT(0, 2); n1 = RT(0, 1) + RT(1, 0); n2 = t; n3 = RT(1, 2) + RT(2, 1); }
    else if (br == 2) { t = 1 - d0 - d1 + d2; n0 = RT(0, 1) - RT(1, 0); n1 = RT(2, 0) + RT(0, 2); n2 = RT(1, 2) + RT(2, 1); n3 = t; }
    else { t = 1 + d0 + d1 + d2; n0 = t; n1 = RT(1, 2) - RT(2, 1); n2 = RT(2, 0) - RT(0, 2); n3 = RT(0, 1) - RT(1, 0); }
    const float st = sqrtf(t);
    const float q0 = n0 / st * 0.5f, q1 = n1 / st * 0.5f, q2 = n2 / st * 0.5f, q3 = n3 / st * 0.5f;
    const float s2 = q1 * q1 + q2 * q2 + q3 * q3;
    const float sn = sqrtf(s2);
    const float two_theta = 2.0f * (q0 < 0.0f ? atan2f(-sn, -q0) : atan2f(sn, q0));
    const float k = s2 > 0.0f ? two_theta / sn : 2.0f;
    // aa_i = q_i k, NaN components masked
    const float a0 = q1 * k, a1 = q2 * k, a2 = q3 * k;
    const float g0 = (a0 != a0) ? 0.f : d_o[0], g1 = (a1 != a1) ? 0.f : d_o[1], g2 = (a2 != a2) ? 0.f : d_o[2];
    float dq0 = 0.f, dq1 = g0 * k, dq2 = g1 * k, dq3 = g2 * k;
    if (s2 > 0.0f) {
        const float dk = g0 * q1 + g1 * q2 + g2 * q3;
        const float dtt = dk / sn;                                   // d two_theta
        float dsn = -dk * two_theta / s2;
        // two_theta = 2 atan2(+-sn, +-q0): d atan2(y, x) = (x dy - y dx) / (x^2 + y^2), the same expression for both sign choices
        const float den = s2 + q0 * q0;
        dsn += 2.0f * dtt * q0 / den;
        dq0 = -2.0f * dtt * sn / den;
        const float ds2 = dsn / (2.0f * sn);
        dq1 += 2.0f * q1 * ds2; dq2 += 2.0f * q2 * ds2; dq3 += 2.0f * q3 * ds2;
    }
    // q_i = 0.5 n_i / sqrt(t)
    const float dn0 = 0.5f * dq0 / st, dn1 = 0.5f * dq1 / st, dn2 = 0.5f * dq2 / st, dn3 = 0.5f * dq3 / st;
    const float dst = -(dq0 * q0 + dq1 * q1 + dq2 * q2 + dq3 * q3) / st;
    float dt = dst / (2.0f * st);
#pragma unroll
    for (int i = 0; i < 9; ++i) dR[i] = 0.f;
    if (br == 0) {
        dt += dn1;
        DRT(1, 2) += dn0; DRT(2, 1) -= dn0; DRT(0, 1) += dn2; DRT(1, 0) += dn2; DRT(2, 0) += dn3; DRT(0, 2) += dn3;
        DRT(0, 0) += dt; DRT(1, 1) -= dt; DRT(2, 2) -= dt;
    } else if (br == 1) {
        dt += dn2;
        DRT(2, 0) += dn0; DRT(0, 2) -= dn0; DRT(0, 1) += dn1; DRT(1, 0) += dn1; DRT(1, 2) += dn3; DRT(2, 1) += dn3;
        DRT(0, 0) -= dt; DRT(1, 1) += dt; DRT(2, 2) -= dt;
    } else if (br == 2) {
        dt += dn3;
        DRT(0, 1) += dn0; DRT(1, 0) -= dn0; DRT(2, 0) += dn1; DRT(0, 2) += dn1; DRT(1, 2) += dn2; DRT(2, 1) += dn2;
        DRT(0, 0) -= dt; DRT(1, 1) -= dt; DRT(2, 2) += dt;
    } else {
        dt += dn0;
        DRT(1, 2) += dn1; DRT(2, 1) -= dn1; DRT(2, 0) += dn2; DRT(0, 2) -= dn2; DRT(0, 1) += dn3; DRT(1, 0) -= dn3;
        DRT(0, 0) += dt; DRT(1, 1) += dt; DRT(2, 2) += dt;
    }
#undef RT
#undef DRT
}
