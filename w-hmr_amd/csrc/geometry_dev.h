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
