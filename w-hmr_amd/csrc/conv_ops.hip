// Small NHWC helpers that let the camera-calibration ResNet-50 (models/cam_model.py:24-81; PARE resnet50 trunk) run on the
// bf16 implicit-GEMM kernel: stem im2col, max-pool, global average pool.
#include "common.h"

// NCHW fp32 image (element strides) -> cols [B*OH*OW, Kpad] bf16 for the 7x7 stride-2 stem (Cin = 3).  Column order
// k = (ci*KH + ky)*8 + kx with kx padded 7 -> 8, so every (ci, ky) segment is one aligned 16-B store and a pixel's row is
// written by 24 neighbouring threads (Kpad = 192 = 24 segments; segments >= Cin*KH and kx = 7 are zero).  The matching
// weight layout is built in models/cam_model.py (_prepare).
__global__ __launch_bounds__(256) void conv_im2col_kernel(const float* __restrict__ x, bf16_t* __restrict__ cols, int B, int Cin,
                                                          int H, int W, int KH, int KW, int S, int pad, int OH, int OW, int Kpad,
                                                          long sb, long sc, long sh, long sw) {
    const int segs = Kpad >> 3;
    const long total = (long)B * OH * OW * segs;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int seg = (int)(idx % segs);
    const long m = idx / segs;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (seg < Cin * KH) {
        const int ci = seg / KH, ky = seg - ci * KH;
        const int ox = (int)(m % OW), oy = (int)((m / OW) % OH), b = (int)(m / ((long)OW * OH));
        const int iy = oy * S - pad + ky, ix0 = ox * S - pad;
        if ((unsigned)iy < (unsigned)H) {
            const float* row = x + b * sb + ci * sc + (long)iy * sh;
#pragma unroll
            for (int kx = 0; kx < 8; ++kx)
                if (kx < KW && (unsigned)(ix0 + kx) < (unsigned)W) v[kx] = row[(long)(ix0 + kx) * sw];
        }
    }
    *(uint4*)(cols + m * Kpad + seg * 8) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                                                      pack_bf16x2(v[6], v[7]));
}

// MaxPool2d(k, stride s, padding p) on NHWC bf16; 8 channels (16 B) per thread.
__global__ __launch_bounds__(256) void maxpool_nhwc_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int B, int H, int W,
                                                           int C, int k, int s, int pad, int OH, int OW) {
    const int c8 = C / 8;
    const long total = (long)B * OH * OW * c8;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int cc = (int)(idx % c8);
    const long pix = idx / c8;
    const int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH), b = (int)(pix / ((long)OW * OH));
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
    for (int dy = 0; dy < k; ++dy) {
        const int iy = oy * s - pad + dy;
        if ((unsigned)iy >= (unsigned)H) continue;
        for (int dx = 0; dx < k; ++dx) {
            const int ix = ox * s - pad + dx;
            if ((unsigned)ix >= (unsigned)W) continue;
            const uint4 v = *(const uint4*)(x + (((size_t)b * H + iy) * W + ix) * C + cc * 8);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                m[2 * e] = fmaxf(m[2 * e], __uint_as_float(w[e] << 16));
                m[2 * e + 1] = fmaxf(m[2 * e + 1], __uint_as_float(w[e] & 0xffff0000u));
            }
        }
    }
    *(uint4*)(y + (size_t)pix * C + cc * 8) = make_uint4(pack_bf16x2(m[0], m[1]), pack_bf16x2(m[2], m[3]), pack_bf16x2(m[4], m[5]),
                                                         pack_bf16x2(m[6], m[7]));
}

// AdaptiveAvgPool2d((1,1)) on NHWC bf16 -> [B, C] fp32: one block per (image, 64-channel group); thread = (row lane 0..31,
// 8-channel group 0..7) with 16-B loads, fp32 accumulation, LDS reduction over the 32 row lanes.
__global__ __launch_bounds__(256) void avgpool_nhwc_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, int HW, int C) {
    __shared__ float red[32][65];
    const int b = blockIdx.y, cg = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const bf16_t* src = x + (size_t)b * HW * C + blockIdx.x * 64 + cg * 8;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int p = rl; p < HW; p += 32) {
        const uint4 v = *(const uint4*)(src + (size_t)p * C);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[2 * e] += __uint_as_float(w[e] << 16); a[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u); }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rl][cg * 8 + e] = a[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = 0.f;
        for (int r = 0; r < 32; ++r) t += red[r][threadIdx.x];
        y[(size_t)b * C + blockIdx.x * 64 + threadIdx.x] = t / (float)HW;
    }
}

// fp32 (parity mode) pools: one thread per output element.
__global__ __launch_bounds__(256) void maxpool_nhwc_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W,
                                                               int C, int k, int s, int pad, int OH, int OW) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * OH * OW * C) return;
    const int c = (int)(idx % C);
    const long pix = idx / C;
    const int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH), b = (int)(pix / ((long)OW * OH));
    float m = -INFINITY;
    for (int dy = 0; dy < k; ++dy) {
        const int iy = oy * s - pad + dy;
        if ((unsigned)iy >= (unsigned)H) continue;
        for (int dx = 0; dx < k; ++dx) {
            const int ix = ox * s - pad + dx;
            if ((unsigned)ix < (unsigned)W) m = fmaxf(m, x[(((size_t)b * H + iy) * W + ix) * C + c]);
        }
    }
    y[idx] = m;
}

__global__ __launch_bounds__(256) void avgpool_nhwc_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int HW, int C) {
    __shared__ float red[4][64];
    const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
    float a = 0.f;
    for (int p = part; p < HW; p += 4) a += x[((size_t)b * HW + p) * C + c];
    red[part][threadIdx.x & 63] = a;
    __syncthreads();
    if (part == 0) y[(size_t)b * C + c] = (red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]) / (float)HW;
}

extern "C" int whmr_conv_im2col(const float* x, void* cols, int B, int Cin, int H, int W, int KH, int KW, int S, int pad, int Kpad,
                                long sb, long sc, long sh, long sw, void* stream) {
    const int OH = (H + 2 * pad - KH) / S + 1, OW = (W + 2 * pad - KW) / S + 1;
    const long total = (long)B * OH * OW * (Kpad / 8);
    if (total <= 0 || KW > 8 || (Kpad & 7) || Kpad < Cin * KH * 8) return (int)hipErrorInvalidValue;
    const long nb = (total + 255) / 256;
    hipLaunchKernelGGL(conv_im2col_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)cols,
                       B, Cin, H, W, KH, KW, S, pad, OH, OW, Kpad, sb, sc, sh, sw);
    WHMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int whmr_maxpool_nhwc(const void* x, void* y, int B, int H, int W, int C, int k, int s, int pad, int is_bf16, void* stream) {
    const int OH = (H + 2 * pad - k) / s + 1, OW = (W + 2 * pad - k) / s + 1;
    if (!is_bf16) {
        const long tot = (long)B * OH * OW * C;
        hipLaunchKernelGGL(maxpool_nhwc_f32_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)x,
                           (float*)y, B, H, W, C, k, s, pad, OH, OW);
        WHMR_CHECK_LAUNCH();
        return 0;
    }
    if (C % 8) return (int)hipErrorInvalidValue;
    const long total = (long)B * OH * OW * (C / 8);
    hipLaunchKernelGGL(maxpool_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       (bf16_t*)y, B, H, W, C, k, s, pad, OH, OW);
    WHMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int whmr_avgpool_nhwc(const void* x, float* y, int B, int HW, int C, int is_bf16, void* stream) {
    if (C % 64 || B <= 0 || HW <= 0) return (int)hipErrorInvalidValue;
    if (!is_bf16) {
        hipLaunchKernelGGL(avgpool_nhwc_f32_kernel, dim3(C / 64, B), dim3(256), 0, (hipStream_t)stream, (const float*)x, y, HW, C);
        WHMR_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(avgpool_nhwc_kernel, dim3(C / 64, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, y, HW, C);
    WHMR_CHECK_LAUNCH();
    return 0;
}
