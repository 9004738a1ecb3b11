// Second convolution of the Tz / focal head (whmr.py:420): Conv2d(64, 5, k7, s2) on the NHWC map of conv0 -> tokens [B, 5, 216]
// (the reference's reshape(B, 5, -1) of the NCHW result, whmr.py:571).  N = 5 output channels is no GEMM shape: one wave per
// output pixel, lane = input channel, 49 taps x 5 FMAs per lane, wave reduction; fp32 accumulation and weights.
// (A one-workgroup-per-image fusion of the following timm Block + est_Tz tail was tried: 265 us vs ~70 us for the 8 small
// launches on the pipelined skinny GEMM -- 5 tokens per image give every weight load only 5 FMAs; dropped.)
#include "common.h"

// PAIR (fp32 only): the map has 128 channels per pixel and channel c of the convolution's input is x[c] + x[64 + c] -- the bf16x3 form of conv0 leaves the
// hi.hi + lo.hi products in columns 0..63 and the hi.lo product in columns 64..127 (models/whmr.py::_tz_operands); they are added as they are read.
template <typename T, bool PAIR = false>
__global__ __launch_bounds__(256) void tz_conv1_kernel(const T* __restrict__ x, const float* __restrict__ w /*[5][49][64]*/,
                                                       float* __restrict__ tok, int B, int IH, int IW, int OH, int OW) {
    constexpr int CS = PAIR ? 128 : 64;                     // channels per pixel in memory
    const int lane = threadIdx.x & 63;
    const int pix = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int npix = OH * OW;
    if (pix >= B * npix) return;
    const int b = pix / npix, p = pix - b * npix;
    const int oy = p / OW, ox = p - oy * OW;
    const T* xb = x + ((size_t)b * IH * IW) * CS + lane;
    float a[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
            const T* xp = xb + ((size_t)(oy * 2 + ky) * IW + (ox * 2 + kx)) * CS;
            float v = io<T>::ld(xp);
            if constexpr (PAIR) v += io<T>::ld(xp + 64);
            const float* wt = w + (ky * 7 + kx) * 64 + lane;
#pragma unroll
            for (int n = 0; n < 5; ++n) a[n] = fmaf(v, wt[n * 49 * 64], a[n]);
        }
#pragma unroll
    for (int n = 0; n < 5; ++n) {
        const float s = wave_sum(a[n]);
        if (lane == 0) tok[((size_t)b * 5 + n) * npix + p] = s;
    }
}

extern "C" int whmr_tz_conv1(const void* x, int x_bf16, const float* w, float* tok, int B, int IH, int IW, void* stream) {
    const int OH = (IH - 7) / 2 + 1, OW = (IW - 7) / 2 + 1;
    if (B <= 0 || OH <= 0 || OW <= 0) return (int)hipErrorInvalidValue;
    const int waves = B * OH * OW;
    hipStream_t st = (hipStream_t)stream;
    if (x_bf16 == 2) hipLaunchKernelGGL((tz_conv1_kernel<float, true>), dim3((waves + 3) / 4), dim3(256), 0, st, (const float*)x, w, tok, B, IH, IW, OH, OW);
    else if (x_bf16) hipLaunchKernelGGL(tz_conv1_kernel<bf16_t>, dim3((waves + 3) / 4), dim3(256), 0, st, (const bf16_t*)x, w, tok, B, IH, IW, OH, OW);
    else hipLaunchKernelGGL(tz_conv1_kernel<float>, dim3((waves + 3) / 4), dim3(256), 0, st, (const float*)x, w, tok, B, IH, IW, OH, OW);
    WHMR_CHECK_LAUNCH();
    return 0;
}

// ---- Composed form of the two Tz-head convolutions (whmr.py:418-421 applied at :567-571), inference views only.
// conv1(conv0(x)) has no bias and no activation in between, so it IS one Conv2d(256, 5, k25, s6):
//     Wc[o, ci, A, B] = sum_{c1, 3u + a = A, 3v + b = B} w1[o, c1, u, v] * w0[c1, ci, a, b]            (built once on the host side in fp64).
// With y = 6 Y + q, x = 6 m + p the NHWC map [B, 128, 96, C] is the dense matrix [(b, y, m), (p, ci)] and
//     P[(b, Y, m), (jA, jB, o)] = sum_{q, p, ci} x[b, 6 Y + q, 6 m + p, ci] * Wc[o, ci, q + 6 jA, p + 6 jB]
// is an implicit GEMM (kernel 6 x 1, stride 6 x 1 over the [128, 16, 6 C] view) that reads every map byte exactly ONCE -- the 7 x 7 / stride-3 form
// gathered each pixel ~5 times through the LDS-DMA path for N = 64 columns of MFMA work.  This kernel is the 25-term tail
//     tok[b, o, r, s] = sum_{jA, jB} P[(b, r + jA, s + jB), (jA, jB, o)]        (output pixel (r, s) reads input rows 6 r + A, A = q + 6 jA)
// over `nsplit` split-K partial planes and `halves` column halves (bf16x3: the W_lo product sits in columns 128 + n).  One wave per (b, r):
// lane = (s, o), 25 loads of 5 contiguous floats per pixel row.
__global__ __launch_bounds__(256) void tz_fold_kernel(const float* __restrict__ P, int ldp, int halves, int nsplit, long split_stride,
                                                      float* __restrict__ tok, int B, int OHp, int OWp, int OH, int OW) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= B * OH) return;
    const int b = w / OH, r = w - b * OH;
    for (int e = lane; e < OW * 5; e += 64) {
        const int s = e / 5, o = e - s * 5;
        float acc = 0.f;
        for (int sp = 0; sp < nsplit; ++sp) {
            const float* Ps = P + sp * split_stride;
#pragma unroll
            for (int jA = 0; jA < 5; ++jA)
#pragma unroll
                for (int jB = 0; jB < 5; ++jB) {
                    const float* row = Ps + ((size_t)(b * OHp + r + jA) * OWp + s + jB) * ldp + (jA * 5 + jB) * 5 + o;
                    acc += row[0];
                    if (halves == 2) acc += row[128];
                }
        }
        tok[((size_t)b * 5 + o) * (OH * OW) + r * OW + s] = acc;
    }
}

extern "C" int whmr_tz_fold(const float* P, int ldp, int halves, int nsplit, long split_stride, float* tok, int B, int OHp, int OWp, int OH,
                            int OW, void* stream) {
    if (B <= 0 || OH <= 0 || OW <= 0 || OH + 4 > OHp || OW + 4 > OWp || (halves != 1 && halves != 2) || nsplit < 1 || ldp < 128 * halves)
        return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(tz_fold_kernel, dim3((B * OH + 3) / 4), dim3(256), 0, (hipStream_t)stream, P, ldp, halves, nsplit, split_stride, tok, B, OHp,
                       OWp, OH, OW);
    WHMR_CHECK_LAUNCH();
    return 0;
}

// ---- the composed convolution in the TRAINING graph (w-hmr_amd/train/heads_autograd.py::TzComposedFn): autograd of whmr.py:567-571 through
// Wc = compose(w0, w1) instead of through the 64-channel map between the two convolutions.
// T [(u, v, o) = 245][(ci, a, b) = Ci * 49] fp32 = w1p . w0 (one small GEMM) -> the space-to-depth weight matrix g [128][(q, p, ci) = 36 Ci] (bf16 or fp32):
// g[(jA, jB, o)][(q, p, ci)] = Wc[o, ci, A = q + 6 jA, B = p + 6 jB] = sum_{3u + a = A, 3v + b = B} T[(u, v, o)][(ci, a, b)], zero for A, B > 24 and rows >= 125.
__global__ __launch_bounds__(256) void tz_compose_kernel(const float* __restrict__ T, int Ci, void* __restrict__ g, int g_bf16) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int ldk = 36 * Ci;
    if (idx >= 128L * ldk) return;
    const int n = (int)(idx / ldk), k = (int)(idx - (long)n * ldk);
    const int q = k / (6 * Ci), p = (k / Ci) % 6, ci = k % Ci;
    float acc = 0.f;
    if (n < 125) {
        const int j = n / 5, o = n - j * 5, A = q + 6 * (j / 5), Bc = p + 6 * (j % 5);
        if (A < 25 && Bc < 25) {
            const int ldt = Ci * 49;
            for (int u = (A > 6 ? (A - 4) / 3 : 0); 3 * u <= A && u < 7; ++u) {
                const int a = A - 3 * u;
                if (a > 6) continue;
                for (int v = (Bc > 6 ? (Bc - 4) / 3 : 0); 3 * v <= Bc && v < 7; ++v) {
                    const int b = Bc - 3 * v;
                    if (b > 6) continue;
                    acc += T[(size_t)((u * 7 + v) * 5 + o) * ldt + ci * 49 + a * 7 + b];
                }
            }
        }
    }
    if (g_bf16) ((bf16_t*)g)[idx] = f32_to_bf16(acc); else ((float*)g)[idx] = acc;
}

// transpose of the above: dT[(u, v, o)][(ci, a, b)] = dG[(jA, jB, o)][(q, p, ci)] at A = 3u + a, B = 3v + b (a pure gather: every (A, B) <= 24 has one home)
__global__ __launch_bounds__(256) void tz_compose_bwd_kernel(const float* __restrict__ dG, int Ci, float* __restrict__ dT) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int ldt = Ci * 49;
    if (idx >= 245L * ldt) return;
    const int row = (int)(idx / ldt), col = (int)(idx - (long)row * ldt);
    const int o = row % 5, uv = row / 5, u = uv / 7, v = uv - u * 7;
    const int ci = col / 49, ab = col - ci * 49, a = ab / 7, b = ab - a * 7;
    const int A = 3 * u + a, Bc = 3 * v + b;
    dT[idx] = dG[(size_t)(((A / 6) * 5 + Bc / 6) * 5 + o) * (36 * Ci) + ((A % 6) * 6 + Bc % 6) * Ci + ci];
}

// transpose of tz_fold_kernel: dP[(b, Y, m)][(jA, jB, o)] = dtok[b, o, Y - jA, m - jB] (zero outside the OH x OW token grid and in columns 125 .. 127), bf16
__global__ __launch_bounds__(256) void tz_unfold_kernel(const float* __restrict__ dtok, bf16_t* __restrict__ dP, int B, int OHp, int OWp, int OH, int OW) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)B * OHp * OWp * 128) return;
    const int n = (int)(idx & 127);
    const long row = idx >> 7;
    const int m = (int)(row % OWp), Y = (int)((row / OWp) % OHp), b = (int)(row / ((long)OWp * OHp));
    float v = 0.f;
    if (n < 125) {
        const int j = n / 5, o = n - j * 5, r = Y - j / 5, s = m - j % 5;
        if (r >= 0 && r < OH && s >= 0 && s < OW) v = dtok[(((size_t)b * 5 + o) * OH + r) * OW + s];
    }
    dP[idx] = f32_to_bf16(v);
}

extern "C" int whmr_tz_compose(const float* T, int Ci, void* g, int g_bf16, void* stream) {
    if (!T || !g || Ci <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(tz_compose_kernel, dim3((unsigned)((128L * 36 * Ci + 255) / 256)), dim3(256), 0, (hipStream_t)stream, T, Ci, g, g_bf16);
    WHMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int whmr_tz_compose_bwd(const float* dG, int Ci, float* dT, void* stream) {
    if (!dG || !dT || Ci <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(tz_compose_bwd_kernel, dim3((unsigned)((245L * 49 * Ci + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dG, Ci, dT);
    WHMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int whmr_tz_unfold(const float* dtok, void* dP, int B, int OHp, int OWp, int OH, int OW, void* stream) {
    if (!dtok || !dP || B <= 0 || OH <= 0 || OW <= 0 || OH + 4 > OHp || OW + 4 > OWp) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(tz_unfold_kernel, dim3((unsigned)(((long)B * OHp * OWp * 128 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dtok, (bf16_t*)dP, B,
                       OHp, OWp, OH, OW);
    WHMR_CHECK_LAUNCH();
    return 0;
}
