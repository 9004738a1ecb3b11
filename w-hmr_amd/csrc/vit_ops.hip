// Memory-bound helpers of the ViT path: LayerNorm and the patch-embed im2col.
//   whmr_layernorm   <- nn.LayerNorm(eps=1e-6) at vit.py:125,133,212,242 (and eps=1e-5 inside the timm Block, whmr.py:423)
//   whmr_patch_im2col <- the gather half of PatchEmbed's Conv2d(k16, s16, pad 2) at vit.py:157,161 (GEMM half: gemm_*.hip)
#include "common.h"

// One wave per row; the row is held in registers (C <= 64*4*MAXV), two-pass mean / variance in fp32 like ATen's CPU kernel.
template <typename TOUT, int MAXV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                        const float* __restrict__ b, TOUT* __restrict__ y,
                                                        int rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * C;
    const int nv = C >> 2;            // float4 per row
    float4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c4 = lane + 64 * i;
        if (c4 < nv) {
            v[i] = *(const float4*)(xr + c4 * 4);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c4 = lane + 64 * i;
        if (c4 < nv) {
            const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
            q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
    TOUT* yr = y + (size_t)row * C;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c4 = lane + 64 * i;
        if (c4 < nv) {
            const float4 gg = *(const float4*)(g + c4 * 4), bb = *(const float4*)(b + c4 * 4);
            const float o0 = (v[i].x - mean) * rstd * gg.x + bb.x, o1 = (v[i].y - mean) * rstd * gg.y + bb.y;
            const float o2 = (v[i].z - mean) * rstd * gg.z + bb.z, o3 = (v[i].w - mean) * rstd * gg.w + bb.w;
            if constexpr (sizeof(TOUT) == 4) {
                *(float4*)(yr + c4 * 4) = make_float4(o0, o1, o2, o3);
            } else {
                *(uint2*)(yr + c4 * 4) = make_uint2(pack_bf16x2(o0, o1), pack_bf16x2(o2, o3));
            }
        }
    }
}

extern "C" int whmr_layernorm(const float* x, const float* gamma, const float* beta, void* y, int rows, int C,
                              float eps, int out_bf16, void* stream) {
    if (rows <= 0 || C <= 0 || (C & 3) || C > 64 * 4 * 8) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((rows + 3) / 4), block(256);
    const bool small = C <= 64 * 4 * 3;
    if (out_bf16) {
        if (small) hipLaunchKernelGGL((layernorm_kernel<bf16_t, 3>), grid, block, 0, st, x, gamma, beta, (bf16_t*)y, rows, C, eps);
        else hipLaunchKernelGGL((layernorm_kernel<bf16_t, 8>), grid, block, 0, st, x, gamma, beta, (bf16_t*)y, rows, C, eps);
    } else {
        if (small) hipLaunchKernelGGL((layernorm_kernel<float, 3>), grid, block, 0, st, x, gamma, beta, (float*)y, rows, C, eps);
        else hipLaunchKernelGGL((layernorm_kernel<float, 8>), grid, block, 0, st, x, gamma, beta, (float*)y, rows, C, eps);
    }
    WHMR_CHECK_LAUNCH();
    return 0;
}

// x: NCHW fp32 with arbitrary strides (the demo passes a sliced view, demo/tester.py:152) ->
// cols [B*Hp*Wp, Cin*P*P] (k = ci*P*P + ky*P + kx, i.e. the flattened conv weight order), zero padded borders.
// One thread per output element: consecutive lanes walk kx, so both the image reads and the patch-row writes coalesce.
template <typename TOUT>
__global__ __launch_bounds__(256) void patch_im2col_kernel(const float* __restrict__ x, TOUT* __restrict__ cols,
                                                           int B, int Cin, int H, int W, int P, int pad, int Hp, int Wp,
                                                           long sb, long sc, long sh, long sw) {
    const int K = Cin * P * P;
    const long total = (long)B * Hp * Wp * K;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int k = (int)(idx % K);
        const long m = idx / K;
        const int kx = k % P, ky = (k / P) % P, ci = k / (P * P);
        const int px = (int)(m % Wp), py = (int)((m / Wp) % Hp), b = (int)(m / ((long)Wp * Hp));
        const int iy = py * P - pad + ky, ix = px * P - pad + kx;
        float v = 0.f;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = x[b * sb + ci * sc + (long)iy * sh + (long)ix * sw];
        io<TOUT>::st(cols + idx, v);
    }
}

// P % 8 == 0: one thread per 8 consecutive kx -- index arithmetic once per 8 elements, one (bf16) or two (fp32) 16-B stores.
template <typename TOUT>
__global__ __launch_bounds__(256) void patch_im2col8_kernel(const float* __restrict__ x, TOUT* __restrict__ cols,
                                                            int B, int Cin, int H, int W, int P, int pad, int Hp, int Wp,
                                                            long sb, long sc, long sh, long sw) {
    const int K8 = Cin * P * P / 8, P8 = P / 8;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * Hp * Wp * K8) return;
    const int k8 = (int)(idx % K8);
    const int m = (int)(idx / K8);
    const int kx = (k8 % P8) * 8, ky = (k8 / P8) % P, ci = k8 / (P8 * P);
    const int px = m % Wp, py = (m / Wp) % Hp, b = m / (Wp * Hp);
    const int iy = py * P - pad + ky, ix = px * P - pad + kx;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if ((unsigned)iy < (unsigned)H) {
        const float* row = x + b * sb + ci * sc + (long)iy * sh;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if ((unsigned)(ix + e) < (unsigned)W) v[e] = row[(long)(ix + e) * sw];
    }
    TOUT* dst = cols + idx * 8;
    if constexpr (sizeof(TOUT) == 2) {
        *(uint4*)dst = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
    } else {
        *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
        *(float4*)(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
}

extern "C" int whmr_patch_im2col(const float* x, void* cols, int B, int Cin, int H, int W, int P, int pad,
                                 long sb, long sc, long sh, long sw, int out_bf16, void* stream) {
    const int Hp = (H + 2 * pad - P) / P + 1, Wp = (W + 2 * pad - P) / P + 1;
    const long total = (long)B * Hp * Wp * Cin * P * P;
    if (total <= 0) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    if (P % 8 == 0 && total / 8 < (1L << 31)) {
        dim3 grid8((unsigned)((total / 8 + 255) / 256)), block8(256);
        if (out_bf16) hipLaunchKernelGGL(patch_im2col8_kernel<bf16_t>, grid8, block8, 0, st, x, (bf16_t*)cols, B, Cin, H, W, P, pad, Hp, Wp, sb, sc, sh, sw);
        else hipLaunchKernelGGL(patch_im2col8_kernel<float>, grid8, block8, 0, st, x, (float*)cols, B, Cin, H, W, P, pad, Hp, Wp, sb, sc, sh, sw);
        WHMR_CHECK_LAUNCH();
        return 0;
    }
    const long nb = (total + 255) / 256;
    dim3 grid((unsigned)(nb < 16384 ? nb : 16384)), block(256);
    if (out_bf16) hipLaunchKernelGGL(patch_im2col_kernel<bf16_t>, grid, block, 0, st, x, (bf16_t*)cols, B, Cin, H, W, P, pad, Hp, Wp, sb, sc, sh, sw);
    else hipLaunchKernelGGL(patch_im2col_kernel<float>, grid, block, 0, st, x, (float*)cols, B, Cin, H, W, P, pad, Hp, Wp, sb, sc, sh, sw);
    WHMR_CHECK_LAUNCH();
    return 0;
}

// dst = (T)src, elementwise fp32 -> bf16 (weight preparation / activation casts that are not fused anywhere)
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        const float4 v = *(const float4*)(src + i);
        *(uint2*)(dst + i) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
    } else {
        for (long j = i; j < n; ++j) dst[j] = f32_to_bf16(src[j]);
    }
}

extern "C" int whmr_cast_f32_bf16(const float* src, void* dst, long n, void* stream) {
    if (n <= 0) return (int)hipErrorInvalidValue;
    if (((uintptr_t)src & 15) || ((uintptr_t)dst & 7)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n);
    WHMR_CHECK_LAUNCH();
    return 0;
}

// fp32 rows [rows, C] -> bf16 rows [rows, 3C] = [hi | lo | hi], x = hi + lo to 16 significand bits: the K-CONCATENATED operand of the bf16x3 numerics
// for the row-major / convolution GEMMs (whmr.py:419,488-498 deconvs and the Tz head's 7x7 conv in numerics 'bf16x3').  Against weights laid
// out [W_hi | W_hi | W_lo] along the same axis, ONE plain bf16 GEMM (K' = 3K, per convolution tap: Cin' = 3 Cin) accumulates
// x_hi.W_hi + x_lo.W_hi + x_hi.W_lo in fp32 -- the three-MFMA product of gemm_blk_x3.hip without touching the row-major kernel.
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long rows, int C4) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * C4) return;
    const long r = idx / C4;
    const int c = (int)(idx - r * C4) * 4;
    const float4 v = *(const float4*)(src + idx * 4);
    uint2 h, l;
    split_bf16x2(v.x, v.y, h.x, l.x);
    split_bf16x2(v.z, v.w, h.y, l.y);
    bf16_t* d = dst + r * (long)(12 * C4) + c;
    *(uint2*)d = h;
    *(uint2*)(d + 4 * C4) = l;
    *(uint2*)(d + 8 * C4) = h;
}

extern "C" int whmr_split3_bf16(const float* src, void* dst, long rows, int C, void* stream) {
    if (rows <= 0 || C <= 0 || (C & 3) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 7)) return (int)hipErrorInvalidValue;
    const long n = rows * (C / 4);
    hipLaunchKernelGGL(split3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, rows, C / 4);
    WHMR_CHECK_LAUNCH();
    return 0;
}

// ---- blocked-layout variants (the bf16 inference path of the ViT keeps its activations in the 512-B units of gemm_blk.hip) ----------
// LayerNorm of the fp32 residual stream x [rows/32][C/4][32][4] -> bf16 GEMM operand [rows/32][C/8][32][8] (OUT_STD = 0) or the plain
// row-major fp32 [rows, C] map the heads consume (OUT_STD = 1: the final last_norm, vit.py:242,330).  One workgroup per 32-row block
// (a contiguous 32*C*4 bytes): thread = (row, part), each part owns NPER consecutive 16-B column units of its row in registers;
// two-pass mean / variance like layernorm_kernel, partial sums combined through LDS.  OUT_STD = 2: the split-bf16 operand pair of the
// "bf16x3" numerics -- y = hi halves, y_lo = lo halves of the fp32 LayerNorm output (both blocked bf16).
template <int NPER, int PARTS, int OUT_STD>
__global__ __launch_bounds__(32 * PARTS) void layernorm_blk_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                                                                   void* __restrict__ y, int rows, int C, float eps, void* __restrict__ y_lo = nullptr,
                                                                   float* __restrict__ mean_out = nullptr) {
    __shared__ float red[2][PARTS][32];
    const int row = threadIdx.x & 31, part = threadIdx.x >> 5;
    const int rb = blockIdx.x;
    const int n4_0 = part * NPER, NC4 = C >> 2;
    const float* xb = x + ((size_t)rb * NC4 + n4_0) * 128 + row * 4;
    float4 v[NPER];
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < NPER; ++q) {
        v[q] = *(const float4*)(xb + q * 128);
        s += (v[q].x + v[q].y) + (v[q].z + v[q].w);
    }
    red[0][part][row] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int pp = 0; pp < PARTS; ++pp) tot += red[0][pp][row];
    const float mean = tot / (float)C;
    if (mean_out && part == 0) mean_out[rb * 32 + row] = mean;       // row means: the first per-row shift of the folded-LayerNorm chain (gemm_blk `shift`)
    float qs = 0.f;
#pragma unroll
    for (int q = 0; q < NPER; ++q) {
        const float dx = v[q].x - mean, dy = v[q].y - mean, dz = v[q].z - mean, dw = v[q].w - mean;
        qs += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
    red[1][part][row] = qs;
    __syncthreads();
    float var = 0.f;
#pragma unroll
    for (int pp = 0; pp < PARTS; ++pp) var += red[1][pp][row];
    const float rstd = 1.0f / sqrtf(var / (float)C + eps);
    const int m = rb * 32 + row;
    if constexpr (OUT_STD == 1) {
        // row-major fp32 output: a thread owns ONE row's columns, so direct stores would put the 32 lanes of a part on 32 different rows (16 B per
        // 128-B line and store).  Each part transposes 32 rows x 32 columns at a time through its own 4.5-KB LDS patch (written and read by the same
        // half wave: no workgroup barrier) and stores whole 128-B row pieces, 8 lanes per row.
        if constexpr (NPER % 8 == 0) {
            __shared__ float tp[PARTS][32][36];
#pragma unroll
            for (int q0 = 0; q0 < NPER; q0 += 8) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float4 gg = *(const float4*)(g + (n4_0 + q0 + q) * 4), bb = *(const float4*)(b + (n4_0 + q0 + q) * 4);
                    *(float4*)&tp[part][row][q * 4] = make_float4((v[q0 + q].x - mean) * rstd * gg.x + bb.x, (v[q0 + q].y - mean) * rstd * gg.y + bb.y,
                                                                  (v[q0 + q].z - mean) * rstd * gg.z + bb.z, (v[q0 + q].w - mean) * rstd * gg.w + bb.w);
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int t = row + 32 * k, r = t >> 3, c4 = t & 7;
                    const float4 o = *(const float4*)&tp[part][r][c4 * 4];
                    if (rb * 32 + r < rows) *(float4*)((float*)y + (size_t)(rb * 32 + r) * C + (n4_0 + q0 + c4) * 4) = o;
                }
                __builtin_amdgcn_wave_barrier();
            }
        } else {
            if (m >= rows) return;
            float* yr = (float*)y + (size_t)m * C + n4_0 * 4;
#pragma unroll
            for (int q = 0; q < NPER; ++q) {
                const float4 gg = *(const float4*)(g + (n4_0 + q) * 4), bb = *(const float4*)(b + (n4_0 + q) * 4);
                *(float4*)(yr + q * 4) = make_float4((v[q].x - mean) * rstd * gg.x + bb.x, (v[q].y - mean) * rstd * gg.y + bb.y,
                                                     (v[q].z - mean) * rstd * gg.z + bb.z, (v[q].w - mean) * rstd * gg.w + bb.w);
            }
        }
    } else if constexpr (OUT_STD == 2) {
        const size_t off = ((size_t)rb * (C >> 3) + (n4_0 >> 1)) * 256 + row * 8;
        bf16_t* yb = (bf16_t*)y + off;
        bf16_t* yl = (bf16_t*)y_lo + off;
#pragma unroll
        for (int q = 0; q < NPER; q += 2) {
            const float4 g0 = *(const float4*)(g + (n4_0 + q) * 4), b0 = *(const float4*)(b + (n4_0 + q) * 4);
            const float4 g1 = *(const float4*)(g + (n4_0 + q + 1) * 4), b1 = *(const float4*)(b + (n4_0 + q + 1) * 4);
            uint4 h, l;
            split_bf16x2((v[q].x - mean) * rstd * g0.x + b0.x, (v[q].y - mean) * rstd * g0.y + b0.y, h.x, l.x);
            split_bf16x2((v[q].z - mean) * rstd * g0.z + b0.z, (v[q].w - mean) * rstd * g0.w + b0.w, h.y, l.y);
            split_bf16x2((v[q + 1].x - mean) * rstd * g1.x + b1.x, (v[q + 1].y - mean) * rstd * g1.y + b1.y, h.z, l.z);
            split_bf16x2((v[q + 1].z - mean) * rstd * g1.z + b1.z, (v[q + 1].w - mean) * rstd * g1.w + b1.w, h.w, l.w);
            *(uint4*)(yb + (q >> 1) * 256) = h;
            *(uint4*)(yl + (q >> 1) * 256) = l;
        }
    } else {
        bf16_t* yb = (bf16_t*)y + ((size_t)rb * (C >> 3) + (n4_0 >> 1)) * 256 + row * 8;
#pragma unroll
        for (int q = 0; q < NPER; q += 2) {
            const float4 g0 = *(const float4*)(g + (n4_0 + q) * 4), b0 = *(const float4*)(b + (n4_0 + q) * 4);
            const float4 g1 = *(const float4*)(g + (n4_0 + q + 1) * 4), b1 = *(const float4*)(b + (n4_0 + q + 1) * 4);
            *(uint4*)(yb + (q >> 1) * 256) =
                make_uint4(pack_bf16x2((v[q].x - mean) * rstd * g0.x + b0.x, (v[q].y - mean) * rstd * g0.y + b0.y),
                           pack_bf16x2((v[q].z - mean) * rstd * g0.z + b0.z, (v[q].w - mean) * rstd * g0.w + b0.w),
                           pack_bf16x2((v[q + 1].x - mean) * rstd * g1.x + b1.x, (v[q + 1].y - mean) * rstd * g1.y + b1.y),
                           pack_bf16x2((v[q + 1].z - mean) * rstd * g1.z + b1.z, (v[q + 1].w - mean) * rstd * g1.w + b1.w));
        }
    }
}

template <int NPER, int PARTS>
static int launch_ln_blk(const float* x, const float* g, const float* b, void* y, int rows, int C, float eps, int out_std, hipStream_t st, void* y_lo = nullptr,
                         float* mean_out = nullptr) {
    const dim3 grid((rows + 31) / 32), block(32 * PARTS);
    if (y_lo) hipLaunchKernelGGL((layernorm_blk_kernel<NPER, PARTS, 2>), grid, block, 0, st, x, g, b, y, rows, C, eps, y_lo, mean_out);
    else if (out_std) hipLaunchKernelGGL((layernorm_blk_kernel<NPER, PARTS, 1>), grid, block, 0, st, x, g, b, y, rows, C, eps, (void*)nullptr, mean_out);
    else hipLaunchKernelGGL((layernorm_blk_kernel<NPER, PARTS, 0>), grid, block, 0, st, x, g, b, y, rows, C, eps, (void*)nullptr, mean_out);
    WHMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int whmr_layernorm_blk(const float* x, const float* gamma, const float* beta, void* y, int rows, int C, float eps, int out_std,
                                  void* stream) {
    if (rows <= 0) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    switch (C) {
        case 768: return launch_ln_blk<24, 8>(x, gamma, beta, y, rows, C, eps, out_std, st);       // ViT-B
        case 1024: return launch_ln_blk<16, 16>(x, gamma, beta, y, rows, C, eps, out_std, st);     // ViT-L
        case 1280: return launch_ln_blk<20, 16>(x, gamma, beta, y, rows, C, eps, out_std, st);     // ViT-H
        case 256: return launch_ln_blk<8, 8>(x, gamma, beta, y, rows, C, eps, out_std, st);
    }
    return (int)hipErrorInvalidValue;
}

// whmr_layernorm_blk (out_std 0) that also writes the row means [ceil(rows/32)*32]: the first LayerNorm of the folded chain runs as an explicit pass
// (the patch-embed producer has no earlier statistics to centre its bf16 copy with) and seeds the per-row shift of the producers behind it.
extern "C" int whmr_layernorm_blk_mean(const float* x, const float* gamma, const float* beta, void* y, float* mean_out, int rows, int C, float eps,
                                       void* stream) {
    if (rows <= 0 || !mean_out) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    switch (C) {
        case 768: return launch_ln_blk<24, 8>(x, gamma, beta, y, rows, C, eps, 0, st, nullptr, mean_out);
        case 1024: return launch_ln_blk<16, 16>(x, gamma, beta, y, rows, C, eps, 0, st, nullptr, mean_out);
        case 1280: return launch_ln_blk<20, 16>(x, gamma, beta, y, rows, C, eps, 0, st, nullptr, mean_out);
        case 256: return launch_ln_blk<8, 8>(x, gamma, beta, y, rows, C, eps, 0, st, nullptr, mean_out);
    }
    return (int)hipErrorInvalidValue;
}

// Same LayerNorm with the result as a split-bf16 operand pair (y_hi + y_lo ~= LN(x) to 16 significand bits): the "bf16x3" numerics.
extern "C" int whmr_layernorm_blk_x3(const float* x, const float* gamma, const float* beta, void* y_hi, void* y_lo, float* mean_out, int rows, int C,
                                     float eps, void* stream) {
    if (rows <= 0 || !y_hi || !y_lo) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    switch (C) {
        case 768: return launch_ln_blk<24, 8>(x, gamma, beta, y_hi, rows, C, eps, 0, st, y_lo, mean_out);
        case 1024: return launch_ln_blk<16, 16>(x, gamma, beta, y_hi, rows, C, eps, 0, st, y_lo, mean_out);
        case 1280: return launch_ln_blk<20, 16>(x, gamma, beta, y_hi, rows, C, eps, 0, st, y_lo, mean_out);
        case 256: return launch_ln_blk<8, 8>(x, gamma, beta, y_hi, rows, C, eps, 0, st, y_lo, mean_out);
    }
    return (int)hipErrorInvalidValue;
}

// PatchEmbed gather into the blocked bf16 operand layout [ceil(M/32)][K/8][32][8] (P % 8 == 0): thread = (row in block, k unit),
// rows fastest, so a half-wave writes one contiguous 512-B unit.
__global__ __launch_bounds__(256) void patch_im2col_blk_kernel(const float* __restrict__ x, bf16_t* __restrict__ cols, int B, int Cin, int H, int W,
                                                               int P, int pad, int Hp, int Wp, long sb, long sc, long sh, long sw,
                                                               bf16_t* __restrict__ cols_lo) {
    const int K8 = Cin * P * P / 8, P8 = P / 8;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int M = B * Hp * Wp;
    const int r = (int)(idx & 31);
    const long u = idx >> 5;
    const int k8 = (int)(u % K8);
    const int rb = (int)(u / K8);
    const int m = rb * 32 + r;
    if (rb > (M - 1) / 32) return;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (m < M) {
        const int kx = (k8 % P8) * 8, ky = (k8 / P8) % P, ci = k8 / (P8 * P);
        const int px = m % Wp, py = (m / Wp) % Hp, b = m / (Wp * Hp);
        const int iy = py * P - pad + ky, ix = px * P - pad + kx;
        if ((unsigned)iy < (unsigned)H) {
            const float* row = x + b * sb + ci * sc + (long)iy * sh;
            if (sw == 1 && !((ix | W) & 1) && !((uintptr_t)(row + ix) & 7)) {
                // unit pixel stride, even start and width: a pair of pixels is inside or outside together -- four 8-byte loads instead of eight dwords
#pragma unroll
                for (int e = 0; e < 8; e += 2)
                    if ((unsigned)(ix + e) < (unsigned)W) { const float2 t = *(const float2*)(row + ix + e); v[e] = t.x; v[e + 1] = t.y; }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if ((unsigned)(ix + e) < (unsigned)W) v[e] = row[(long)(ix + e) * sw];
            }
        }
    }
    if (cols_lo) {                                      // split-bf16 pair (bf16x3 numerics): pixel = hi + lo
        uint4 h, l;
        split_bf16x2(v[0], v[1], h.x, l.x); split_bf16x2(v[2], v[3], h.y, l.y); split_bf16x2(v[4], v[5], h.z, l.z); split_bf16x2(v[6], v[7], h.w, l.w);
        *(uint4*)(cols + ((size_t)rb * K8 + k8) * 256 + r * 8) = h;
        *(uint4*)(cols_lo + ((size_t)rb * K8 + k8) * 256 + r * 8) = l;
        return;
    }
    *(uint4*)(cols + ((size_t)rb * K8 + k8) * 256 + r * 8) =
        make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
}

static int patch_im2col_blk_launch(const float* x, void* cols, void* cols_lo, int B, int Cin, int H, int W, int P, int pad,
                                   long sb, long sc, long sh, long sw, void* stream) {
    const int Hp = (H + 2 * pad - P) / P + 1, Wp = (W + 2 * pad - P) / P + 1;
    if (B <= 0 || Hp <= 0 || Wp <= 0 || (P % 8)) return (int)hipErrorInvalidValue;
    const long M = (long)B * Hp * Wp, K8 = (long)Cin * P * P / 8;
    const long total = ((M + 31) / 32) * 32 * K8;
    hipLaunchKernelGGL(patch_im2col_blk_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)cols, B, Cin, H, W,
                       P, pad, Hp, Wp, sb, sc, sh, sw, (bf16_t*)cols_lo);
    WHMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int whmr_patch_im2col_blk(const float* x, void* cols, int B, int Cin, int H, int W, int P, int pad,
                                     long sb, long sc, long sh, long sw, void* stream) {
    return patch_im2col_blk_launch(x, cols, nullptr, B, Cin, H, W, P, pad, sb, sc, sh, sw, stream);
}

// the same gather with the pixels as a split-bf16 pair (cols_hi + cols_lo = the fp32 pixel to 16 significand bits): bf16x3 numerics
extern "C" int whmr_patch_im2col_blk_x3(const float* x, void* cols_hi, void* cols_lo, int B, int Cin, int H, int W, int P, int pad,
                                        long sb, long sc, long sh, long sw, void* stream) {
    if (!cols_lo) return (int)hipErrorInvalidValue;
    return patch_im2col_blk_launch(x, cols_hi, cols_lo, B, Cin, H, W, P, pad, sb, sc, sh, sw, stream);
}
