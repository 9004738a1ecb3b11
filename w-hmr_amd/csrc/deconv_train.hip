// Training-mode pieces of the deconv pyramid: ConvTranspose2d(k4,s2,p1) -> BatchNorm2d (batch statistics) -> ReLU
// (reference: models/whmr.py:459-501 layers, :560-564 use; autograd of those modules as driven by core/trainer.py:410-470).
// The three matrix products (forward sub-pixel phases, dX = strided conv of dZ, dW = X^T . col(dZ)) run on the GEMM kernels
// (gemm_bf16*.hip / gemm_f32.hip); this file holds the memory-bound kernels around them, all on channels-last [M, C] maps:
//   whmr_bn_stats        per-channel mean / 1/sqrt(var+eps) over M = B*H*W rows (two-stage, deterministic) + running-stat update
//   whmr_bn_apply_relu   y = relu(z*a + b),  a = gamma*invstd, b = beta - mean*a
//   whmr_bn_relu_bwd     g = dy*[z*a+b > 0];  dbeta = sum g, dgamma = sum g*xhat;  dz = a*(g - dbeta/M - xhat*dgamma/M)
//   whmr_im2col_t        T[(ky,kx,c), m] = src[b, oy*S+ky-P, ox*S+kx-P, c]: the transposed column matrix of dZ, operand of dW
// Every thread owns 8 consecutive channels of a row (16-B bf16 / 2x16-B fp32 accesses); a block walks a contiguous row chunk.
#include "common.h"

#define BN_PARTS 1024           // row chunks of the two-stage reductions (4 blocks per CU)

template <typename T> __device__ __forceinline__ void load8(const T* p, float v[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float v[8]) {
    const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float v[8]) {
    const uint4 q = *(const uint4*)p;
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[2 * e] = __uint_as_float(w[e] << 16); v[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float v[8]);
template <> __device__ __forceinline__ void store8<float>(float* p, const float v[8]) {
    *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
    *(float4*)(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float v[8]) {
    *(uint4*)p = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
}

// MODE 0: s1 = sum z, s2 = sum (z - shift)^2 with shift = the channel's first-row value (guards the E[z^2] - mean^2 cancellation);
// MODE 1: s1 = sum g, s2 = sum g * xhat.   partial[part][2][C].  stats = [mean | invstd | a | b] (MODE 1 only).
template <typename TZ, typename TDY, int MODE>
__global__ __launch_bounds__(256) void bn_partial_kernel(const TZ* __restrict__ z, const TDY* __restrict__ dy, const float* __restrict__ stats,
                                                         long M, int C, float* __restrict__ partial) {
    extern __shared__ float red[];                  // [2][RPB][C]
    const int tpr = C >> 3, rpb = 256 / tpr;
    const int cg = threadIdx.x % tpr, rl = threadIdx.x / tpr, c = cg * 8;
    const long rows_per = (M + gridDim.x - 1) / gridDim.x;
    const long r_begin = (long)blockIdx.x * rows_per, r_end = min(M, r_begin + rows_per);
    float s1[8], s2[8], k0[8], k1[8], k2[8], k3[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
    if (MODE == 0) {
        load8<TZ>(z + c, k0);                        // shift = row 0 (same for every block: the partials add up)
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) { k0[e] = stats[c + e]; k1[e] = stats[C + c + e]; k2[e] = stats[2 * C + c + e]; k3[e] = stats[3 * C + c + e]; }
    }
#pragma unroll 4
    for (long r = r_begin + rl; r < r_end; r += rpb) {
        float v[8];
        load8<TZ>(z + r * C + c, v);
        if (MODE == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[e] - k0[e]; s1[e] += d; s2[e] = fmaf(d, d, s2[e]); }
        } else {
            float g[8];
            load8<TDY>(dy + r * C + c, g);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float gg = fmaf(v[e], k2[e], k3[e]) > 0.f ? g[e] : 0.f;
                s1[e] += gg;
                s2[e] = fmaf(gg, (v[e] - k0[e]) * k1[e], s2[e]);
            }
        }
    }
    float* r1 = red, *r2 = red + rpb * C;
#pragma unroll
    for (int e = 0; e < 8; ++e) { r1[rl * C + c + e] = s1[e]; r2[rl * C + c + e] = s2[e]; }
    __syncthreads();
    for (int cc = threadIdx.x; cc < C; cc += 256) {
        float a = 0.f, b = 0.f;
        for (int r = 0; r < rpb; ++r) { a += r1[r * C + cc]; b += r2[r * C + cc]; }
        partial[((size_t)blockIdx.x * 2) * C + cc] = a;
        partial[((size_t)blockIdx.x * 2 + 1) * C + cc] = b;
    }
}

// block = 16 channels x 16 part lanes; fixed-order double sums.
// MODE 0: stats = [mean | invstd | a | b], running stats updated (momentum; unbiased variance) when given.
// MODE 1: dbeta, dgamma (+= when accumulate), coef = [dbeta/M | dgamma/M].
// MODE 2 (SyncBatchNorm forward): sums64 = [sum z | sum z^2 | M] in double, un-shifted -- what the ranks add up (one all-reduce) before
//         bn_stats_from_sums_kernel.  The shifted partials keep the fp32 accumulation of a row chunk well conditioned; the un-shift
//         (s1 + M shift, s2 + 2 shift s1 + M shift^2) happens here in double, so E[z^2] - mean^2 of the GLOBAL batch has 53-bit operands.
// MODE 3 (SyncBatchNorm backward): sums64 = [sum g | sum g xhat] of THIS rank's rows in double (xhat from the global statistics) and the
//         LOCAL dbeta / dgamma (torch's SyncBatchNorm semantics: the parameter gradients stay per rank, the data-parallel reducer averages them).
template <int MODE>
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partial, int nparts, long M, int C, const float* __restrict__ z0_f32,
                                                          const bf16_t* __restrict__ z0_bf16, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps, float momentum, float* __restrict__ stats,
                                                          float* __restrict__ running_mean, float* __restrict__ running_var,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate,
                                                          float* __restrict__ coef, double* __restrict__ sums64 = nullptr) {
    __shared__ double red[2][16][17];
    const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4, c = blockIdx.x * 16 + cl;
    double a = 0.0, b = 0.0;
    if (c < C) {
#pragma unroll 8
        for (int p = pl; p < nparts; p += 16) {
            a += (double)partial[((size_t)p * 2) * C + c];
            b += (double)partial[((size_t)p * 2 + 1) * C + c];
        }
    }
    red[0][pl][cl] = a; red[1][pl][cl] = b;
    __syncthreads();
    if (pl == 0 && c < C) {
        double s1 = 0.0, s2 = 0.0;
        for (int p = 0; p < 16; ++p) { s1 += red[0][p][cl]; s2 += red[1][p][cl]; }
        if (MODE == 0) {
            const double shift = z0_f32 ? (double)z0_f32[c] : (double)bf16_to_f32(z0_bf16[c]);
            const double dm = s1 / (double)M;                   // mean - shift
            const double mean = shift + dm;
            double var = s2 / (double)M - dm * dm;              // biased variance (what the normalisation uses)
            if (var < 0.0) var = 0.0;
            const float invstd = (float)(1.0 / sqrt(var + (double)eps));
            const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
            const float aa = g * invstd;
            stats[c] = (float)mean; stats[C + c] = invstd; stats[2 * C + c] = aa; stats[3 * C + c] = bt - (float)mean * aa;
            if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
            if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(var * (double)M / (double)(M > 1 ? M - 1 : 1));
        } else if (MODE == 2) {
            const double shift = z0_f32 ? (double)z0_f32[c] : (double)bf16_to_f32(z0_bf16[c]);
            sums64[c] = s1 + (double)M * shift;
            sums64[C + c] = s2 + 2.0 * shift * s1 + (double)M * shift * shift;
            if (c == 0) sums64[2 * C] = (double)M;
        } else {
            dbeta[c] = accumulate ? dbeta[c] + (float)s1 : (float)s1;
            dgamma[c] = accumulate ? dgamma[c] + (float)s2 : (float)s2;
            if (MODE == 1) { coef[c] = (float)(s1 / (double)M); coef[C + c] = (float)(s2 / (double)M); }
            else { sums64[c] = s1; sums64[C + c] = s2; }
        }
    }
}

// SyncBatchNorm: stats = [mean | invstd | a | b] from the moments of the GLOBAL batch (sums64 = [sum z | sum z^2 | rows], already summed over
// the ranks), running statistics updated with the global batch's unbiased variance -- the tail of bn_finalize_kernel<0> on other inputs.
__global__ __launch_bounds__(256) void bn_stats_from_sums_kernel(const double* __restrict__ sums64, int C, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float eps, float momentum, float* __restrict__ stats,
                                                                 float* __restrict__ running_mean, float* __restrict__ running_var) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const double n = sums64[2 * C];
    const double mean = sums64[c] / n;
    double var = sums64[C + c] / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
    const float aa = g * invstd;
    stats[c] = (float)mean; stats[C + c] = invstd; stats[2 * C + c] = aa; stats[3 * C + c] = bt - (float)mean * aa;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(var * n / (n > 1.0 ? n - 1.0 : 1.0));
}

// SyncBatchNorm backward: coef = [sum g / N | sum g xhat / N] from the globally summed sums and the global row count
__global__ __launch_bounds__(256) void bn_coef_from_sums_kernel(const double* __restrict__ sums64, const double* __restrict__ count, int C, float* __restrict__ coef) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const double n = *count;
    coef[c] = (float)(sums64[c] / n); coef[C + c] = (float)(sums64[C + c] / n);
}

template <typename TZ, typename TY>
__global__ __launch_bounds__(256) void bn_apply_relu_kernel(const TZ* __restrict__ z, const float* __restrict__ stats, TY* __restrict__ y, long M, int C) {
    const int tpr = C >> 3, rpb = 256 / tpr;
    const int cg = threadIdx.x % tpr, rl = threadIdx.x / tpr, c = cg * 8;
    float a[8], b[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = stats[2 * C + c + e]; b[e] = stats[3 * C + c + e]; }
#pragma unroll 2
    for (long r = (long)blockIdx.x * rpb + rl; r < M; r += (long)gridDim.x * rpb) {
        float v[8];
        load8<TZ>(z + r * C + c, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(fmaf(v[e], a[e], b[e]), 0.f);
        store8<TY>(y + r * C + c, v);
    }
}

template <typename TZ, typename TDY, typename TDZ>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const TZ* __restrict__ z, const TDY* __restrict__ dy, const float* __restrict__ stats,
                                                           const float* __restrict__ coef, TDZ* __restrict__ dz, long M, int C) {
    const int tpr = C >> 3, rpb = 256 / tpr;
    const int cg = threadIdx.x % tpr, rl = threadIdx.x / tpr, c = cg * 8;
    float mean[8], istd[8], a[8], b[8], dbm[8], dgm[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        mean[e] = stats[c + e]; istd[e] = stats[C + c + e]; a[e] = stats[2 * C + c + e]; b[e] = stats[3 * C + c + e];
        dbm[e] = coef[c + e]; dgm[e] = coef[C + c + e];
    }
#pragma unroll 2
    for (long r = (long)blockIdx.x * rpb + rl; r < M; r += (long)gridDim.x * rpb) {
        float v[8], g[8];
        load8<TZ>(z + r * C + c, v);
        load8<TDY>(dy + r * C + c, g);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float gg = fmaf(v[e], a[e], b[e]) > 0.f ? g[e] : 0.f;
            const float xh = (v[e] - mean[e]) * istd[e];
            v[e] = a[e] * (gg - dbm[e] - xh * dgm[e]);
        }
        store8<TDZ>(dz + r * C + c, v);
    }
}

static inline bool bn_shape_ok(long M, int C) { return M > 0 && C >= 8 && C <= 2048 && (C & 7) == 0 && 256 % (C >> 3) == 0; }

extern "C" int whmr_bn_stats(const void* z, int z_bf16, long M, int C, const float* gamma, const float* beta, float eps, float momentum,
                             float* running_mean, float* running_var, float* stats, float* scratch, void* stream) {
    if (!bn_shape_ok(M, C)) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    const int rpb = 256 / (C >> 3);
    const int parts = (int)min((long)BN_PARTS, (M + rpb - 1) / rpb);
    const size_t lds = (size_t)2 * rpb * C * sizeof(float);
    if (z_bf16) hipLaunchKernelGGL((bn_partial_kernel<bf16_t, bf16_t, 0>), dim3(parts), dim3(256), lds, st, (const bf16_t*)z, (const bf16_t*)nullptr, (const float*)nullptr, M, C, scratch);
    else hipLaunchKernelGGL((bn_partial_kernel<float, float, 0>), dim3(parts), dim3(256), lds, st, (const float*)z, (const float*)nullptr, (const float*)nullptr, M, C, scratch);
    WHMR_CHECK_LAUNCH();
    hipLaunchKernelGGL((bn_finalize_kernel<0>), dim3((C + 15) / 16), dim3(256), 0, st, scratch, parts, M, C, z_bf16 ? nullptr : (const float*)z,
                       z_bf16 ? (const bf16_t*)z : nullptr, gamma, beta, eps, momentum, stats, running_mean, running_var, (float*)nullptr,
                       (float*)nullptr, 0, (float*)nullptr);
    WHMR_CHECK_LAUNCH();
    return 0;
}

// ---- SyncBatchNorm (core/trainer.py:83: convert_sync_batchnorm before DDP): whmr_bn_stats cut where the ranks' sums meet.
//   whmr_bn_sums             sums64 [2C + 1] = sum z | sum z^2 | rows of THIS rank        -> all-reduce(SUM) over the ranks (host)
//   whmr_bn_stats_from_sums  stats [4C] (+ running statistics) from the summed vector
// scratch: >= BN_PARTS * 2 * C floats.
extern "C" int whmr_bn_sums(const void* z, int z_bf16, long M, int C, double* sums64, float* scratch, void* stream) {
    if (!bn_shape_ok(M, C) || !sums64) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    const int rpb = 256 / (C >> 3);
    const int parts = (int)min((long)BN_PARTS, (M + rpb - 1) / rpb);
    const size_t lds = (size_t)2 * rpb * C * sizeof(float);
    if (z_bf16) hipLaunchKernelGGL((bn_partial_kernel<bf16_t, bf16_t, 0>), dim3(parts), dim3(256), lds, st, (const bf16_t*)z, (const bf16_t*)nullptr, (const float*)nullptr, M, C, scratch);
    else hipLaunchKernelGGL((bn_partial_kernel<float, float, 0>), dim3(parts), dim3(256), lds, st, (const float*)z, (const float*)nullptr, (const float*)nullptr, M, C, scratch);
    WHMR_CHECK_LAUNCH();
    hipLaunchKernelGGL((bn_finalize_kernel<2>), dim3((C + 15) / 16), dim3(256), 0, st, scratch, parts, M, C, z_bf16 ? nullptr : (const float*)z,
                       z_bf16 ? (const bf16_t*)z : nullptr, (const float*)nullptr, (const float*)nullptr, 0.f, 0.f, (float*)nullptr, (float*)nullptr,
                       (float*)nullptr, (float*)nullptr, (float*)nullptr, 0, (float*)nullptr, sums64);
    WHMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int whmr_bn_stats_from_sums(const double* sums64, int C, const float* gamma, const float* beta, float eps, float momentum,
                                       float* running_mean, float* running_var, float* stats, void* stream) {
    if (C <= 0 || !sums64 || !stats) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(bn_stats_from_sums_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums64, C, gamma, beta, eps, momentum, stats,
                       running_mean, running_var);
    WHMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int whmr_bn_apply_relu(const void* z, int z_bf16, const float* stats, void* y, int y_bf16, long M, int C, void* stream) {
    if (!bn_shape_ok(M, C)) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    const int rpb = 256 / (C >> 3);
    const int blocks = (int)min((long)4096, (M + rpb - 1) / rpb);
    if (z_bf16 && y_bf16) hipLaunchKernelGGL((bn_apply_relu_kernel<bf16_t, bf16_t>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)z, stats, (bf16_t*)y, M, C);
    else if (!z_bf16 && !y_bf16) hipLaunchKernelGGL((bn_apply_relu_kernel<float, float>), dim3(blocks), dim3(256), 0, st, (const float*)z, stats, (float*)y, M, C);
    else if (!z_bf16 && y_bf16) hipLaunchKernelGGL((bn_apply_relu_kernel<float, bf16_t>), dim3(blocks), dim3(256), 0, st, (const float*)z, stats, (bf16_t*)y, M, C);
    else return (int)hipErrorInvalidValue;
    WHMR_CHECK_LAUNCH();
    return 0;
}

template <typename TZ, typename TDY, typename TDZ>
static int bn_bwd_launch(const void* z, const void* dy, const float* stats, void* dz, float* dgamma, float* dbeta, int accumulate, long M, int C,
                         float* scratch, hipStream_t st) {
    const int rpb = 256 / (C >> 3);
    const int parts = (int)min((long)BN_PARTS, (M + rpb - 1) / rpb);
    const size_t lds = (size_t)2 * rpb * C * sizeof(float);
    float* coef = scratch + (size_t)BN_PARTS * 2 * C;
    hipLaunchKernelGGL((bn_partial_kernel<TZ, TDY, 1>), dim3(parts), dim3(256), lds, st, (const TZ*)z, (const TDY*)dy, stats, M, C, scratch);
    WHMR_CHECK_LAUNCH();
    hipLaunchKernelGGL((bn_finalize_kernel<1>), dim3((C + 15) / 16), dim3(256), 0, st, scratch, parts, M, C, (const float*)nullptr, (const bf16_t*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, 0.f, 0.f, (float*)nullptr, (float*)nullptr, (float*)nullptr, dgamma, dbeta,
                       accumulate, coef);
    WHMR_CHECK_LAUNCH();
    const int blocks = (int)min((long)4096, (M + rpb - 1) / rpb);
    hipLaunchKernelGGL((bn_bwd_apply_kernel<TZ, TDY, TDZ>), dim3(blocks), dim3(256), 0, st, (const TZ*)z, (const TDY*)dy, stats, coef, (TDZ*)dz, M, C);
    WHMR_CHECK_LAUNCH();
    return 0;
}

// scratch: >= (BN_PARTS*2 + 2) * C floats
extern "C" int whmr_bn_relu_bwd(const void* z, int z_bf16, const void* dy, int dy_bf16, const float* stats, void* dz, int dz_bf16, float* dgamma,
                                float* dbeta, int accumulate, long M, int C, float* scratch, void* stream) {
    if (!bn_shape_ok(M, C)) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    if (z_bf16 && dy_bf16 && dz_bf16) return bn_bwd_launch<bf16_t, bf16_t, bf16_t>(z, dy, stats, dz, dgamma, dbeta, accumulate, M, C, scratch, st);
    if (z_bf16 && !dy_bf16 && dz_bf16) return bn_bwd_launch<bf16_t, float, bf16_t>(z, dy, stats, dz, dgamma, dbeta, accumulate, M, C, scratch, st);
    if (!z_bf16 && !dy_bf16 && !dz_bf16) return bn_bwd_launch<float, float, float>(z, dy, stats, dz, dgamma, dbeta, accumulate, M, C, scratch, st);
    return (int)hipErrorInvalidValue;
}

// ---- SyncBatchNorm backward: whmr_bn_relu_bwd cut at its reduction.
//   whmr_bn_bwd_sums   sums64 [2C] = sum g | sum g xhat over THIS rank's rows (g = dy where relu(bn(z)) > 0; stats = the GLOBAL statistics) and the
//                      LOCAL dgamma / dbeta ((+)= when accumulate)                        -> all-reduce(SUM) of sums64 over the ranks (host)
//   whmr_bn_bwd_apply  dz = a (g - sum_g / N - xhat sum_gx / N) with the summed vector and N = *count (the forward's summed row count, on the device)
// scratch: >= (BN_PARTS * 2 + 2) * C floats.
template <typename TZ, typename TDY>
static int bn_bwd_sums_launch(const void* z, const void* dy, const float* stats, float* dgamma, float* dbeta, int accumulate, long M, int C,
                              double* sums64, float* scratch, hipStream_t st) {
    const int rpb = 256 / (C >> 3);
    const int parts = (int)min((long)BN_PARTS, (M + rpb - 1) / rpb);
    const size_t lds = (size_t)2 * rpb * C * sizeof(float);
    hipLaunchKernelGGL((bn_partial_kernel<TZ, TDY, 1>), dim3(parts), dim3(256), lds, st, (const TZ*)z, (const TDY*)dy, stats, M, C, scratch);
    WHMR_CHECK_LAUNCH();
    hipLaunchKernelGGL((bn_finalize_kernel<3>), dim3((C + 15) / 16), dim3(256), 0, st, scratch, parts, M, C, (const float*)nullptr, (const bf16_t*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, 0.f, 0.f, (float*)nullptr, (float*)nullptr, (float*)nullptr, dgamma, dbeta,
                       accumulate, (float*)nullptr, sums64);
    WHMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int whmr_bn_bwd_sums(const void* z, int z_bf16, const void* dy, int dy_bf16, const float* stats, float* dgamma, float* dbeta, int accumulate,
                                long M, int C, double* sums64, float* scratch, void* stream) {
    if (!bn_shape_ok(M, C) || !sums64 || !dgamma || !dbeta) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    if (z_bf16 && dy_bf16) return bn_bwd_sums_launch<bf16_t, bf16_t>(z, dy, stats, dgamma, dbeta, accumulate, M, C, sums64, scratch, st);
    if (z_bf16 && !dy_bf16) return bn_bwd_sums_launch<bf16_t, float>(z, dy, stats, dgamma, dbeta, accumulate, M, C, sums64, scratch, st);
    if (!z_bf16 && !dy_bf16) return bn_bwd_sums_launch<float, float>(z, dy, stats, dgamma, dbeta, accumulate, M, C, sums64, scratch, st);
    return (int)hipErrorInvalidValue;
}

extern "C" int whmr_bn_bwd_apply(const void* z, int z_bf16, const void* dy, int dy_bf16, const float* stats, const double* sums64, const double* count,
                                 void* dz, int dz_bf16, long M, int C, float* scratch, void* stream) {
    if (!bn_shape_ok(M, C) || !sums64 || !count) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    float* coef = scratch + (size_t)BN_PARTS * 2 * C;
    hipLaunchKernelGGL(bn_coef_from_sums_kernel, dim3((C + 255) / 256), dim3(256), 0, st, sums64, count, C, coef);
    WHMR_CHECK_LAUNCH();
    const int rpb = 256 / (C >> 3);
    const int blocks = (int)min((long)4096, (M + rpb - 1) / rpb);
    if (z_bf16 && dy_bf16 && dz_bf16) hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, bf16_t, bf16_t>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)z, (const bf16_t*)dy, stats, coef, (bf16_t*)dz, M, C);
    else if (z_bf16 && !dy_bf16 && dz_bf16) hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, float, bf16_t>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)z, (const float*)dy, stats, coef, (bf16_t*)dz, M, C);
    else if (!z_bf16 && !dy_bf16 && !dz_bf16) hipLaunchKernelGGL((bn_bwd_apply_kernel<float, float, float>), dim3(blocks), dim3(256), 0, st, (const float*)z, (const float*)dy, stats, coef, (float*)dz, M, C);
    else return (int)hipErrorInvalidValue;
    WHMR_CHECK_LAUNCH();
    return 0;
}

// T[(ky*KW + kx)*C + c][m] = src[b, oy*S + ky - P, ox*S + kx - P, c] (0 outside), m = (b*OH + oy)*OW + ox, columns M..Mpad-1 zero.
// Block: 64 rows m x 64 channels of one tap; 16-B reads along c, transposed through LDS, 8 consecutive m per thread on the way out.
template <typename T>
__global__ __launch_bounds__(256) void im2col_t_kernel(const T* __restrict__ src, T* __restrict__ dst, int B, int IH, int IW, int C, int OH, int OW,
                                                       int KW, int S, int P, long M, long Mpad) {
    constexpr int VEC = 16 / sizeof(T);              // elements per 16-B access
    constexpr int TPRW = 64 / VEC;                   // threads per 64-channel row
    constexpr int ROWS = 256 / TPRW;                 // rows per pass
    __shared__ __attribute__((aligned(16))) T tile[64][64 + 2 * VEC / 4 + 2];
    const int t = threadIdx.x;
    const long m0 = (long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64, tap = blockIdx.z, ky = tap / KW, kx = tap % KW;
#pragma unroll
    for (int i = 0; i < 64 / ROWS; ++i) {
        const int row = t / TPRW + ROWS * i, cv = (t % TPRW) * VEC;
        const long m = m0 + row;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (m < M) {
            const int ox = (int)(m % OW), oy = (int)((m / OW) % OH), b = (int)(m / ((long)OW * OH));
            const int iy = oy * S + ky - P, ix = ox * S + kx - P;
            if (iy >= 0 && iy < IH && ix >= 0 && ix < IW) v = *(const uint4*)(src + (((long)b * IH + iy) * IW + ix) * C + c0 + cv);
        }
        const T* e = (const T*)&v;
#pragma unroll
        for (int k = 0; k < VEC; ++k) tile[row][cv + k] = e[k];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = (t >> 3) + 32 * i, r8 = (t & 7) * 8;
        if (m0 + r8 < Mpad) {
            __attribute__((aligned(16))) T o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = tile[r8 + k][c];
            T* d = dst + ((long)tap * C + c0 + c) * Mpad + m0 + r8;
#pragma unroll
            for (int k = 0; k < 8 / VEC; ++k) *(uint4*)(d + k * VEC) = *(const uint4*)(o + k * VEC);
        }
    }
}

extern "C" int whmr_im2col_t(const void* src, void* dst, int is_bf16, int B, int IH, int IW, int C, int OH, int OW, int KH, int KW, int S, int P,
                             long Mpad, void* stream) {
    const long M = (long)B * OH * OW;
    if (M <= 0 || (C & 63) || (Mpad & 7) || Mpad < M || KH <= 0 || KW <= 0) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)((Mpad + 63) / 64), C / 64, KH * KW), block(256);
    if (is_bf16) hipLaunchKernelGGL((im2col_t_kernel<bf16_t>), grid, block, 0, st, (const bf16_t*)src, (bf16_t*)dst, B, IH, IW, C, OH, OW, KW, S, P, M, Mpad);
    else hipLaunchKernelGGL((im2col_t_kernel<float>), grid, block, 0, st, (const float*)src, (float*)dst, B, IH, IW, C, OH, OW, KW, S, P, M, Mpad);
    WHMR_CHECK_LAUNCH();
    return 0;
}

// col2im (gather form, no atomics): dx[b, iy, ix, c] = sum over the (oy, ox, ky, kx) with oy*S + ky - P == iy, ox*S + kx - P == ix of
// dcol[(b, oy, ox)][(ky*KW + kx)*C + c].  The data gradient of a strided Conv2d whose column-space gradient dcol = dY . W came from the
// GEMM kernel (Tz-head 7x7 s3 conv, models/whmr.py:419: every input pixel collects <= 9 of the 49 taps).  Thread = 8 channels of a pixel.
template <typename T, typename TO>
__global__ __launch_bounds__(256) void col2im_kernel(const T* __restrict__ dcol, long ldcol, TO* __restrict__ dx, int B, int IH, int IW, int C,
                                                     int OH, int OW, int KH, int KW, int S, int P) {
    const int tpr = C >> 3, rpb = 256 / tpr;
    const int cg = threadIdx.x % tpr, rl = threadIdx.x / tpr, c = cg * 8;
    const long npix = (long)B * IH * IW;
    const long pix = (long)blockIdx.x * rpb + rl;
    if (pix >= npix) return;
    const int ix = (int)(pix % IW), iy = (int)((pix / IW) % IH), b = (int)(pix / ((long)IW * IH));
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    const int oy_lo = max(0, (iy + P - KH + 1 + S - 1) / S), oy_hi = min(OH - 1, (iy + P) / S);
    const int ox_lo = max(0, (ix + P - KW + 1 + S - 1) / S), ox_hi = min(OW - 1, (ix + P) / S);
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
        const int ky = iy + P - oy * S;
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {
            const int kx = ix + P - ox * S;
            float v[8];
            load8<T>(dcol + (((long)b * OH + oy) * OW + ox) * ldcol + (long)(ky * KW + kx) * C + c, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += v[e];
        }
    }
    store8<TO>(dx + pix * C + c, acc);
}

extern "C" int whmr_col2im(const void* dcol, int dcol_bf16, long ldcol, void* dx, int dx_bf16, int B, int IH, int IW, int C, int OH, int OW,
                           int KH, int KW, int S, int P, void* stream) {
    if (B <= 0 || (C & 7) || C > 2048 || 256 % (C >> 3) || S <= 0) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    const int rpb = 256 / (C >> 3);
    const long npix = (long)B * IH * IW;
    dim3 grid((unsigned)((npix + rpb - 1) / rpb)), block(256);
    if (dcol_bf16 && dx_bf16) hipLaunchKernelGGL((col2im_kernel<bf16_t, bf16_t>), grid, block, 0, st, (const bf16_t*)dcol, ldcol, (bf16_t*)dx, B, IH, IW, C, OH, OW, KH, KW, S, P);
    else if (dcol_bf16) hipLaunchKernelGGL((col2im_kernel<bf16_t, float>), grid, block, 0, st, (const bf16_t*)dcol, ldcol, (float*)dx, B, IH, IW, C, OH, OW, KH, KW, S, P);
    else if (!dx_bf16) hipLaunchKernelGGL((col2im_kernel<float, float>), grid, block, 0, st, (const float*)dcol, ldcol, (float*)dx, B, IH, IW, C, OH, OW, KH, KW, S, P);
    else return (int)hipErrorInvalidValue;
    WHMR_CHECK_LAUNCH();
    return 0;
}
