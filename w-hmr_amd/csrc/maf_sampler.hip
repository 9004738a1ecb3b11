// Mesh-aligned feature sampler: (projection ->) bilinear gather -> 3-layer point MLP, one fused launch per iteration.
//
// Replaces MAF_Extractor.forward / .sampling / .reduce_dim (models/maf_extractor.py:126-143,103-124,75-101):
//   p2d = projection(p3d, cam)                              (utils/geometry.py:289-307, only when p3d is given)
//   f   = grid_sample(fmap, p2d, bilinear, zeros padding, align_corners=True)     [256 channels per point]
//   y   = relu(W2 . [leaky(W1 . [leaky(W0 f + b0); f] + b1); f] + b2)             leaky slope 0.01
//   out[b, c*P + p] = y[c]   (channel-major flatten, maf_extractor.py:100)
// The feature map is addressed with explicit element strides, so both the NHWC maps produced by this library's
// deconv GEMMs (256 contiguous channels per texel: the gather is 1 KiB-coalesced) and a caller's NCHW tensor work.
// Block = 128 threads = PT points of one image; features and activations live in LDS; weights are pre-transposed
// to [in][out] on the host so every weight read is coalesced and L2-resident (66 K floats).
#include <type_traits>
#include "common.h"

#define CF 256
#define PT 8

struct whmr_maf_weights {
    const float* w0t; const float* b0;   // [256][128], [128]
    const float* w1t; const float* b1;   // [384][64],  [64]     input order: [y0 (128) | f (256)]
    const float* w2t; const float* b2;   // [320][32],  [32]     input order: [y1 (64)  | f (256)]
    // optional bf16 copies in nn.Conv1d layout [out][in] for the MFMA variant (bf16 feature maps); null = VALU kernel only
    const bf16_t* w0b; const bf16_t* w1b; const bf16_t* w2b;
};

template <typename TF>
__global__ __launch_bounds__(128) void maf_sample_kernel(const TF* __restrict__ fmap, long sb, long sc, long sy, long sx,
                                                         int H, int W, const float* __restrict__ pts2d,
                                                         const float* __restrict__ pts3d, const float* __restrict__ cam, long cam_ld,
                                                         float focal, float res_w, float res_h,
                                                         const whmr_maf_weights wts, int P, float* __restrict__ out,
                                                         long out_stride, float* __restrict__ point_feat) {
    // point-minor layouts: one ds_read_b128 fetches a channel's value for 4 points
    __shared__ __attribute__((aligned(16))) float sF[CF][PT];
    __shared__ __attribute__((aligned(16))) float sY0[128][PT];
    __shared__ __attribute__((aligned(16))) float sY1[64][PT];
    __shared__ float sXY[PT][2];
    const int tid = threadIdx.x;
    const int b = blockIdx.y, p0 = blockIdx.x * PT;
    const int np = min(PT, P - p0);

    const bool direct = !pts2d && !pts3d;      // reduce_dim() on already-sampled features [B,256,P] (sx = point stride)
    if (tid < PT) {
        float x = 0.f, y = 0.f;
        if (tid < np && !direct) {
            const int p = p0 + tid;
            if (pts3d) {
                const float s = cam[cam_ld * b], tx = cam[cam_ld * b + 1], ty = cam[cam_ld * b + 2];
                const float tz = 2 * focal / (res_h * s + 1e-9f);
                const float* q = pts3d + ((size_t)b * P + p) * 3;
                const float z = q[2] + tz;
                x = (focal * ((q[0] + tx) / z)) / (res_w / 2.f);
                y = (focal * ((q[1] + ty) / z)) / (res_h / 2.f);
            } else {
                x = pts2d[((size_t)b * P + p) * 2];
                y = pts2d[((size_t)b * P + p) * 2 + 1];
            }
        }
        sXY[tid][0] = x; sXY[tid][1] = y;
    }
    __syncthreads();

    // ---- bilinear gather (ATen grid_sampler_2d, align_corners=True, zeros padding)
    const TF* fb = fmap + (size_t)b * sb;
    for (int pp = 0; pp < np && direct; ++pp) {
        sF[tid][pp] = io<TF>::ld(fb + (size_t)tid * sc + (size_t)(p0 + pp) * sx);
        sF[tid + 128][pp] = io<TF>::ld(fb + (size_t)(tid + 128) * sc + (size_t)(p0 + pp) * sx);
    }
    for (int pp = 0; pp < np && !direct; ++pp) {
        const float ix = ((sXY[pp][0] + 1.f) / 2.f) * (float)(W - 1);
        const float iy = ((sXY[pp][1] + 1.f) / 2.f) * (float)(H - 1);
        const float fx0 = floorf(ix), fy0 = floorf(iy);
        const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
        const float wnw = ((fx0 + 1.f) - ix) * ((fy0 + 1.f) - iy), wne = (ix - fx0) * ((fy0 + 1.f) - iy);
        const float wsw = ((fx0 + 1.f) - ix) * (iy - fy0), wse = (ix - fx0) * (iy - fy0);
        const bool okx0 = (unsigned)x0 < (unsigned)W, okx1 = (unsigned)x1 < (unsigned)W;
        const bool oky0 = (unsigned)y0 < (unsigned)H, oky1 = (unsigned)y1 < (unsigned)H;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = tid + 128 * h;
            const TF* fc = fb + (size_t)c * sc;
            float v = 0.f;
            if (oky0 && okx0) v += io<TF>::ld(fc + y0 * sy + x0 * sx) * wnw;
            if (oky0 && okx1) v += io<TF>::ld(fc + y0 * sy + x1 * sx) * wne;
            if (oky1 && okx0) v += io<TF>::ld(fc + y1 * sy + x0 * sx) * wsw;
            if (oky1 && okx1) v += io<TF>::ld(fc + y1 * sy + x1 * sx) * wse;
            sF[c][pp] = v;
            if (point_feat) point_feat[((size_t)b * CF + c) * P + p0 + pp] = v;
        }
    }
    for (int pp = np; pp < PT; ++pp) { sF[tid][pp] = 0.f; sF[tid + 128][pp] = 0.f; }
    __syncthreads();

    // ---- point MLP.  acc[q] += sum_i W^T[i][o] * in[i][pbase + q]: the weights of 8 consecutive inputs are fetched as 8
    // independent (coalesced, L2-resident) loads before any FMA -- one exposed load latency per 8 inputs instead of one per
    // input -- and the activations come from LDS as 16-B / 8-B broadcast-free reads (point-minor layout).
    auto layer = [&](const float* __restrict__ wt, int ld, int o, int n_in, const float (*in)[PT], int pbase, auto& acc) {
        constexpr int NP = sizeof(acc) / sizeof(float);
        for (int i = 0; i < n_in; i += 8) {
            float w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = wt[(size_t)(i + u) * ld + o];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if constexpr (NP == 8) {
                    const float4 a = *(const float4*)&in[i + u][0], c = *(const float4*)&in[i + u][4];
                    acc[0] = fmaf(w[u], a.x, acc[0]); acc[1] = fmaf(w[u], a.y, acc[1]); acc[2] = fmaf(w[u], a.z, acc[2]); acc[3] = fmaf(w[u], a.w, acc[3]);
                    acc[4] = fmaf(w[u], c.x, acc[4]); acc[5] = fmaf(w[u], c.y, acc[5]); acc[6] = fmaf(w[u], c.z, acc[6]); acc[7] = fmaf(w[u], c.w, acc[7]);
                } else if constexpr (NP == 4) {
                    const float4 a = *(const float4*)&in[i + u][pbase];
                    acc[0] = fmaf(w[u], a.x, acc[0]); acc[1] = fmaf(w[u], a.y, acc[1]); acc[2] = fmaf(w[u], a.z, acc[2]); acc[3] = fmaf(w[u], a.w, acc[3]);
                } else {
                    const float2 a = *(const float2*)&in[i + u][pbase];
                    acc[0] = fmaf(w[u], a.x, acc[0]); acc[1] = fmaf(w[u], a.y, acc[1]);
                }
            }
        }
    };
    {   // layer 0: 256 -> 128, thread = output channel, all PT points
        float acc[PT];
        const float bv = wts.b0[tid];
#pragma unroll
        for (int pp = 0; pp < PT; ++pp) acc[pp] = bv;
        layer(wts.w0t, 128, tid, CF, sF, 0, acc);
#pragma unroll
        for (int pp = 0; pp < PT; ++pp) sY0[tid][pp] = acc[pp] > 0.f ? acc[pp] : 0.01f * acc[pp];
    }
    __syncthreads();
    {   // layer 1: [128 | 256] -> 64, thread = (output channel, half of the points)
        const int o = tid & 63, hp = (tid >> 6) * (PT / 2);
        float acc[PT / 2];
        const float bv = wts.b1[o];
#pragma unroll
        for (int q = 0; q < PT / 2; ++q) acc[q] = bv;
        layer(wts.w1t, 64, o, 128, sY0, hp, acc);
        layer(wts.w1t + 128 * 64, 64, o, CF, sF, hp, acc);
#pragma unroll
        for (int q = 0; q < PT / 2; ++q) sY1[o][hp + q] = acc[q] > 0.f ? acc[q] : 0.01f * acc[q];
    }
    __syncthreads();
    {   // layer 2: [64 | 256] -> 32, thread = (output channel, quarter of the points); ReLU; channel-major store
        const int o = tid & 31, qp = (tid >> 5) * (PT / 4);
        float acc[PT / 4];
        const float bv = wts.b2[o];
#pragma unroll
        for (int q = 0; q < PT / 4; ++q) acc[q] = bv;
        layer(wts.w2t, 32, o, 64, sY1, qp, acc);
        layer(wts.w2t + 64 * 32, 32, o, CF, sF, qp, acc);
#pragma unroll
        for (int q = 0; q < PT / 4; ++q) {
            const int p = p0 + qp + q;
            if (p < P) out[(size_t)b * out_stride + (size_t)o * P + p] = fmaxf(acc[q], 0.f);
        }
    }
}

// ---- MFMA variant for bf16 NHWC feature maps (the perf numerics mode): 32 points per workgroup (points of consecutive images
// are packed: global point g = b * P + p), gather -> F[32][256] bf16 in LDS, then the three layers on v_mfma_f32_32x32x16_bf16
// with the operands swapped (rows = output channels, columns = points) so that every lane owns one point: the activations go
// back to LDS as 8-B row pieces and the final channel-major store is coalesced over points.  fp32 accumulation, bf16 weights.
__device__ __forceinline__ uint32_t maf_f_addr(int row, int chunk, int nchunk_mask) { return row * ((nchunk_mask + 1) * 16) + ((chunk ^ (row & nchunk_mask)) << 4); }

__global__ __launch_bounds__(256) void maf_sample_mfma_kernel(const bf16_t* __restrict__ fmap, long sb, long sy, long sx, int H, int W,
                                                              const float* __restrict__ pts2d, const float* __restrict__ pts3d,
                                                              const float* __restrict__ cam, long cam_ld, float focal, float res_w, float res_h,
                                                              const whmr_maf_weights wts, int B, int P, float* __restrict__ out, long out_stride) {
    __shared__ __attribute__((aligned(16))) char sFb[32 * 512];       // F  [32 points][256 ch] bf16, 16-B chunks XOR-swizzled by the row
    __shared__ __attribute__((aligned(16))) char sY0b[32 * 256];      // Y0 [32][128]
    __shared__ __attribute__((aligned(16))) char sY1b[32 * 128];      // Y1 [32][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const long g0 = (long)blockIdx.x * 32, total = (long)B * P;
    // Weight fragments are REQUESTED before they are needed (round 3): the layers are chains of MFMAs whose weight operand came from global
    // memory four at a time, i.e. 4 + 6 + 5 dependent L2 round trips behind the gather -- most of the launch.  Layer 0's 16 fragments are in
    // flight under the gather, layer 1's 24 under layer 0, layer 2's 20 under layer 1.
    auto w_frag = [&](const bf16_t* w, int ld, int n0, int kk) { return *(const bf16x8_t*)(w + (size_t)(n0 + l31) * ld + kk * 16 + hi * 8); };
    bf16x8_t w0f[16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) w0f[kk] = w_frag(wts.w0b, 256, wave * 32, kk);
    // ---- gather: 8 threads per point, 32 channels (4 x 16 B) each
    {
        const int r = tid >> 3, part = tid & 7;
        const long g = g0 + r;
        float acc[32];
#pragma unroll
        for (int e = 0; e < 32; ++e) acc[e] = 0.f;
        if (g < total) {
            const int b = (int)(g / P), p = (int)(g - (long)b * P);
            float x, y;
            if (pts3d) {
                const float s = cam[cam_ld * b], tx = cam[cam_ld * b + 1], ty = cam[cam_ld * b + 2];
                const float tz = 2 * focal / (res_h * s + 1e-9f);
                const float* q = pts3d + ((size_t)b * P + p) * 3;
                const float z = q[2] + tz;
                x = (focal * ((q[0] + tx) / z)) / (res_w / 2.f);
                y = (focal * ((q[1] + ty) / z)) / (res_h / 2.f);
            } else {
                x = pts2d[((size_t)b * P + p) * 2];
                y = pts2d[((size_t)b * P + p) * 2 + 1];
            }
            const float ix = ((x + 1.f) / 2.f) * (float)(W - 1), iy = ((y + 1.f) / 2.f) * (float)(H - 1);
            const float fx0 = floorf(ix), fy0 = floorf(iy);
            const int x0 = (int)fx0, y0 = (int)fy0;
            const float wx1 = ix - fx0, wx0 = (fx0 + 1.f) - ix, wy1 = iy - fy0, wy0 = (fy0 + 1.f) - iy;
            const bf16_t* fb = fmap + (size_t)b * sb + part * 32;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int xx = x0 + (t & 1), yy = y0 + (t >> 1);
                const float wgt = ((t & 1) ? wx1 : wx0) * ((t >> 1) ? wy1 : wy0);
                if ((unsigned)xx < (unsigned)W && (unsigned)yy < (unsigned)H) {
                    const uint4* src = (const uint4*)(fb + (size_t)yy * sy + (size_t)xx * sx);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const uint4 v = src[c];
                        const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc[c * 8 + 2 * e] = fmaf(__uint_as_float(w4[e] << 16), wgt, acc[c * 8 + 2 * e]);
                            acc[c * 8 + 2 * e + 1] = fmaf(__uint_as_float(w4[e] & 0xffff0000u), wgt, acc[c * 8 + 2 * e + 1]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            *(uint4*)(sFb + maf_f_addr(r, part * 4 + c, 31)) = make_uint4(pack_bf16x2(acc[c * 8], acc[c * 8 + 1]), pack_bf16x2(acc[c * 8 + 2], acc[c * 8 + 3]),
                                                                           pack_bf16x2(acc[c * 8 + 4], acc[c * 8 + 5]), pack_bf16x2(acc[c * 8 + 6], acc[c * 8 + 7]));
    }
    __syncthreads();
    auto f_frag = [&](const char* base, int kk, int mask) { return *(const bf16x8_t*)(base + maf_f_addr(l31, kk * 2 + hi, mask)); };
    bf16x8_t w1f[24];
    if (wave < 2) {
#pragma unroll
        for (int kk = 0; kk < 24; ++kk) w1f[kk] = w_frag(wts.w1b, 384, wave * 32, kk);
    }
    // rows (regs) = output channel n0 + (r&3) + 8(r>>2) + 4hi, column (lane) = point l31
    auto store_act = [&](char* dst, int mask, int n0, const f32x16_t& a, const float* bias) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = a[4 * q + e] + bias[n0 + 8 * q + 4 * hi + e];
                v[e] = t > 0.f ? t : 0.01f * t;                                  // LeakyReLU(0.01)
            }
            const int col = n0 + 8 * q + 4 * hi;                                 // 4 consecutive channels = 8 B inside one 16-B chunk
            *(uint2*)(dst + maf_f_addr(l31, col >> 3, mask) + (col & 7) * 2) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        }
    };
    {   // layer 0: 256 -> 128; wave = one 32-channel tile
        f32x16_t a;
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = 0.f;
        // two accumulators per layer: a chain of dependent MFMAs waits the full pipeline latency per step (16 / 24 / 20 steps here)
        f32x16_t a2;
#pragma unroll
        for (int r = 0; r < 16; ++r) a2[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) {
            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0f[kk], f_frag(sFb, kk, 31), a, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0f[kk + 1], f_frag(sFb, kk + 1, 31), a2, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] += a2[r];
        store_act(sY0b, 15, wave * 32, a, wts.b0);
    }
    bf16x8_t w2f[20];
    if (wave == 0) {
#pragma unroll
        for (int kk = 0; kk < 20; ++kk) w2f[kk] = w_frag(wts.w2b, 320, 0, kk);
    }
    __syncthreads();
    if (wave < 2) {   // layer 1: [Y0 (128) | F (256)] -> 64
        f32x16_t a;
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = 0.f;
        f32x16_t a2;
#pragma unroll
        for (int r = 0; r < 16; ++r) a2[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 8; kk += 2) {
            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1f[kk], f_frag(sY0b, kk, 15), a, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1f[kk + 1], f_frag(sY0b, kk + 1, 15), a2, 0, 0, 0);
        }
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) {
            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1f[8 + kk], f_frag(sFb, kk, 31), a, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1f[9 + kk], f_frag(sFb, kk + 1, 31), a2, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] += a2[r];
        store_act(sY1b, 7, wave * 32, a, wts.b1);
    }
    __syncthreads();
    if (wave == 0) {  // layer 2: [Y1 (64) | F (256)] -> 32, ReLU, channel-major store out[b][o * P + p]
        f32x16_t a;
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = 0.f;
        f32x16_t a2;
#pragma unroll
        for (int r = 0; r < 16; ++r) a2[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; kk += 2) {
            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2f[kk], f_frag(sY1b, kk, 7), a, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2f[kk + 1], f_frag(sY1b, kk + 1, 7), a2, 0, 0, 0);
        }
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) {
            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2f[4 + kk], f_frag(sFb, kk, 31), a, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2f[5 + kk], f_frag(sFb, kk + 1, 31), a2, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] += a2[r];
        const long g = g0 + l31;
        if (g < total) {
            const int b = (int)(g / P), p = (int)(g - (long)b * P);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = (r & 3) + 8 * (r >> 2) + 4 * hi;
                out[(size_t)b * out_stride + (size_t)o * P + p] = fmaxf(a[r] + wts.b2[o], 0.f);
            }
        }
    }
}

// fmap_bf16: element type of the feature map.  pts2d XOR (pts3d, cam) selects the sampling points; with neither, fmap is
// an already-sampled [B,256,P] feature tensor (strides sb, sc, sx) and only the MLP runs (MAF_Extractor.reduce_dim).
// out row b starts at out + b*out_stride (>= 32*P), so the result can land inside the regressor's input buffer;
// cam row b starts at cam + b*cam_ld (a column slice of the regressor state is fine).
extern "C" int whmr_maf_sample(const void* fmap, int fmap_bf16, long sb, long sc, long sy, long sx, int H, int W,
                               const float* pts2d, const float* pts3d, const float* cam, long cam_ld, float focal, float res_w,
                               float res_h, const whmr_maf_weights* w, int B, int P, float* out, long out_stride,
                               float* point_feat, void* stream) {
    if (B <= 0 || P <= 0 || (pts2d && pts3d) || (pts3d && !cam) || out_stride < 32L * P) return (int)hipErrorInvalidValue;
    dim3 grid((P + PT - 1) / PT, B), block(128);
    hipStream_t st = (hipStream_t)stream;
    if (fmap_bf16 && sc == 1 && w->w0b && w->w1b && w->w2b && (pts2d || pts3d) && !point_feat && !((sb | sy | sx) & 7) && !((uintptr_t)fmap & 15)) {
        const long tiles = ((long)B * P + 31) / 32;
        hipLaunchKernelGGL(maf_sample_mfma_kernel, dim3((unsigned)tiles), dim3(256), 0, st, (const bf16_t*)fmap, sb, sy, sx, H, W, pts2d, pts3d, cam,
                           cam_ld, focal, res_w, res_h, *w, B, P, out, out_stride);
        WHMR_CHECK_LAUNCH();
        return 0;
    }
    if (fmap_bf16) hipLaunchKernelGGL(maf_sample_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)fmap, sb, sc, sy, sx, H, W,
                                      pts2d, pts3d, cam, cam_ld, focal, res_w, res_h, *w, P, out, out_stride, point_feat);
    else hipLaunchKernelGGL(maf_sample_kernel<float>, grid, block, 0, st, (const float*)fmap, sb, sc, sy, sx, H, W, pts2d,
                            pts3d, cam, cam_ld, focal, res_w, res_h, *w, P, out, out_stride, point_feat);
    WHMR_CHECK_LAUNCH();
    return 0;
}

// Tail of the Tz head (whmr.py:574-577): tokens [B,5,216] -> transpose/AvgPool1d(5) = mean over the 5 tokens ->
// Linear(216,12) -> Linear(12,1) -> BatchNorm1d(1) (eval) -> sigmoid -> * 10.
__global__ __launch_bounds__(64) void tz_tail_kernel(const float* __restrict__ tok, int T, int D, const float* __restrict__ w0,
                                                     const float* __restrict__ b0, int Hd, const float* __restrict__ w1,
                                                     const float* __restrict__ b1, const float* __restrict__ bn /*w,b,mean,var*/,
                                                     float bn_eps, float* __restrict__ tz) {
    __shared__ float sm[1024];
    __shared__ float sh[64];
    const int b = blockIdx.x, lane = threadIdx.x;
    for (int d = lane; d < D; d += 64) {
        float s = 0.f;
        for (int t = 0; t < T; ++t) s += tok[((size_t)b * T + t) * D + d];
        sm[d] = s / (float)T;
    }
    __syncthreads();
    if (lane < Hd) {
        float a = 0.f;
        for (int d = 0; d < D; ++d) a = fmaf(sm[d], w0[lane * D + d], a);
        sh[lane] = a + b0[lane];
    }
    __syncthreads();
    if (lane == 0) {
        float a = 0.f;
        for (int j = 0; j < Hd; ++j) a = fmaf(sh[j], w1[j], a);
        a += b1[0];
        a = (a - bn[2]) / sqrtf(bn[3] + bn_eps) * bn[0] + bn[1];
        tz[b] = 10.0f * (1.0f / (1.0f + expf(-a)));
    }
}

extern "C" int whmr_tz_tail(const float* tok, int B, int T, int D, const float* w0, const float* b0, int Hd, const float* w1,
                            const float* b1, const float* bn4, float bn_eps, float* tz, void* stream) {
    if (B <= 0 || D > 1024 || Hd > 64) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(tz_tail_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, tok, T, D, w0, b0, Hd, w1, b1, bn4, bn_eps, tz);
    WHMR_CHECK_LAUNCH();
    return 0;
}


// =====================================================================================================================
// Backward of the sampler (training; reference: autograd of MAF_Extractor.sampling / .reduce_dim, models/maf_extractor.py:75-124,
// driven by core/trainer.py:410-470).  The sample points carry no gradient: the reference detaches markers / pred_cam before the
// call (models/whmr.py:586-592) and the iteration-0 grid is a constant buffer.  What flows back:
//   d(fmap)  : the bilinear weights times d(f), scatter-added into an fp32 gradient map (hardware fp32 atomics: two points may
//              share a texel; the map is zero-filled by the caller)
//   d(MLP weights): the kernel recomputes the activations of its 8 points and writes, point-minor (= K-contiguous for the GEMM),
//              XT = [y0 (128) ; f (256) ; y1 (64)] x [B*P] and DT = [d_pre0 (128) ; d_pre1 (64) ; d_pre2 (32)] x [B*P];
//              dW_l = DT_l . XT_l^T then runs on whmr_gemm_f32 (K = B*P), the bias gradients are DT's row sums.
// wts holds the transposed [in][out] weights of the forward; w0 / w1 / w2 are the Conv1d-layout [out][in] matrices.
// bf16 gradient maps: two channels per 32-bit word, added with a compare-and-swap loop (gfx950 has no packed-bf16 atomic add in HIP's
// portable surface); texels shared by two points are rare, so the loop almost always runs once.
__device__ __forceinline__ void atomic_add_bf16x2(uint32_t* addr, float lo, float hi) {
    uint32_t old = *addr, assumed;
    do {
        assumed = old;
        const float a = __uint_as_float(assumed << 16) + lo, b = __uint_as_float(assumed & 0xffff0000u) + hi;
        old = atomicCAS(addr, assumed, pack_bf16x2(a, b));
    } while (old != assumed);
}

// TG = maf_compact: d(f) is not scattered; row b*P + p of d_fmap (fp32, MAF_ROW floats) takes the 256 channel gradients of point p, its four texel offsets
// (y*W + x as int bits, -1 = outside) and its four bilinear weights -- whmr_maf_scatter adds the rows to a gradient map later, on any stream.
struct maf_compact { float v; };
#define MAF_ROW 264

template <typename TF, typename TG>
__global__ __launch_bounds__(256) void maf_sample_bwd_kernel(const TF* __restrict__ fmap, long sb, long sc, long sy, long sx, int H, int W,
                                                             const float* __restrict__ pts2d, const float* __restrict__ pts3d,
                                                             const float* __restrict__ cam, long cam_ld, float focal, float res_w, float res_h,
                                                             const whmr_maf_weights wts, const float* __restrict__ w0, const float* __restrict__ w1,
                                                             const float* __restrict__ w2, int P, const float* __restrict__ d_out, long dout_stride,
                                                             TG* __restrict__ d_fmap, long gsb, long gsc, long gsy, long gsx,
                                                             float* __restrict__ XT, float* __restrict__ DT, long ldt) {
    __shared__ float sF[CF][PT], sY0[128][PT], sY1[64][PT];
    __shared__ float sD0[128][PT], sD1[64][PT], sD2[32][PT], sDF[CF][PT];
    __shared__ float sWgt[PT][4];
    __shared__ int sIdx[PT][4];                // texel offsets (y*W + x) or -1
    const int tid = threadIdx.x;
    const int b = blockIdx.y, p0 = blockIdx.x * PT;
    const int np = min(PT, P - p0);
    if (tid < PT) {
        float wq[4] = {0.f, 0.f, 0.f, 0.f};
        int iq[4] = {-1, -1, -1, -1};
        if (tid < np) {
            const int p = p0 + tid;
            float x, y;
            if (pts3d) {
                const float s = cam[cam_ld * b], tx = cam[cam_ld * b + 1], ty = cam[cam_ld * b + 2];
                const float tz = 2 * focal / (res_h * s + 1e-9f);
                const float* q = pts3d + ((size_t)b * P + p) * 3;
                const float z = q[2] + tz;
                x = (focal * ((q[0] + tx) / z)) / (res_w / 2.f);
                y = (focal * ((q[1] + ty) / z)) / (res_h / 2.f);
            } else {
                x = pts2d[((size_t)b * P + p) * 2];
                y = pts2d[((size_t)b * P + p) * 2 + 1];
            }
            const float ix = ((x + 1.f) / 2.f) * (float)(W - 1), iy = ((y + 1.f) / 2.f) * (float)(H - 1);
            const float fx0 = floorf(ix), fy0 = floorf(iy);
            const int x0 = (int)fx0, y0 = (int)fy0;
            const float wx1 = ix - fx0, wx0 = (fx0 + 1.f) - ix, wy1 = iy - fy0, wy0 = (fy0 + 1.f) - iy;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int xx = x0 + (t & 1), yy = y0 + (t >> 1);
                if ((unsigned)xx < (unsigned)W && (unsigned)yy < (unsigned)H) { iq[t] = yy * W + xx; wq[t] = ((t & 1) ? wx1 : wx0) * ((t >> 1) ? wy1 : wy0); }
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) { sWgt[tid][t] = wq[t]; sIdx[tid][t] = iq[t]; }
    }
    __syncthreads();
    {   // gather: thread = channel
        const TF* fc = fmap + (size_t)b * sb + (size_t)tid * sc;
        for (int pp = 0; pp < PT; ++pp) {
            float v = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int id = sIdx[pp][t];
                if (id >= 0) v += io<TF>::ld(fc + (size_t)(id / W) * sy + (size_t)(id % W) * sx) * sWgt[pp][t];
            }
            sF[tid][pp] = v;
        }
    }
    __syncthreads();
    // ---- forward recompute (same summation order as maf_sample_kernel: bias first, inputs in index order)
    if (tid < 128) {
        float acc[PT];
#pragma unroll
        for (int pp = 0; pp < PT; ++pp) acc[pp] = wts.b0[tid];
#pragma unroll 8
        for (int i = 0; i < CF; ++i) {
            const float w = wts.w0t[(size_t)i * 128 + tid];
#pragma unroll
            for (int pp = 0; pp < PT; ++pp) acc[pp] = fmaf(w, sF[i][pp], acc[pp]);
        }
#pragma unroll
        for (int pp = 0; pp < PT; ++pp) sY0[tid][pp] = acc[pp] > 0.f ? acc[pp] : 0.01f * acc[pp];
    }
    __syncthreads();
    {
        const int o = tid & 63, hp = (tid >> 6) * 2;
        float acc[2] = {wts.b1[o], wts.b1[o]};
#pragma unroll 8
        for (int i = 0; i < 128; ++i) { const float w = wts.w1t[(size_t)i * 64 + o]; acc[0] = fmaf(w, sY0[i][hp], acc[0]); acc[1] = fmaf(w, sY0[i][hp + 1], acc[1]); }
#pragma unroll 8
        for (int i = 0; i < CF; ++i) { const float w = wts.w1t[(size_t)(128 + i) * 64 + o]; acc[0] = fmaf(w, sF[i][hp], acc[0]); acc[1] = fmaf(w, sF[i][hp + 1], acc[1]); }
        sY1[o][hp] = acc[0] > 0.f ? acc[0] : 0.01f * acc[0];
        sY1[o][hp + 1] = acc[1] > 0.f ? acc[1] : 0.01f * acc[1];
    }
    __syncthreads();
    {   // layer 2 + ReLU gate + upstream gradient: thread = (output channel, point)
        const int o = tid & 31, pp = tid >> 5;
        float acc = wts.b2[o];
#pragma unroll 8
        for (int i = 0; i < 64; ++i) acc = fmaf(wts.w2t[(size_t)i * 32 + o], sY1[i][pp], acc);
#pragma unroll 8
        for (int i = 0; i < CF; ++i) acc = fmaf(wts.w2t[(size_t)(64 + i) * 32 + o], sF[i][pp], acc);
        const int p = p0 + pp;
        sD2[o][pp] = (p < P && acc > 0.f) ? d_out[(size_t)b * dout_stride + (size_t)o * P + p] : 0.f;
    }
    __syncthreads();
    // ---- d(inputs of layer 2) = W2^T d2: inputs [y1 (64) | f (256)]
    for (int i = tid; i < 320; i += 256) {
        float acc[PT];
#pragma unroll
        for (int pp = 0; pp < PT; ++pp) acc[pp] = 0.f;
#pragma unroll 8
        for (int o = 0; o < 32; ++o) {
            const float w = w2[(size_t)o * 320 + i];
#pragma unroll
            for (int pp = 0; pp < PT; ++pp) acc[pp] = fmaf(w, sD2[o][pp], acc[pp]);
        }
        if (i < 64) {
#pragma unroll
            for (int pp = 0; pp < PT; ++pp) sD1[i][pp] = sY1[i][pp] > 0.f ? acc[pp] : 0.01f * acc[pp];
        } else {
#pragma unroll
            for (int pp = 0; pp < PT; ++pp) sDF[i - 64][pp] = acc[pp];
        }
    }
    __syncthreads();
    // ---- layer 1: inputs [y0 (128) | f (256)]
    for (int i = tid; i < 384; i += 256) {
        float acc[PT];
#pragma unroll
        for (int pp = 0; pp < PT; ++pp) acc[pp] = 0.f;
#pragma unroll 8
        for (int o = 0; o < 64; ++o) {
            const float w = w1[(size_t)o * 384 + i];
#pragma unroll
            for (int pp = 0; pp < PT; ++pp) acc[pp] = fmaf(w, sD1[o][pp], acc[pp]);
        }
        if (i < 128) {
#pragma unroll
            for (int pp = 0; pp < PT; ++pp) sD0[i][pp] = sY0[i][pp] > 0.f ? acc[pp] : 0.01f * acc[pp];
        } else {
#pragma unroll
            for (int pp = 0; pp < PT; ++pp) sDF[i - 128][pp] += acc[pp];
        }
    }
    __syncthreads();
    {   // layer 0: inputs f (256); thread = channel
        float acc[PT];
#pragma unroll
        for (int pp = 0; pp < PT; ++pp) acc[pp] = 0.f;
#pragma unroll 8
        for (int o = 0; o < 128; ++o) {
            const float w = w0[(size_t)o * CF + tid];
#pragma unroll
            for (int pp = 0; pp < PT; ++pp) acc[pp] = fmaf(w, sD0[o][pp], acc[pp]);
        }
        // scatter d(f) into the gradient map
        if constexpr (std::is_same<TG, maf_compact>::value) {
            float* rows = (float*)d_fmap + ((size_t)b * P + p0) * MAF_ROW;
            for (int pp = 0; pp < np; ++pp) rows[(size_t)pp * MAF_ROW + tid] = sDF[tid][pp] + acc[pp];
            if (tid < np * 8) {
                const int pp = tid >> 3, t = tid & 7;
                rows[(size_t)pp * MAF_ROW + CF + t] = t < 4 ? __int_as_float(sIdx[pp][t]) : sWgt[pp][t - 4];
            }
        } else if constexpr (std::is_same<TG, float>::value) {
            float* gc = d_fmap ? d_fmap + (size_t)b * gsb + (size_t)tid * gsc : nullptr;
            for (int pp = 0; pp < np; ++pp) {
                const float df = sDF[tid][pp] + acc[pp];
                if (gc) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int id = sIdx[pp][t];
                        if (id >= 0) unsafeAtomicAdd(gc + (size_t)(id / W) * gsy + (size_t)(id % W) * gsx, df * sWgt[pp][t]);
                    }
                }
            }
        } else {        // bf16 channels-last map (gsc == 1): even threads add the channel pair (tid, tid + 1) as one 32-bit word
#pragma unroll
            for (int pp = 0; pp < PT; ++pp) sDF[tid][pp] += acc[pp];
            __syncthreads();
            if (d_fmap && !(tid & 1)) {
                bf16_t* gc = d_fmap + (size_t)b * gsb + tid;
                for (int pp = 0; pp < np; ++pp) {
                    const float d0 = sDF[tid][pp], d1 = sDF[tid + 1][pp];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int id = sIdx[pp][t];
                        if (id >= 0) atomic_add_bf16x2((uint32_t*)(gc + (size_t)(id / W) * gsy + (size_t)(id % W) * gsx), d0 * sWgt[pp][t], d1 * sWgt[pp][t]);
                    }
                }
            }
        }
    }
    // ---- operands of the weight-gradient GEMMs, point-minor
    const long col0 = (long)b * P + p0;
    for (int e = tid; e < 448 * PT; e += 256) {
        const int r = e / PT, pp = e % PT;
        if (pp < np) XT[(size_t)r * ldt + col0 + pp] = r < 128 ? sY0[r][pp] : (r < 384 ? sF[r - 128][pp] : sY1[r - 384][pp]);
    }
    for (int e = tid; e < 224 * PT; e += 256) {
        const int r = e / PT, pp = e % PT;
        if (pp < np) DT[(size_t)r * ldt + col0 + pp] = r < 128 ? sD0[r][pp] : (r < 192 ? sD1[r - 128][pp] : sD2[r - 192][pp]);
    }
}

// d_out [B, >= 32*P rows of stride dout_stride]; d_fmap (nullable) fp32 with element strides (gsb, gsc, gsy, gsx), accumulated into
// (d_fmap_bf16 = 2: d_fmap is the compact [B*P][MAF_ROW] fp32 record of whmr_maf_scatter instead, strides unused);
// XT [448, ldt], DT [224, ldt] with ldt >= B*P (the caller zero-fills the padding columns once).
extern "C" int whmr_maf_sample_bwd(const void* fmap, int fmap_bf16, long sb, long sc, long sy, long sx, int H, int W, const float* pts2d,
                                   const float* pts3d, const float* cam, long cam_ld, float focal, float res_w, float res_h,
                                   const whmr_maf_weights* w, const float* w0, const float* w1, const float* w2, int B, int P,
                                   const float* d_out, long dout_stride, void* d_fmap, int d_fmap_bf16, long gsb, long gsc, long gsy, long gsx,
                                   float* XT, float* DT, long ldt, void* stream) {
    if (B <= 0 || P <= 0 || (!pts2d == !pts3d) || (pts3d && !cam) || dout_stride < 32L * P || ldt < (long)B * P) return (int)hipErrorInvalidValue;
    if (d_fmap_bf16 == 2 && !d_fmap) return (int)hipErrorInvalidValue;
    if (d_fmap && d_fmap_bf16 == 1 && (gsc != 1 || ((gsb | gsy | gsx) & 1) || ((uintptr_t)d_fmap & 3))) return (int)hipErrorInvalidValue;
    dim3 grid((P + PT - 1) / PT, B), block(256);
    hipStream_t st = (hipStream_t)stream;
#define MAF_BWD(TF, TG) hipLaunchKernelGGL((maf_sample_bwd_kernel<TF, TG>), grid, block, 0, st, (const TF*)fmap, sb, sc, sy, sx, H, W, pts2d, pts3d, cam, \
                                           cam_ld, focal, res_w, res_h, *w, w0, w1, w2, P, d_out, dout_stride, (TG*)d_fmap, gsb, gsc, gsy, gsx, XT, DT, ldt)
    if (d_fmap_bf16 == 2) { if (fmap_bf16) MAF_BWD(bf16_t, maf_compact); else MAF_BWD(float, maf_compact); }
    else if (fmap_bf16 && d_fmap_bf16) MAF_BWD(bf16_t, bf16_t);
    else if (fmap_bf16) MAF_BWD(bf16_t, float);
    else if (!d_fmap_bf16) MAF_BWD(float, float);
    else return (int)hipErrorInvalidValue;
#undef MAF_BWD
    WHMR_CHECK_LAUNCH();
    return 0;
}

// The deferred half of the sampler's backward: rows of the compact record (maf_compact above) added to a gradient map that other consumers of the
// feature map have already written -- 4 texels x 256 channels per point instead of a dense zero-filled map plus a full-size add.
template <typename TG>
__global__ __launch_bounds__(256) void maf_scatter_kernel(const float* __restrict__ rec, long n, int P, int W, TG* __restrict__ d_fmap, long gsb, long gsc,
                                                          long gsy, long gsx) {
    constexpr bool BF = std::is_same<TG, bf16_t>::value;
    const int tid = threadIdx.x;
    const long pt = BF ? (long)blockIdx.x * 2 + (tid >> 7) : (long)blockIdx.x;
    if (pt >= n) return;
    const float* row = rec + (size_t)pt * MAF_ROW;
    const int b = (int)(pt / P);
    int id[4];
    float wg[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { id[t] = __float_as_int(row[CF + t]); wg[t] = row[CF + 4 + t]; }
    if constexpr (BF) {
        const int c = (tid & 127) * 2;
        const float2 d = *(const float2*)(row + c);
        bf16_t* gc = d_fmap + (size_t)b * gsb + c;
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (id[t] >= 0) atomic_add_bf16x2((uint32_t*)(gc + (size_t)(id[t] / W) * gsy + (size_t)(id[t] % W) * gsx), d.x * wg[t], d.y * wg[t]);
    } else {
        const float d = row[tid];
        float* gc = d_fmap + (size_t)b * gsb + (size_t)tid * gsc;
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (id[t] >= 0) unsafeAtomicAdd(gc + (size_t)(id[t] / W) * gsy + (size_t)(id[t] % W) * gsx, d * wg[t]);
    }
}

extern "C" int whmr_maf_scatter(const float* rec, int B, int P, int W, void* d_fmap, int d_fmap_bf16, long gsb, long gsc, long gsy, long gsx, void* stream) {
    if (!rec || !d_fmap || B <= 0 || P <= 0 || W <= 0 || (d_fmap_bf16 != 0 && d_fmap_bf16 != 1)) return (int)hipErrorInvalidValue;
    if (d_fmap_bf16 && (gsc != 1 || ((gsb | gsy | gsx) & 1) || ((uintptr_t)d_fmap & 3))) return (int)hipErrorInvalidValue;
    const long n = (long)B * P;
    hipStream_t st = (hipStream_t)stream;
    if (d_fmap_bf16) hipLaunchKernelGGL((maf_scatter_kernel<bf16_t>), dim3((unsigned)((n + 1) / 2)), dim3(256), 0, st, rec, n, P, W, (bf16_t*)d_fmap, gsb, gsc, gsy, gsx);
    else hipLaunchKernelGGL((maf_scatter_kernel<float>), dim3((unsigned)n), dim3(256), 0, st, rec, n, P, W, (float*)d_fmap, gsb, gsc, gsy, gsx);
    WHMR_CHECK_LAUNCH();
    return 0;
}
