// Backward-pass helpers for the ViT backbone (north_star "forward/backward"; reference: autograd of
// models/ViTPose/mmpose/models/backbones/vit.py:61-140,313-332 as driven by core/trainer.py:410-470).
// The matrix products of the backward pass (dX = dY.W, dW = dY^T.X) run on the same GEMM kernels as the forward
// (gemm_bf16*.hip / gemm_f32.hip); this file holds the memory-bound pieces around them:
//   whmr_transpose_cast   [R,C] -> [C,Rpad] with dtype conversion (operands of the dW products; weights W -> W^T)
//   whmr_colsum           bias gradients  db[c] = sum_r dY[r,c]         (deterministic two-stage)
//   whmr_layernorm_bwd    dx (+ residual-stream gradient), dgamma, dbeta (deterministic two-stage)
//   whmr_gelu_bwd         d_pre = d_hid * (Phi(pre) + pre * phi(pre))    (exact erf form, like nn.GELU)
#include "common.h"

template <typename TIN, typename TOUT>
__global__ __launch_bounds__(256) void transpose_cast_kernel(const TIN* __restrict__ src, long ld_src, TOUT* __restrict__ dst, long ld_dst,
                                                             int R, int C, int Rpad) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        tile[ty + 8 * i][tx] = (r < R && c < C) ? io<TIN>::ld(src + (size_t)r * ld_src + c) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (c < C && r < Rpad) io<TOUT>::st(dst + (size_t)c * ld_dst + r, tile[tx][ty + 8 * i]);
    }
}

// bf16 -> bf16 fast path (the operands of the dW products): 64x64 tiles, 16-B global loads and stores.
// Requires R, C, Rpad, ld_src, ld_dst multiples of 8 and 16-B aligned bases (checked by the launcher).
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ src, long ld_src, bf16_t* __restrict__ dst, long ld_dst,
                                                             int R, int C, int Rpad) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[64][66];
    const int t = threadIdx.x;
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (t >> 3) + 32 * i, c8 = (t & 7) * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r0 + row < R && c0 + c8 < C) v = *(const uint4*)(src + (size_t)(r0 + row) * ld_src + c0 + c8);
        uint32_t* d = (uint32_t*)&tile[row][c8];
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = (t >> 3) + 32 * i, r8 = (t & 7) * 8;
        if (c0 + c < C && r0 + r8 < Rpad) {
            uint32_t w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] = (uint32_t)tile[r8 + 2 * k][c] | ((uint32_t)tile[r8 + 2 * k + 1][c] << 16);
            *(uint4*)(dst + (size_t)(c0 + c) * ld_dst + r0 + r8) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}

// Same transpose, plus the column sums of the source tile (bias gradient db[c] = sum_r dY[r, c]: the transpose of dY for the dW product
// reads every element of dY anyway): partial[blockIdx.x][c] over the tile's 64 rows, finished by colsum_final_kernel in a fixed order.
__global__ __launch_bounds__(256) void transpose_bf16_colsum_kernel(const bf16_t* __restrict__ src, long ld_src, bf16_t* __restrict__ dst, long ld_dst,
                                                                    int R, int C, int Rpad, float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[64][66];
    __shared__ float csum[4][64];
    const int t = threadIdx.x;
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (t >> 3) + 32 * i, c8 = (t & 7) * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r0 + row < R && c0 + c8 < C) v = *(const uint4*)(src + (size_t)(r0 + row) * ld_src + c0 + c8);
        uint32_t* d = (uint32_t*)&tile[row][c8];
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    {   // column sums: thread = (column, quarter of the rows)
        const int c = t & 63, q = t >> 6;
        float a = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) a += bf16_to_f32(tile[q * 16 + r][c]);
        csum[q][c] = a;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = (t >> 3) + 32 * i, r8 = (t & 7) * 8;
        if (c0 + c < C && r0 + r8 < Rpad) {
            uint32_t w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] = (uint32_t)tile[r8 + 2 * k][c] | ((uint32_t)tile[r8 + 2 * k + 1][c] << 16);
            *(uint4*)(dst + (size_t)(c0 + c) * ld_dst + r0 + r8) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    __syncthreads();
    if (t < 64 && c0 + t < C && r0 < R)
        partial[(size_t)blockIdx.x * C + c0 + t] = (csum[0][t] + csum[1][t]) + (csum[2][t] + csum[3][t]);
}

extern "C" int whmr_transpose_cast(const void* src, int src_bf16, long ld_src, void* dst, int dst_bf16, long ld_dst, int R, int C, int Rpad,
                                   void* stream) {
    if (R <= 0 || C <= 0 || Rpad < R || ld_dst < Rpad || ld_src < C) return (int)hipErrorInvalidValue;
    dim3 grid((Rpad + 31) / 32, (C + 31) / 32), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (src_bf16 && dst_bf16 && !((R | C | Rpad | ld_src | ld_dst) & 7) && !(((uintptr_t)src | (uintptr_t)dst) & 15)) {
        hipLaunchKernelGGL(transpose_bf16_kernel, dim3((Rpad + 63) / 64, (C + 63) / 64), block, 0, st, (const bf16_t*)src, ld_src, (bf16_t*)dst,
                           ld_dst, R, C, Rpad);
        WHMR_CHECK_LAUNCH();
        return 0;
    }
    if (src_bf16 && dst_bf16) hipLaunchKernelGGL((transpose_cast_kernel<bf16_t, bf16_t>), grid, block, 0, st, (const bf16_t*)src, ld_src, (bf16_t*)dst, ld_dst, R, C, Rpad);
    else if (src_bf16) hipLaunchKernelGGL((transpose_cast_kernel<bf16_t, float>), grid, block, 0, st, (const bf16_t*)src, ld_src, (float*)dst, ld_dst, R, C, Rpad);
    else if (dst_bf16) hipLaunchKernelGGL((transpose_cast_kernel<float, bf16_t>), grid, block, 0, st, (const float*)src, ld_src, (bf16_t*)dst, ld_dst, R, C, Rpad);
    else hipLaunchKernelGGL((transpose_cast_kernel<float, float>), grid, block, 0, st, (const float*)src, ld_src, (float*)dst, ld_dst, R, C, Rpad);
    WHMR_CHECK_LAUNCH();
    return 0;
}

// ---- bf16 operand copies of ALL weights of a module in one launch: item i = an fp32 matrix [N, K] (nn.Linear layout), its bf16 copy (the W
// operand of the forward GEMM, y = x . W^T) and its bf16 transpose [K, N] (the W operand of the data-gradient GEMM, dX = dY . W).  The
// optimizer rewrites every weight every step, so a training step re-made these 2 x 49 copies of the ViT-B with 68 small launches; here a
// workgroup finds its matrix by its first-tile index (64 x 64 tiles, items sorted by tile_begin) and does both copies from one read.
struct whmr_wprep_item { const float* src; bf16_t* dst; bf16_t* dst_t; int32_t N, K, tile_begin, tiles_k; };

__global__ __launch_bounds__(256) void weights_prepare_kernel(const whmr_wprep_item* __restrict__ items, int n_items) {
    __shared__ float tile[64][65];
    int lo = 0, hi = n_items - 1;                    // last item whose tile_begin <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].tile_begin <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const whmr_wprep_item it = items[lo];
    const int t = blockIdx.x - it.tile_begin;
    const int n0 = (t / it.tiles_k) * 64, k0 = (t % it.tiles_k) * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;          // 64 columns x 4 rows per pass
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int n = n0 + ty + 4 * i, k = k0 + tx;
        const float v = (n < it.N && k < it.K) ? it.src[(size_t)n * it.K + k] : 0.f;
        tile[ty + 4 * i][tx] = v;
        if (it.dst && n < it.N && k < it.K) it.dst[(size_t)n * it.K + k] = f32_to_bf16(v);
    }
    __syncthreads();
    if (!it.dst_t) return;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int k = k0 + ty + 4 * i, n = n0 + tx;
        if (k < it.K && n < it.N) it.dst_t[(size_t)k * it.N + n] = f32_to_bf16(tile[tx][ty + 4 * i]);
    }
}

// items: DEVICE array of n_items descriptors sorted by tile_begin (item i owns tiles [tile_begin_i, tile_begin_{i+1}), ceil(N/64) * ceil(K/64) of them,
// tiles_k = ceil(K/64)); total_tiles = their sum.  dst / dst_t may be null.
extern "C" int whmr_weights_prepare(const void* items, int n_items, int total_tiles, void* stream) {
    if (!items || n_items <= 0 || total_tiles <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(weights_prepare_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, (const whmr_wprep_item*)items, n_items);
    WHMR_CHECK_LAUNCH();
    return 0;
}

// ---- column sums: partial[chunk][c] over row chunks, then a fixed-order sum of the partials (+= into out when accumulate)
#define CS_CHUNKS 64
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, long ld, int R, int C, float* __restrict__ partial) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
    const int rows_per = (R + (int)gridDim.y - 1) / (int)gridDim.y;
    const int r_begin = blockIdx.y * rows_per, r_end = min(R, r_begin + rows_per);
    float a = 0.f;
    if (c < C)
        for (int r = r_begin + part; r < r_end; r += 4) a += io<T>::ld(x + (size_t)r * ld + c);
    red[part][threadIdx.x & 63] = a;
    __syncthreads();
    if (part == 0 && c < C) partial[(size_t)blockIdx.y * C + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// block = 64 columns x 4 quarter-sums of the partials (fixed order: deterministic)
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, int nparts, int C, float* __restrict__ out,
                                                           int accumulate) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
    const int per = (nparts + 3) / 4;
    float a = 0.f;
    if (c < C) {
        const int p1 = min(nparts, (part + 1) * per);
#pragma unroll 8
        for (int p = part * per; p < p1; ++p) a += partial[(size_t)p * C + c];
    }
    red[part][threadIdx.x & 63] = a;
    __syncthreads();
    if (part == 0 && c < C) {
        const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        out[c] = accumulate ? out[c] + v : v;
    }
}

// bf16, C % 8 == 0: thread = 8 consecutive columns (16-B loads), block = 256 columns x 8 row lanes
__global__ __launch_bounds__(256) void colsum_partial8_kernel(const bf16_t* __restrict__ x, long ld, int R, int C, float* __restrict__ partial) {
    __shared__ float red[8][256 + 8];
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = blockIdx.x * 256 + cg * 8;
    const int rows_per = (R + (int)gridDim.y - 1) / (int)gridDim.y;
    const int r_begin = blockIdx.y * rows_per, r_end = min(R, r_begin + rows_per);
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c < C)
        for (int r = r_begin + rl; r < r_end; r += 8) {
            const uint4 v = *(const uint4*)(x + (size_t)r * ld + c);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[2 * e] += __uint_as_float(w[e] << 16); a[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u); }
        }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rl][cg * 8 + e] = a[e];
    __syncthreads();
    const int cc = blockIdx.x * 256 + threadIdx.x;
    if (cc < C) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) t += red[r][threadIdx.x];
        partial[(size_t)blockIdx.y * C + cc] = t;
    }
}

// scratch: >= max(64 * C, 2^20) floats
extern "C" int whmr_colsum(const void* x, int is_bf16, long ld, int R, int C, float* out, int accumulate, float* scratch, void* stream) {
    if (R <= 0 || C <= 0) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    const bool vec8 = is_bf16 && !(C & 7) && !(ld & 7) && !((uintptr_t)x & 15);
    // row chunks: 64 for the ViT shapes (12544 rows, 3-12 column blocks); tall narrow maps (IUV head: 786432 x 128) get more chunks so
    // that the partial pass fills the chip (64 blocks took 0.7 ms there).  chunks * C <= 2^20 floats of scratch.
    const int ncb = vec8 ? (C + 255) / 256 : (C + 63) / 64;
    int chunks = CS_CHUNKS;
    while (chunks < 1024 && (long)ncb * chunks < 1024 && R / chunks > 512 && (long)2 * chunks * C <= (1L << 20)) chunks *= 2;
    if (vec8)
        hipLaunchKernelGGL(colsum_partial8_kernel, dim3(ncb, chunks), dim3(256), 0, st, (const bf16_t*)x, ld, R, C, scratch);
    else if (is_bf16) hipLaunchKernelGGL(colsum_partial_kernel<bf16_t>, dim3(ncb, chunks), dim3(256), 0, st, (const bf16_t*)x, ld, R, C, scratch);
    else hipLaunchKernelGGL(colsum_partial_kernel<float>, dim3(ncb, chunks), dim3(256), 0, st, (const float*)x, ld, R, C, scratch);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 63) / 64), dim3(256), 0, st, scratch, chunks, C, out, accumulate);
    WHMR_CHECK_LAUNCH();
    return 0;
}

// ---- LayerNorm backward.  One wave per row (row kept in registers, like the forward kernel); waves stride over the rows and
// keep their dgamma / dbeta partial sums in registers; one partial per workgroup, reduced in a fixed order by the second kernel.
//   xhat = (x - mean) * rstd;  g = dy * gamma;  dx = rstd * (g - mean(g) - xhat * mean(g * xhat))  [+ dres]
template <int MAXV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            const float* __restrict__ gamma, const float* __restrict__ dres,
                                                            float* __restrict__ dx, float* __restrict__ partial, int rows, int C, float eps,
                                                            bf16_t* __restrict__ cast_out, const float* __restrict__ row_scale) {
    extern __shared__ float red[];                 // [4 waves][2][C]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = C >> 2;
    float4 dg[MAXV], db[MAXV], gm[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        dg[i] = db[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int c4 = lane + 64 * i;
        gm[i] = c4 < nv ? *(const float4*)(gamma + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const float* xr = x + (size_t)row * C;
        const float* dyr = dy + (size_t)row * C;
        float4 v[MAXV], d[MAXV], rres[MAXV];
        float s = 0.f;
        const float rs = (cast_out && row_scale) ? row_scale[row] : 1.0f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c4 = lane + 64 * i;
            if (c4 < nv) {
                v[i] = *(const float4*)(xr + c4 * 4);
                d[i] = *(const float4*)(dyr + c4 * 4);
                // the residual gradient is requested with the row (it was a dependent round trip behind the three reductions)
                rres[i] = dres ? *(const float4*)(dres + (size_t)row * C + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c4 = lane + 64 * i;
            if (c4 < nv) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
        const float mean = wave_sum(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c4 = lane + 64 * i;
            if (c4 < nv) {
                v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
                q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c4 = lane + 64 * i;
            if (c4 < nv) {
                v[i].x *= rstd; v[i].y *= rstd; v[i].z *= rstd; v[i].w *= rstd;          // xhat
                dg[i].x += d[i].x * v[i].x; dg[i].y += d[i].y * v[i].y; dg[i].z += d[i].z * v[i].z; dg[i].w += d[i].w * v[i].w;
                db[i].x += d[i].x; db[i].y += d[i].y; db[i].z += d[i].z; db[i].w += d[i].w;
                d[i].x *= gm[i].x; d[i].y *= gm[i].y; d[i].z *= gm[i].z; d[i].w *= gm[i].w;      // g
                sg += (d[i].x + d[i].y) + (d[i].z + d[i].w);
                sgx += (d[i].x * v[i].x + d[i].y * v[i].y) + (d[i].z * v[i].z + d[i].w * v[i].w);
            }
        }
        const float mg = wave_sum(sg) / (float)C, mgx = wave_sum(sgx) / (float)C;
        float* dxr = dx + (size_t)row * C;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c4 = lane + 64 * i;
            if (c4 < nv) {
                float4 o = make_float4(rstd * (d[i].x - mg - v[i].x * mgx), rstd * (d[i].y - mg - v[i].y * mgx),
                                       rstd * (d[i].z - mg - v[i].z * mgx), rstd * (d[i].w - mg - v[i].w * mgx));
                if (dres) { o.x += rres[i].x; o.y += rres[i].y; o.z += rres[i].z; o.w += rres[i].w; }
                *(float4*)(dxr + c4 * 4) = o;
                if (cast_out) {                        // the bf16 GEMM operand of the next branch's backward, stochastic-depth row factor applied
                    *(uint2*)(cast_out + (size_t)row * C + c4 * 4) = make_uint2(pack_bf16x2(o.x * rs, o.y * rs), pack_bf16x2(o.z * rs, o.w * rs));
                }
            }
        }
    }
    // workgroup partial of dgamma / dbeta
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c4 = lane + 64 * i;
        if (c4 < nv) {
            *(float4*)(red + (wave * 2 + 0) * C + c4 * 4) = dg[i];
            *(float4*)(red + (wave * 2 + 1) * C + c4 * 4) = db[i];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        const int which = i / C, c = i - which * C;
        partial[(size_t)blockIdx.x * 2 * C + i] = (red[(0 * 2 + which) * C + c] + red[(1 * 2 + which) * C + c]) +
                                                  (red[(2 * 2 + which) * C + c] + red[(3 * 2 + which) * C + c]);
    }
}

// 16 waves per 64 columns: each adds a contiguous 1/16 of the partial rows (64 loads deep, was 256 with 4 waves: 13.5 us, latency-bound), then a
// fixed-order tree over the 16 sums
__global__ __launch_bounds__(1024) void layernorm_bwd_final_kernel(const float* __restrict__ partial, int nparts, int C, float* __restrict__ dgamma,
                                                                   float* __restrict__ dbeta, int accumulate) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, i = blockIdx.x * 64 + lane, part = threadIdx.x >> 6;
    const int per = (nparts + 15) / 16;
    float a = 0.f;
    if (i < 2 * C) {
        const int p1 = min(nparts, (part + 1) * per);
#pragma unroll 16
        for (int p = part * per; p < p1; ++p) a += partial[(size_t)p * 2 * C + i];
    }
    red[part][lane] = a;
    __syncthreads();
    if (part == 0 && i < 2 * C) {
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = red[k][lane];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1)
#pragma unroll
            for (int k = 0; k < o; ++k) v[k] += v[k + o];
        float* dst = i < C ? dgamma + i : dbeta + (i - C);
        *dst = accumulate ? *dst + v[0] : v[0];
    }
}

#define LNB_BLOCKS 1024
// dx may alias dres.  scratch: >= LNB_BLOCKS * 2 * C floats (2048 * C).
// cast_out (nullable) [rows, C] bf16 = dx * row_scale[row] (row_scale nullable = 1): the compute-dtype copy of the residual-stream gradient that
// the NEXT branch's backward multiplies (whmr_scale_rows_cast / the plain cast as a by-product of this pass).
extern "C" int whmr_layernorm_bwd(const float* x, const float* dy, const float* gamma, const float* dres, float* dx, float* dgamma,
                                  float* dbeta, int accumulate, int rows, int C, float eps, float* scratch, void* cast_out,
                                  const float* row_scale, void* stream) {
    if (rows <= 0 || C <= 0 || (C & 3) || C > 64 * 4 * 4) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = rows < LNB_BLOCKS * 4 ? (rows + 3) / 4 : LNB_BLOCKS;
    const size_t lds = (size_t)8 * C * sizeof(float);
    if (C <= 64 * 4 * 3) hipLaunchKernelGGL(layernorm_bwd_kernel<3>, dim3(nblk), dim3(256), lds, st, x, dy, gamma, dres, dx, scratch, rows, C, eps, (bf16_t*)cast_out, row_scale);
    else hipLaunchKernelGGL(layernorm_bwd_kernel<4>, dim3(nblk), dim3(256), lds, st, x, dy, gamma, dres, dx, scratch, rows, C, eps, (bf16_t*)cast_out, row_scale);
    hipLaunchKernelGGL(layernorm_bwd_final_kernel, dim3((2 * C + 63) / 64), dim3(1024), 0, st, scratch, nblk, C, dgamma, dbeta, accumulate);
    WHMR_CHECK_LAUNCH();
    return 0;
}

// ---- GELU backward (exact erf form): d_pre = d_hid * (Phi(x) + x * phi(x)),  x = pre-activation
template <typename TP, typename TO>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const TP* __restrict__ pre, const float* __restrict__ dhid, TO* __restrict__ dpre, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = io<TP>::ld(pre + i);
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * expf(-0.5f * x * x);
    io<TO>::st(dpre + i, dhid[i] * (cdf + x * pdf));
}

// 8 elements per thread (bf16 in / bf16 or fp32 gradient in / bf16 out): the shapes of the training path
template <typename TD>
__global__ __launch_bounds__(256) void gelu_bwd8_kernel(const bf16_t* __restrict__ pre, const TD* __restrict__ dhid, bf16_t* __restrict__ dpre, long n8) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const uint4 pv = *(const uint4*)(pre + i * 8);
    const uint32_t pw[4] = {pv.x, pv.y, pv.z, pv.w};
    float g[8];
    if constexpr (sizeof(TD) == 4) {
        const float4 a = *(const float4*)(dhid + i * 8), b = *(const float4*)(dhid + i * 8 + 4);
        g[0] = a.x; g[1] = a.y; g[2] = a.z; g[3] = a.w; g[4] = b.x; g[5] = b.y; g[6] = b.z; g[7] = b.w;
    } else {
        const uint4 dv = *(const uint4*)(dhid + i * 8);
        const uint32_t dw[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { g[2 * e] = __uint_as_float(dw[e] << 16); g[2 * e + 1] = __uint_as_float(dw[e] & 0xffff0000u); }
    }
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = (e & 1) ? __uint_as_float(pw[e >> 1] & 0xffff0000u) : __uint_as_float(pw[e >> 1] << 16);
        const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
        const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
        o[e] = g[e] * (cdf + x * pdf);
    }
    *(uint4*)(dpre + i * 8) = make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
}

// dhid_bf16: the incoming gradient is bf16 (bf16 mode keeps d_hid in the compute dtype), else fp32
extern "C" int whmr_gelu_bwd(const void* pre, int pre_bf16, const void* dhid_v, int dhid_bf16, void* dpre, int out_bf16, long n, void* stream) {
    if (n <= 0) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (pre_bf16 && out_bf16 && !(n & 7)) {
        const dim3 g8((unsigned)((n / 8 + 255) / 256));
        if (dhid_bf16) hipLaunchKernelGGL(gelu_bwd8_kernel<bf16_t>, g8, block, 0, st, (const bf16_t*)pre, (const bf16_t*)dhid_v, (bf16_t*)dpre, n / 8);
        else hipLaunchKernelGGL(gelu_bwd8_kernel<float>, g8, block, 0, st, (const bf16_t*)pre, (const float*)dhid_v, (bf16_t*)dpre, n / 8);
        WHMR_CHECK_LAUNCH();
        return 0;
    }
    if (dhid_bf16) return (int)hipErrorInvalidValue;
    const float* dhid = (const float*)dhid_v;
    if (pre_bf16 && out_bf16) hipLaunchKernelGGL((gelu_bwd_kernel<bf16_t, bf16_t>), grid, block, 0, st, (const bf16_t*)pre, dhid, (bf16_t*)dpre, n);
    else if (pre_bf16) hipLaunchKernelGGL((gelu_bwd_kernel<bf16_t, float>), grid, block, 0, st, (const bf16_t*)pre, dhid, (float*)dpre, n);
    else if (out_bf16) hipLaunchKernelGGL((gelu_bwd_kernel<float, bf16_t>), grid, block, 0, st, (const float*)pre, dhid, (bf16_t*)dpre, n);
    else hipLaunchKernelGGL((gelu_bwd_kernel<float, float>), grid, block, 0, st, (const float*)pre, dhid, (float*)dpre, n);
    WHMR_CHECK_LAUNCH();
    return 0;
}

// GELU forward as its own pass (training keeps the pre-activation; the inference path fuses GELU into the fc1 epilogue)
template <typename T>
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const T* __restrict__ pre, T* __restrict__ out, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) io<T>::st(out + i, gelu_erf(io<T>::ld(pre + i)));
}

// bf16, 8 elements per thread (16-B loads / stores): the one-element form ran at 2.2 TB/s on the 12544 x 3072 hidden map
__global__ __launch_bounds__(256) void gelu_fwd8_kernel(const bf16_t* __restrict__ pre, bf16_t* __restrict__ out, long n8) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const uint4 pv = *(const uint4*)(pre + i * 8);
    const uint32_t pw[4] = {pv.x, pv.y, pv.z, pv.w};
    uint32_t o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
        o[e] = pack_bf16x2(gelu_erf(__uint_as_float(pw[e] << 16)), gelu_erf(__uint_as_float(pw[e] & 0xffff0000u)));
    *(uint4*)(out + i * 8) = make_uint4(o[0], o[1], o[2], o[3]);
}

extern "C" int whmr_gelu_fwd(const void* pre, void* out, int is_bf16, long n, void* stream) {
    if (n <= 0) return (int)hipErrorInvalidValue;
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (is_bf16 && !(n & 7) && !(((uintptr_t)pre | (uintptr_t)out) & 15))
        hipLaunchKernelGGL(gelu_fwd8_kernel, dim3((unsigned)((n / 8 + 255) / 256)), block, 0, (hipStream_t)stream, (const bf16_t*)pre, (bf16_t*)out, n / 8);
    else if (is_bf16) hipLaunchKernelGGL(gelu_fwd_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)pre, (bf16_t*)out, n);
    else hipLaunchKernelGGL(gelu_fwd_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)pre, (float*)out, n);
    WHMR_CHECK_LAUNCH();
    return 0;
}

// bf16 [R,C] -> bf16 [C,Rpad] transpose fused with the column sums of the source (out[c] (+)= sum_r src[r,c]).  Needs R, C, Rpad, ld_src, ld_dst
// multiples of 8 and 16-B aligned bases; scratch >= ceil(R/64) * C floats.
extern "C" int whmr_transpose_colsum(const void* src, long ld_src, void* dst, long ld_dst, int R, int C, int Rpad, float* out, int accumulate,
                                     float* scratch, void* stream) {
    if (R <= 0 || C <= 0 || Rpad < R || ld_dst < Rpad || ld_src < C) return (int)hipErrorInvalidValue;
    if (((R | C | Rpad | ld_src | ld_dst) & 7) || (((uintptr_t)src | (uintptr_t)dst) & 15)) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    const int nrb = (R + 63) / 64;                     // row blocks that hold real rows (blocks beyond them only write the zero padding)
    hipLaunchKernelGGL(transpose_bf16_colsum_kernel, dim3((Rpad + 63) / 64, (C + 63) / 64), dim3(256), 0, st, (const bf16_t*)src, ld_src, (bf16_t*)dst,
                       ld_dst, R, C, Rpad, scratch);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 63) / 64), dim3(256), 0, st, scratch, nrb, C, out, accumulate);
    WHMR_CHECK_LAUNCH();
    return 0;
}

// dst[m, :] = (T)(scale[m] * src[m, :]): the stochastic-depth mask applied to the residual-stream gradient that enters a dropped branch's
// backward (autograd of vit.py:132-139: d branch = mask / keep_prob * d out), fused with the cast to the GEMM operand dtype.
__global__ __launch_bounds__(256) void scale_rows_cast_kernel(const float* __restrict__ src, const float* __restrict__ scale, void* __restrict__ dst,
                                                              int M, int C4, int out_bf16) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)M * C4) return;
    const int m = (int)(idx / C4);
    const float s = scale[m];
    const float4 v = *(const float4*)(src + idx * 4);
    if (out_bf16) *(uint2*)((bf16_t*)dst + idx * 4) = make_uint2(pack_bf16x2(v.x * s, v.y * s), pack_bf16x2(v.z * s, v.w * s));
    else *(float4*)((float*)dst + idx * 4) = make_float4(v.x * s, v.y * s, v.z * s, v.w * s);
}

extern "C" int whmr_scale_rows_cast(const float* src, const float* scale, void* dst, int M, int C, int out_bf16, void* stream) {
    if (M <= 0 || C <= 0 || (C & 3)) return (int)hipErrorInvalidValue;
    const long n4 = (long)M * (C >> 2);
    hipLaunchKernelGGL(scale_rows_cast_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, scale, dst, M, C >> 2, out_bf16);
    WHMR_CHECK_LAUNCH();
    return 0;
}
