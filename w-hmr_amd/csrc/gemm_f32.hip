// fp32 MFMA GEMM / implicit-GEMM (parity path + skinny regressor GEMMs):  C = epilogue(A[M,K] . W[N,K]^T)
//
// Exact-f32 arithmetic on v_mfma_f32_32x32x2_f32 (bitwise an fmaf chain over k; no reduced precision on gfx950),
// so the result differs from the reference's CPU sgemm only by summation order.  Any M, N, K (all masked).
// Used for: every GEMM of the fp32 "parity" mode of the ViT / deconv path, and always for the regressor stages
// (whmr.py:118-126: fc1 -> fc2 -> decpose/decshape/deccam), Global_Orient_Regressor (whmr.py:295-301), the Tz
// head linears (whmr.py:425-430), the second Tz conv (whmr.py:420) -- M = batch, weight-streaming bound.
//
// 64x64x16 block tile, 4 waves (2x2), one 32x32 accumulator per wave, register-staged LDS (rows padded to 17
// dwords: conflict-free ds_read_b32 fragment reads).  Same descriptor / gather / scatter modes as the bf16 kernel.
#include "common.h"
#include "gemm_params.h"

#define FBM 64
#define FBN 64
#define FBK 16
#define FLD 17

template <bool GATHER, bool SPLIT>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const whmr_gemm p, int k_per_split) {
    __shared__ float sA[FBM * FLD];
    __shared__ float sB[FBN * FLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_n = (p.N + FBN - 1) / FBN;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int m0 = tm * FBM, n0 = tn * FBN;
    const float* __restrict__ A = (const float*)p.A;
    const float* __restrict__ W = (const float*)p.W;

    // staging: thread -> (row = tid>>2, 4 consecutive k = (tid&3)*4)
    const int srow = tid >> 2, sk = (tid & 3) * 4;
    const int am = m0 + srow, bn = n0 + srow;
    const bool a_ok = am < p.M, b_ok = bn < p.N;
    int ay = 0, ax = 0;
    const float* a_base = A;
    if (a_ok) {
        if constexpr (GATHER) {
            const int ohw = p.OH * p.OW;
            const int b = am / ohw, rem = am - b * ohw;
            const int oy = rem / p.OW, ox = rem - oy * p.OW;
            ay = oy * p.SH - p.PH;
            ax = ox * p.SW - p.PW;
            a_base = A + (size_t)b * p.IH * p.IW * p.Cin;
        } else {
            a_base = A + (size_t)am * p.lda;
        }
    }
    const float* b_base = W + (size_t)(b_ok ? bn : 0) * p.K;

    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    const int k_begin = SPLIT ? blockIdx.y * k_per_split : 0;
    const int k_end = SPLIT ? min(p.K, k_begin + k_per_split) : p.K;
    for (int k0 = k_begin; k0 < k_end; k0 += FBK) {
        float av[4], bw[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = k0 + sk + e;
            float a = 0.f, b = 0.f;
            if (k < k_end) {
                if (a_ok) {
                    if constexpr (GATHER) {
                        const int tap = k / p.Cin, ci = k - tap * p.Cin;
                        const int ky = tap / p.KW, kx = tap - ky * p.KW;
                        const int iy = ay + ky, ix = ax + kx;
                        if ((unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW)
                            a = a_base[((size_t)iy * p.IW + ix) * p.Cin + ci];
                    } else {
                        a = a_base[k];
                    }
                }
                if (b_ok) b = b_base[k];
            }
            av[e] = a; bw[e] = b;
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sA[srow * FLD + sk + e] = av[e];
            sB[srow * FLD + sk + e] = bw[e];
        }
        __syncthreads();
        // A operand: lane l holds A[i = l&31][k = l>>5]; B operand: B[k = l>>5][j = l&31]
#pragma unroll
        for (int kk = 0; kk < FBK; kk += 2) {
            const float a = sA[(wm * 32 + l31) * FLD + kk + hi];
            const float b = sB[(wn * 32 + l31) * FLD + kk + hi];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }

    const int n = n0 + wn * 32 + l31;
    if (n >= p.N) return;
    if constexpr (SPLIT) {           // raw partial sums; bias / activation / residual happen in splitk_reduce_kernel
        float* ws = (float*)p.workspace + (size_t)blockIdx.y * p.M * p.N;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (m < p.M) ws[(size_t)m * p.N + n] = acc[r];
        }
        return;
    }
    const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (m >= p.M) continue;
        float v = acc[r] + bv;
        float rv = 0.f;
        if (p.residual) {
            const int rr = p.res_row_mod > 0 ? m % p.res_row_mod : m;
            rv = p.residual[(size_t)rr * p.ldr + n];
        }
        if (p.epi_flags & 2) v += rv;                 // ResNet bottleneck: skip added before the ReLU
        if (p.act == 1) v = gelu_erf(v);
        else if (p.act == 2) v = fmaxf(v, 0.f);
        if (p.row_scale) v *= p.row_scale[m];
        if (!(p.epi_flags & 2)) v += rv;
        size_t off;
        if (p.c_mode == 1) {
            const int ohw = p.OH * p.OW;
            const int b = m / ohw, rem = m - b * ohw;
            const int oy = rem / p.OW, ox = rem - oy * p.OW;
            off = (size_t)(p.c_off + b * p.osb + oy * p.osy + ox * p.osx) + n;
        } else {
            off = (size_t)m * p.ldc + n;
        }
        if (p.out_bf16) ((bf16_t*)p.C)[off] = f32_to_bf16(v);
        else ((float*)p.C)[off] = v;
    }
}

// ---- large-M variant (the ViT / deconv / conv GEMMs of the fp32 parity mode: M = all tokens or pixels).  128x128x16 block tile, 4 waves
// (2x2), wave tile 64x64 = 2x2 accumulators; 16-B global loads of the NEXT K step are in flight under the 32 MFMAs of the current one
// (register-staged, two LDS buffers, one barrier per step).  LDS rows hold the 16 k of a step permuted so that the four operands a lane
// feeds to four consecutive v_mfma_f32_32x32x2_f32 (k = 8g + 2j + hi, j = 0..3) are ONE ds_read_b128: position of k in its row =
// 8 (k / 8) + 4 (k % 2) + (k % 8) / 2.  The K order of the MFMA chain is unchanged (0, 1, 2, ...): same bits as the 64x64 kernel.
// Requires K % 4 == 0 and 16-B aligned operand rows (plain: lda % 4 == 0; gather: Cin % 4 == 0); the launcher falls back otherwise.
#define GBM 128
#define GBN 128
#define GLD 20
template <bool GATHER>
__global__ __launch_bounds__(256, 2) void gemm_f32_big_kernel(const whmr_gemm p) {
    __shared__ __attribute__((aligned(16))) float sA[2][GBM * GLD];
    __shared__ __attribute__((aligned(16))) float sB[2][GBN * GLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_n = (p.N + GBN - 1) / GBN;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int m0 = tm * GBM, n0 = tn * GBN;
    const float* __restrict__ A = (const float*)p.A;
    const float* __restrict__ W = (const float*)p.W;
    // staging: thread -> rows (tid >> 2) and (tid >> 2) + 64, 4 consecutive k at (tid & 3) * 4
    const int srow = tid >> 2, sk = (tid & 3) * 4;
    const float* a_base[2];
    const float* b_base[2];
    bool a_ok[2], b_ok[2];
    int ay[2] = {0, 0}, ax[2] = {0, 0};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int am = m0 + srow + 64 * h, bn = n0 + srow + 64 * h;
        a_ok[h] = am < p.M; b_ok[h] = bn < p.N;
        a_base[h] = A;
        if (a_ok[h]) {
            if constexpr (GATHER) {
                const int ohw = p.OH * p.OW;
                const int b = am / ohw, rem = am - b * ohw;
                const int oy = rem / p.OW, ox = rem - oy * p.OW;
                ay[h] = oy * p.SH - p.PH; ax[h] = ox * p.SW - p.PW;
                a_base[h] = A + (size_t)b * p.IH * p.IW * p.Cin;
            } else {
                a_base[h] = A + (size_t)am * p.lda;
            }
        }
        b_base[h] = W + (size_t)(b_ok[h] ? bn : 0) * p.K;
    }
    float4 ra[2], rb[2];
    auto fetch = [&](int k0) {
        const int k = k0 + sk;
        const bool k_ok = k < p.K;                               // K % 4 == 0: a 16-B group is inside or outside as a whole
        int ky = 0, kx = 0, ci = k;
        if constexpr (GATHER) {
            const int tap = k / p.Cin;
            ci = k - tap * p.Cin;
            ky = tap / p.KW; kx = tap - ky * p.KW;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
            if (k_ok) {
                if (a_ok[h]) {
                    if constexpr (GATHER) {
                        const int iy = ay[h] + ky, ix = ax[h] + kx;
                        if ((unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW)
                            a = *(const float4*)(a_base[h] + ((size_t)iy * p.IW + ix) * p.Cin + ci);
                    } else {
                        a = *(const float4*)(a_base[h] + k);
                    }
                }
                if (b_ok[h]) b = *(const float4*)(b_base[h] + k);
            }
            ra[h] = a; rb[h] = b;
        }
    };
    // k = sk + e (e = 0..3) -> position 8 (k / 8) + 4 (k % 2) + (k % 8) / 2
    const int pbase = (sk >> 3) * 8 + ((sk & 4) >> 1);           // sk % 8 == 0: positions 0,4,1,5;  sk % 8 == 4: positions 2,6,3,7
    auto stash = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float* da = &sA[buf][(srow + 64 * h) * GLD + pbase];
            float* db = &sB[buf][(srow + 64 * h) * GLD + pbase];
            da[0] = ra[h].x; da[4] = ra[h].y; da[1] = ra[h].z; da[5] = ra[h].w;
            db[0] = rb[h].x; db[4] = rb[h].y; db[1] = rb[h].z; db[5] = rb[h].w;
        }
    };
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hi = lane >> 5;
    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    fetch(0);
    stash(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < p.K; k0 += 16) {
        const bool more = k0 + 16 < p.K;
        if (more) fetch(k0 + 16);                                // in flight under the MFMAs below
        const float* pa = &sA[buf][(wm * 64 + l31) * GLD + hi * 4];
        const float* pb = &sB[buf][(wn * 64 + l31) * GLD + hi * 4];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const float4 a0 = *(const float4*)(pa + g * 8), a1 = *(const float4*)(pa + 32 * GLD + g * 8);
            const float4 b0 = *(const float4*)(pb + g * 8), b1 = *(const float4*)(pb + 32 * GLD + g * 8);
            const float av[2][4] = {{a0.x, a0.y, a0.z, a0.w}, {a1.x, a1.y, a1.z, a1.w}};
            const float bv[2][4] = {{b0.x, b0.y, b0.z, b0.w}, {b1.x, b1.y, b1.z, b1.w}};
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][j4], bv[j][j4], acc[i][j], 0, 0, 0);
        }
        if (more) {
            stash(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + l31;
        if (n >= p.N) continue;
        const float bvs = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (m >= p.M) continue;
                float v = acc[i][j][r] + bvs;
                float rv = 0.f;
                if (p.residual) {
                    const int rr = p.res_row_mod > 0 ? m % p.res_row_mod : m;
                    rv = p.residual[(size_t)rr * p.ldr + n];
                }
                if (p.epi_flags & 2) v += rv;
                if (p.act == 1) v = gelu_erf(v);
                else if (p.act == 2) v = fmaxf(v, 0.f);
                if (p.row_scale) v *= p.row_scale[m];
                if (!(p.epi_flags & 2)) v += rv;
                size_t off;
                if (p.c_mode == 1) {
                    const int ohw = p.OH * p.OW;
                    const int b = m / ohw, rem = m - b * ohw;
                    const int oy = rem / p.OW, ox = rem - oy * p.OW;
                    off = (size_t)(p.c_off + b * p.osb + oy * p.osy + ox * p.osx) + n;
                } else {
                    off = (size_t)m * p.ldc + n;
                }
                if (p.out_bf16) ((bf16_t*)p.C)[off] = f32_to_bf16(v);
                else ((float*)p.C)[off] = v;
            }
    }
}

// ---- skinny variant (M <= 1024, typically M = batch: regressor / global-orient / Tz linears, SMPL pose-corrective product; whmr.py:118-126,295-301).
// These GEMMs stream a weight matrix (9-17 MB) past 64 activation rows: the job is to keep HBM busy, not the matrix pipes.
// Same 64x64 tile and exact-f32 MFMA as above, but K steps of 32 with the NEXT step's global loads (4 x 16 B per thread)
// issued before the current step's MFMAs, so every block always has 16 KB in flight instead of one exposed round trip
// per 16-wide K step.  Rows are only 4-byte aligned in general (K = 2149, 207): the 16-B loads are declared align(4).
#define SBK 32
#define SLD 33
typedef float f32x4u_t __attribute__((ext_vector_type(4), aligned(4)));

// TR: operands given REDUCTION-MAJOR (epi_flags bits 4 / 5): bit 0 -> A is [K, lda >= M], bit 1 -> W is [K, N] (dense).  The backward products of
// nn.Linear (dX = dY . W: W as stored is the reduction-major operand; dW = dY^T . X: both are) then need no transposed copies -- the
// tile is transposed on its way into LDS (16-B loads along the contiguous m / n, scalar LDS stores down a column of the padded tile).
template <bool SPLIT, int TR = 0>
__global__ __launch_bounds__(256) void gemm_f32_skinny_kernel(const whmr_gemm p, int k_per_split) {
    __shared__ float sA[64 * SLD];
    __shared__ float sB[64 * SLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * 64, m0 = blockIdx.z * 64;
    const float* __restrict__ A = (const float*)p.A;
    const float* __restrict__ W = (const float*)p.W;
    const int k_begin = SPLIT ? blockIdx.y * k_per_split : 0;
    const int k_end = SPLIT ? min(p.K, k_begin + k_per_split) : p.K;
    // staging: thread -> rows (tid >> 3) and (tid >> 3) + 32, 4 consecutive k at (tid & 7) * 4
    const int srow = tid >> 3, sk = (tid & 7) * 4;
    const float* a_ptr[2];
    const float* b_ptr[2];
    bool a_ok[2], b_ok[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int r = srow + 32 * h;
        a_ok[h] = m0 + r < p.M;
        b_ok[h] = n0 + r < p.N;
        a_ptr[h] = A + (size_t)(a_ok[h] ? m0 + r : 0) * p.lda + sk;
        b_ptr[h] = W + (size_t)(b_ok[h] ? n0 + r : 0) * p.K + sk;
    }
    float ra[2][4], rb[2][4];
    // reduction-major operand: thread -> k rows (tid >> 4) and (tid >> 4) + 16, 4 consecutive m (or n) at (tid & 15) * 4
    const int tk = tid >> 4, tc = (tid & 15) * 4;
    auto fetch_t = [&](const float* __restrict__ src, long ld, int c0, int cmax, int k0, float (*r)[4]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = k0 + tk + 16 * h;
            const float* q = src + (size_t)(k < k_end ? k : 0) * ld + c0 + tc;
            if (k < k_end && c0 + tc + 4 <= cmax) {
                const f32x4u_t v = *(const f32x4u_t*)q;
#pragma unroll
                for (int e = 0; e < 4; ++e) r[h][e] = v[e];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) r[h][e] = (k < k_end && c0 + tc + e < cmax) ? q[e] : 0.f;
            }
        }
    };
    auto fetch_rm = [&](const float* const* ptr, const bool* ok, int k0, float (*r)[4]) {      // row-major operand: 4 consecutive k of one row
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (k0 + sk + 4 <= k_end) {                          // whole 16-B group inside the slice
                const f32x4u_t v = ok[h] ? *(const f32x4u_t*)(ptr[h] + k0) : f32x4u_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) r[h][e] = v[e];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) r[h][e] = (k0 + sk + e < k_end && ok[h]) ? ptr[h][k0 + e] : 0.f;
            }
        }
    };
    auto fetch = [&](int k0) {
        if constexpr (TR & 1) fetch_t(A, p.lda, m0, p.M, k0, ra);
        else fetch_rm(a_ptr, a_ok, k0, ra);
        if constexpr (TR & 2) fetch_t(W, p.N, n0, p.N, k0, rb);
        else fetch_rm(b_ptr, b_ok, k0, rb);
    };
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hi = lane >> 5;
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (k_begin < k_end) fetch(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += SBK) {
        __syncthreads();                                         // previous step's fragment reads are done
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if constexpr (TR & 1) sA[(tc + e) * SLD + tk + 16 * h] = ra[h][e];
                else sA[(srow + 32 * h) * SLD + sk + e] = ra[h][e];
                if constexpr (TR & 2) sB[(tc + e) * SLD + tk + 16 * h] = rb[h][e];
                else sB[(srow + 32 * h) * SLD + sk + e] = rb[h][e];
            }
        __syncthreads();
        if (k0 + SBK < k_end) fetch(k0 + SBK);                   // in flight under the MFMAs below
#pragma unroll
        for (int kk = 0; kk < SBK; kk += 2) {
            const float a = sA[(wm * 32 + l31) * SLD + kk + hi];
            const float b = sB[(wn * 32 + l31) * SLD + kk + hi];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
    const int n = n0 + wn * 32 + l31;
    if (n >= p.N) return;
    if constexpr (SPLIT) {
        float* ws = (float*)p.workspace + (size_t)blockIdx.y * p.M * p.N;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (m < p.M) ws[(size_t)m * p.N + n] = acc[r];
        }
        return;
    }
    const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (m >= p.M) continue;
        float v = acc[r] + bv;
        const float rv = p.residual ? p.residual[(size_t)(p.res_row_mod > 0 ? m % p.res_row_mod : m) * p.ldr + n] : 0.f;
        if (p.epi_flags & 2) v += rv;
        if (p.act == 1) v = gelu_erf(v);
        else if (p.act == 2) v = fmaxf(v, 0.f);
        if (p.row_scale) v *= p.row_scale[m];
        if (!(p.epi_flags & 2)) v += rv;
        const size_t off = (size_t)m * p.ldc + n;
        if (p.out_bf16) ((bf16_t*)p.C)[off] = f32_to_bf16(v);
        else ((float*)p.C)[off] = v;
    }
}

// Sum the split-K partials in a fixed order (deterministic), then the usual epilogue.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const whmr_gemm p, int splits) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)p.M * p.N) return;
    const int m = (int)(idx / p.N), n = (int)(idx - (long)m * p.N);
    const float* ws = (const float*)p.workspace + idx;
    const size_t stride = (size_t)p.M * p.N;
    float v = 0.f;
    int s = 0;
    for (; s + 8 <= splits; s += 8) {                 // 8 independent loads in flight, summed in split order
        float t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = ws[(size_t)(s + e) * stride];
#pragma unroll
        for (int e = 0; e < 8; ++e) v += t[e];
    }
    for (; s < splits; ++s) v += ws[(size_t)s * stride];
    if (p.bias) v += p.bias[n];
    const float rv = p.residual ? p.residual[(size_t)(p.res_row_mod > 0 ? m % p.res_row_mod : m) * p.ldr + n] : 0.f;
    if (p.epi_flags & 2) v += rv;
    if (p.act == 1) v = gelu_erf(v);
    else if (p.act == 2) v = fmaxf(v, 0.f);
    if (p.row_scale) v *= p.row_scale[m];
    if (!(p.epi_flags & 2)) v += rv;
    const size_t off = (size_t)m * p.ldc + n;
    if (p.out_bf16) ((bf16_t*)p.C)[off] = f32_to_bf16(v);
    else ((float*)p.C)[off] = v;
}

static int g_f32_big = 1;       // whmr_gemm_f32_set_big(0): A/B and tests against the 64x64 kernel
extern "C" int whmr_gemm_f32_set_big(int on) { g_f32_big = on; return 0; }

extern "C" int whmr_gemm_f32(const whmr_gemm* pp, int flags, void* stream) {
    (void)flags;
    const whmr_gemm& p = *pp;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return (int)hipErrorInvalidValue;
    const int tiles = ((p.M + FBM - 1) / FBM) * ((p.N + FBN - 1) / FBN);
    hipStream_t st = (hipStream_t)stream;
    const int tr = (p.epi_flags >> 4) & 3;                  // reduction-major operands: the skinny kernel only
    if (tr && !(p.M <= 1024 && p.a_mode == 0 && p.c_mode == 0)) return (int)hipErrorInvalidValue;
    if (p.M <= 1024 && p.a_mode == 0 && p.c_mode == 0) {
        // weight-streaming regime: ~2 blocks per CU so that the whole weight matrix is in flight at once
        const int tiles_n = (p.N + 63) / 64, tiles_m = (p.M + 63) / 64, tiles = tiles_n * tiles_m;
        int splits = 1;
        if (p.workspace && tiles < 384 && p.K >= 256) {
            splits = (512 + tiles - 1) / tiles;              // ~2 blocks per CU
            if (splits > p.K / 128) splits = p.K / 128;      // >= 4 pipelined K steps each; partial-sum traffic <= 1/2 of the weights
            if (splits > 32) splits = 32;
            while (splits > 1 && (int64_t)splits * p.M * p.N * 4 > p.workspace_bytes) --splits;
        }
        if (splits > 1) {
            int kps = (p.K + splits - 1) / splits;
            kps = (kps + SBK - 1) / SBK * SBK;
            splits = (p.K + kps - 1) / kps;
            const dim3 g(tiles_n, splits, tiles_m);
            if (tr == 0) hipLaunchKernelGGL((gemm_f32_skinny_kernel<true, 0>), g, dim3(256), 0, st, p, kps);
            else if (tr == 1) hipLaunchKernelGGL((gemm_f32_skinny_kernel<true, 1>), g, dim3(256), 0, st, p, kps);
            else if (tr == 2) hipLaunchKernelGGL((gemm_f32_skinny_kernel<true, 2>), g, dim3(256), 0, st, p, kps);
            else hipLaunchKernelGGL((gemm_f32_skinny_kernel<true, 3>), g, dim3(256), 0, st, p, kps);
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(((long)p.M * p.N + 255) / 256)), dim3(256), 0, st, p, splits);
        } else {
            const dim3 g(tiles_n, 1, tiles_m);
            if (tr == 0) hipLaunchKernelGGL((gemm_f32_skinny_kernel<false, 0>), g, dim3(256), 0, st, p, 0);
            else if (tr == 1) hipLaunchKernelGGL((gemm_f32_skinny_kernel<false, 1>), g, dim3(256), 0, st, p, 0);
            else if (tr == 2) hipLaunchKernelGGL((gemm_f32_skinny_kernel<false, 2>), g, dim3(256), 0, st, p, 0);
            else hipLaunchKernelGGL((gemm_f32_skinny_kernel<false, 3>), g, dim3(256), 0, st, p, 0);
        }
        WHMR_CHECK_LAUNCH();
        return 0;
    }
    // large M with aligned rows: the 128x128 double-buffered kernel (5x the 64x64 kernel's rate on the ViT shapes)
    {
        const bool aligned = !(p.K & 3) && !((uintptr_t)p.A & 15) && !((uintptr_t)p.W & 15) &&
                             (p.a_mode == 1 ? !(p.Cin & 3) : !(p.lda & 3));
        const long big_tiles = (long)((p.M + GBM - 1) / GBM) * ((p.N + GBN - 1) / GBN);
        if (aligned && p.M >= 1024 && p.N >= 64 && big_tiles >= 128 && g_f32_big) {
            if (p.a_mode == 1) hipLaunchKernelGGL((gemm_f32_big_kernel<true>), dim3((unsigned)big_tiles), dim3(256), 0, st, p);
            else hipLaunchKernelGGL((gemm_f32_big_kernel<false>), dim3((unsigned)big_tiles), dim3(256), 0, st, p);
            WHMR_CHECK_LAUNCH();
            return 0;
        }
    }
    // Skinny shapes (M = batch: regressor / global-orient / Tz linears) leave most CUs idle and are weight-streaming bound:
    // split K across blocks so that ~2 blocks per CU stream the weight matrix concurrently.
    int splits = 1;
    if (p.a_mode == 0 && p.c_mode == 0 && p.workspace && tiles < 128 && p.K >= 512) {
        splits = (384 + tiles - 1) / tiles;
        if (splits > p.K / 128) splits = p.K / 128;
        if (splits > 32) splits = 32;
        while (splits > 1 && (int64_t)splits * p.M * p.N * 4 > p.workspace_bytes) --splits;
    }
    if (splits > 1) {
        int kps = (p.K + splits - 1) / splits;
        kps = (kps + FBK - 1) / FBK * FBK;
        splits = (p.K + kps - 1) / kps;
        hipLaunchKernelGGL((gemm_f32_kernel<false, true>), dim3(tiles, splits), dim3(256), 0, st, p, kps);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(((long)p.M * p.N + 255) / 256)), dim3(256), 0, st, p, splits);
    } else if (p.a_mode == 1) {
        hipLaunchKernelGGL((gemm_f32_kernel<true, false>), dim3(tiles), dim3(256), 0, st, p, 0);
    } else {
        hipLaunchKernelGGL((gemm_f32_kernel<false, false>), dim3(tiles), dim3(256), 0, st, p, 0);
    }
    WHMR_CHECK_LAUNCH();
    return 0;
}
