// bf16 MFMA GEMM / implicit-GEMM for gfx950:  C[M,N] = epilogue(A[M,K] . W[N,K]^T)
//
// Replaces the reference's nn.Linear / Conv2d / ConvTranspose2d call sites on the hot path:
//   vit.py:93,96,101,112 (qkv / proj), vit.py:66-75 (fc1 + GELU, fc2), vit.py:157-164 (patch embed, after im2col),
//   whmr.py:488-498 (deconv k4 s2 p1 as 4 sub-pixel 2x2 convs, BN folded, ReLU), whmr.py:419 (7x7 s3 conv).
//
// Structure: 128x128x64 block tile, 4 waves (2x2), each wave 64x64 = 2x2 v_mfma_f32_32x32x16_bf16 tiles,
// fp32 accumulation.  Operands are staged HBM -> LDS with global_load_lds (16 B/lane, no VGPR round trip),
// double buffered, one barrier per K tile.  The LDS image is lane-linear, so the bank-conflict XOR swizzle
// (16-B chunk ^= (row>>1)&7) is applied to the per-lane SOURCE address and again on the ds_read side.
// A rows can be gathered from an NHWC image (conv taps; out-of-image taps read a zero page).
#include "common.h"
#include "gemm_params.h"

#define BM 128
#define BN 128
#define BK 64
#define STAGE_BYTES (BM * BK * 2 + BN * BK * 2)   // 32 KiB

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

template <bool GLDS>
__device__ __forceinline__ void stage_chunk(const bf16_t* src, char* lds_base, int wave_chunk0, int lane) {
    if constexpr (GLDS) {
        // LDS destination = wave-uniform base + lane * 16
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(lds_base + wave_chunk0 * 16), 16, 0, 0);
    } else {
        *(uint4*)(lds_base + (wave_chunk0 + lane) * 16) = *(const uint4*)src;
    }
}

template <int OUT_BF16, int ACT, bool GATHER, bool GLDS>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const whmr_gemm p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int lid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = lid / tiles_n, tn = lid % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const bf16_t* __restrict__ A = (const bf16_t*)p.A;
    const bf16_t* __restrict__ W = (const bf16_t*)p.W;

    // ---- per-thread staging geometry: 4 chunks of A and 4 of B per K tile
    // chunk c = tid + 256*i -> tile row c>>3 = (tid>>3) + 32*i, physical 16-B slot tid&7
    const int srow = tid >> 3, pc = tid & 7;
    const bf16_t* a_src[4];
    int a_y[4], a_x[4];
    const bf16_t* b_src[4];
    int lc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = srow + 32 * i;
        lc[i] = pc ^ ((row >> 1) & 7);
        int m = m0 + row;
        if (m > p.M - 1) m = p.M - 1;
        if constexpr (GATHER) {
            const int ohw = p.OH * p.OW;
            const int b = m / ohw, rem = m - b * ohw;
            const int oy = rem / p.OW, ox = rem - oy * p.OW;
            a_y[i] = oy * p.SH - p.PH;
            a_x[i] = ox * p.SW - p.PW;
            a_src[i] = A + (size_t)b * p.IH * p.IW * p.Cin + lc[i] * 8;
        } else {
            a_src[i] = A + (size_t)m * p.lda + lc[i] * 8;
        }
        int nr = n0 + row;
        if (nr > p.N - 1) nr = p.N - 1;             // N tail: clamp the load, mask the store
        b_src[i] = W + (size_t)nr * p.K + lc[i] * 8;
    }

    auto stage = [&](int kt, int s) {
        char* sa = smem + s * STAGE_BYTES;
        char* sb = sa + BM * BK * 2;
        const int k0 = kt * BK;
        int ky = 0, kx = 0, ci0 = 0;
        if constexpr (GATHER) {
            const int tap = k0 / p.Cin;
            ci0 = k0 - tap * p.Cin;
            ky = tap / p.KW;
            kx = tap - ky * p.KW;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16_t* src;
            if constexpr (GATHER) {
                const int iy = a_y[i] + ky, ix = a_x[i] + kx;
                const bool ok = (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
                src = ok ? a_src[i] + ((size_t)iy * p.IW + ix) * p.Cin + ci0 : (const bf16_t*)p.zeros + pc * 8;
            } else {
                src = a_src[i] + k0;
            }
            stage_chunk<GLDS>(src, sa, wave * 64 + 256 * i, lane);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) stage_chunk<GLDS>(b_src[i] + k0, sb, wave * 64 + 256 * i, lane);
    };

    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment row byte offsets + swizzle keys (constant over the K loop)
    int a_off[2], a_sw[2], b_off[2], b_sw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ra = wm * 64 + i * 32 + l31, rb = wn * 64 + i * 32 + l31;
        a_off[i] = ra * (BK * 2); a_sw[i] = (ra >> 1) & 7;
        b_off[i] = rb * (BK * 2); b_sw[i] = (rb >> 1) & 7;
    }

    const int nkt = p.K / BK;
    stage(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int s = kt & 1;
        if (kt + 1 < nkt) stage(kt + 1, s ^ 1);
        const char* sa = smem + s * STAGE_BYTES;
        const char* sb = sa + BM * BK * 2;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8_t af[2], bfr[2];
            const int c = kk * 2 + hi;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = *(const bf16x8_t*)(sa + a_off[i] + ((c ^ a_sw[i]) << 4));
                bfr[i] = *(const bf16x8_t*)(sb + b_off[i] + ((c ^ b_sw[i]) << 4));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue.  Accumulators (C/D map: col = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5)) get bias + activation in
    // registers, are transposed through the now-idle LDS (two 64-row passes, fp32, rows padded to 132 dwords), and
    // leave as whole 512-B rows: 16-B residual loads and 16-B (fp32) / 8-B (bf16) stores, fully coalesced.
    constexpr int CLD = BN + 4;
    float* sC = (float*)smem;                    // 64 x 132 fp32 = 33 KiB
    float bv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + l31;
        bv[j] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
    }
    const float* __restrict__ res = p.residual;
    bool spatial = false;
    if constexpr (GATHER) spatial = (p.c_mode == 1);
    const int c4 = (tid & 31) * 4;               // this thread's 4 consecutive columns in the tile
    const int rsub = tid >> 5;                   // row within an 8-row group
    const bool vec_ok = (n0 + BN <= p.N) && ((p.ldc & 3) == 0 || spatial) && (!res || (p.ldr & 3) == 0);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (wm == pass) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[i][j][r] + bv[j];
                        if (ACT == 1) v = gelu_fast(v);
                        if (ACT == 2) v = fmaxf(v, 0.f);
                        sC[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * CLD + wn * 64 + j * 32 + l31] = v;
                    }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int lr = it * 8 + rsub;                       // row inside this 64-row pass
            const int m = m0 + pass * 64 + lr;
            if (m >= p.M) continue;
            float4 v = *(const float4*)(sC + lr * CLD + c4);
            size_t crow;
            if (spatial) {
                const int ohw = p.OH * p.OW;
                const int b = m / ohw, rem = m - b * ohw;
                const int oy = rem / p.OW, ox = rem - oy * p.OW;
                crow = (size_t)(p.c_off + b * p.osb + oy * p.osy + ox * p.osx);
            } else {
                crow = (size_t)m * p.ldc;
            }
            const int n = n0 + c4;
            if (vec_ok) {
                if (res) {
                    const size_t rrow = (size_t)(p.res_row_mod > 0 ? m % p.res_row_mod : m) * p.ldr;
                    const float4 rv = *(const float4*)(res + rrow + n);
                    v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
                }
                if (OUT_BF16) *(uint2*)((bf16_t*)p.C + crow + n) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
                else *(float4*)((float*)p.C + crow + n) = v;
            } else {
                const float vv[4] = {v.x, v.y, v.z, v.w};
                size_t rrow = 0;
                if (res) rrow = (size_t)(p.res_row_mod > 0 ? m % p.res_row_mod : m) * p.ldr;
                for (int e = 0; e < 4; ++e) {
                    if (n + e >= p.N) break;
                    float o = vv[e];
                    if (res) o += res[rrow + n + e];
                    if (OUT_BF16) ((bf16_t*)p.C)[crow + n + e] = f32_to_bf16(o);
                    else ((float*)p.C)[crow + n + e] = o;
                }
            }
        }
        __syncthreads();
    }
}

template <int OUT_BF16, int ACT, bool GATHER>
static int launch(const whmr_gemm& p, hipStream_t st, bool glds) {
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    if (glds) hipLaunchKernelGGL((gemm_bf16_kernel<OUT_BF16, ACT, GATHER, true>), dim3(tiles), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((gemm_bf16_kernel<OUT_BF16, ACT, GATHER, false>), dim3(tiles), dim3(256), 0, st, p);
    WHMR_CHECK_LAUNCH();
    return 0;
}

template <int OUT_BF16, bool GATHER>
static int launch_act(const whmr_gemm& p, hipStream_t st, bool glds) {
    switch (p.act) {
        case 0: return launch<OUT_BF16, 0, GATHER>(p, st, glds);
        case 1: return launch<OUT_BF16, 1, GATHER>(p, st, glds);
        case 2: return launch<OUT_BF16, 2, GATHER>(p, st, glds);
    }
    return (int)hipErrorInvalidValue;
}

// flags bit 0: stage through registers instead of global_load_lds (A/B test + safety fallback inside the HIP path)
extern "C" int whmr_gemm_bf16(const whmr_gemm* pp, int flags, void* stream) {
    const whmr_gemm& p = *pp;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return (int)hipErrorInvalidValue;
    if (p.K % BK) return (int)hipErrorInvalidValue;
    if (p.a_mode == 1 && (p.Cin % BK || !p.zeros)) return (int)hipErrorInvalidValue;
    if (p.a_mode == 0 && (p.lda % 8)) return (int)hipErrorInvalidValue;
    const bool glds = !(flags & 1);
    hipStream_t st = (hipStream_t)stream;
    if (p.a_mode == 1)
        return p.out_bf16 ? launch_act<1, true>(p, st, glds) : launch_act<0, true>(p, st, glds);
    return p.out_bf16 ? launch_act<1, false>(p, st, glds) : launch_act<0, false>(p, st, glds);
}
