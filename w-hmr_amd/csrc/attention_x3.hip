// Attention core of the "bf16x3" (split-bf16) numerics on the BLOCKED token layout:  O = softmax(scale * Q K^T) V  per (image, head),
// vit.py:102-111, with every operand a PAIR of bf16 tensors (x = x_hi + x_lo, 16 significand bits) and every product three MFMAs
// (hi.hi + hi.lo + lo.hi, fp32 accumulate) -- fp32-grade results on the bf16 matrix pipes.  Same structure as
// attention_bf16_chunk_kernel<NKT, BLK = true> (attention.hip): one workgroup per (image, head), one wave per 32 queries, K and V^T (hi and lo)
// in LDS, S^T = K.Q^T so that a lane owns one query (softmax lane-local, online over chunks of 64 keys), the probabilities are split into
// hi / lo in registers and feed O^T += V^T.P^T straight from the accumulator layout; O leaves as a hi / lo pair in the GEMM epilogue's
// (lane = row, 8 consecutive columns) ownership.  qkv_{hi,lo}: [ceil(B*N/32)][3*H*8][32][8], out_{hi,lo}: [ceil(B*N/32)][H*8][32][8].
#include "common.h"

#define LOG2E 1.4426950408889634f

__device__ __forceinline__ size_t x3_blk_elem(int m, int col8, int ld8) { return ((size_t)(m >> 5) * ld8 + col8) * 256 + (m & 31) * 8; }

template <int NKT>
__global__ __launch_bounds__(NKT * 64, 1) void attention_x3_blk_kernel(const bf16_t* __restrict__ qkv_hi, const bf16_t* __restrict__ qkv_lo,
                                                                       bf16_t* __restrict__ out_hi, bf16_t* __restrict__ out_lo, int N, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NPAD = NKT * 32;
    constexpr int VS = NPAD + 4;
    constexpr int KB = NPAD * 128, VB = 64 * VS * 2;          // bytes of one K image / one V^T image
    char* Ks[2] = {smem, smem + KB};
    bf16_t* Vt[2] = {(bf16_t*)(smem + 2 * KB), (bf16_t*)(smem + 2 * KB + VB)};
    const bf16_t* src[2] = {qkv_hi, qkv_lo};
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int C = H * 64, ld8 = (3 * C) >> 3, m_img = b * N;
    // Prologue: every global load (K 2 x 4 pieces per thread, V 2 x 2 x 2, the lane's Q fragments 2 x 4) is issued before the first one is
    // consumed -- one memory round trip instead of one per staging-loop iteration (this kernel runs ONE workgroup per CU: nothing else hides them).
    const int q0 = wave * 32;
    const int qrow = (q0 + l31 > N - 1) ? N - 1 : q0 + l31;
    uint4 kreg[2][4], vreg[2][2][2];
    bf16x8_t qf[2][4];
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        const bf16_t* qkv = src[part];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + i * (NKT * 64), key = c % NPAD, ch = c / NPAD;
            kreg[part][i] = make_uint4(0, 0, 0, 0);
            if (key < N) kreg[part][i] = *(const uint4*)(qkv + x3_blk_elem(m_img + key, (C + h * 64) / 8 + ch, ld8));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + i * (NKT * 64), kp = c % (NPAD / 2), ch = c / (NPAD / 2), k0 = 2 * kp;
            vreg[part][i][0] = vreg[part][i][1] = make_uint4(0, 0, 0, 0);
            if (k0 < N) vreg[part][i][0] = *(const uint4*)(qkv + x3_blk_elem(m_img + k0, (2 * C + h * 64) / 8 + ch, ld8));
            if (k0 + 1 < N) vreg[part][i][1] = *(const uint4*)(qkv + x3_blk_elem(m_img + k0 + 1, (2 * C + h * 64) / 8 + ch, ld8));
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) qf[part][kk] = *(const bf16x8_t*)(qkv + x3_blk_elem(m_img + qrow, (h * 64) / 8 + kk * 2 + hi, ld8));
    }
#pragma unroll
    for (int part = 0; part < 2; ++part) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + i * (NKT * 64), key = c % NPAD, ch = c / NPAD;
            *(uint4*)(Ks[part] + key * 128 + ((ch ^ ((key >> 1) & 7)) << 4)) = kreg[part][i];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + i * (NKT * 64), kp = c % (NPAD / 2), ch = c / (NPAD / 2), k0 = 2 * kp;
            const uint4 v0 = vreg[part][i][0], v1 = vreg[part][i][1];
            const uint32_t a[4] = {v0.x, v0.y, v0.z, v0.w}, bb[4] = {v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                *(uint32_t*)(Vt[part] + (ch * 8 + 2 * e) * VS + k0) = (a[e] & 0xffffu) | (bb[e] << 16);
                *(uint32_t*)(Vt[part] + (ch * 8 + 2 * e + 1) * VS + k0) = (a[e] >> 16) | (bb[e] & 0xffff0000u);
            }
        }
    }
#pragma unroll
    for (int part = 0; part < 2; ++part)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) asm volatile("" :: "v"(qf[part][kk]));   // landed before the barrier (no sinking behind it)
    __syncthreads();

    const float sc = scale * LOG2E;
    float m = -INFINITY, l = 0.f;
    f32x16_t o[2];                                // O^T: o[dh][r] = O[q = l31][d = dh*32 + (r&3) + 8*(r>>2) + 4*hi]
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dh][r] = 0.f;
#pragma unroll 1
    for (int c0 = 0; c0 < NKT; c0 += 2) {
        const int nt = (NKT - c0) < 2 ? (NKT - c0) : 2;
        f32x16_t s[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (t < nt) {
                    const int key = (c0 + t) * 32 + l31;
                    const int off = key * 128 + (((kk * 2 + hi) ^ ((key >> 1) & 7)) << 4);
                    const bf16x8_t kh = *(const bf16x8_t*)(Ks[0] + off), kl = *(const bf16x8_t*)(Ks[1] + off);
                    s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qf[0][kk], s[t], 0, 0, 0);
                    s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qf[1][kk], s[t], 0, 0, 0);
                    s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qf[0][kk], s[t], 0, 0, 0);
                }
            }
        }
        float cm = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (t < nt) {
                    const int key = (c0 + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (c0 + t == NKT - 1 && key >= N) s[t][r] = -INFINITY;
                    cm = fmaxf(cm, s[t][r]);
                }
            }
        cm = fmaxf(cm, __shfl_xor(cm, 32, 64));
        const float m_new = fmaxf(m, cm * sc);
        const float alpha = exp2f(m - m_new);                            // 0 on the first chunk (m = -inf)
        float ps = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (t < nt) {
                    const float e = exp2f(fmaf(s[t][r], sc, -m_new));
                    s[t][r] = e;
                    ps += e;
                }
            }
        l = fmaf(l, alpha, ps);
        m = m_new;
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dh][r] *= alpha;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t < nt) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    union { bf16x8_t v; uint32_t u[4]; } ph, pl;
#pragma unroll
                    for (int i = 0; i < 4; ++i) split_bf16x2(s[t][8 * j + 2 * i], s[t][8 * j + 2 * i + 1], ph.u[i], pl.u[i]);
#pragma unroll
                    for (int dh = 0; dh < 2; ++dh) {
                        const int voff = (dh * 32 + l31) * VS + (c0 + t) * 32 + 16 * j + 4 * hi;
                        union { bf16x8_t v; uint2 u[2]; } vh, vl;
                        vh.u[0] = *(const uint2*)(Vt[0] + voff);
                        vh.u[1] = *(const uint2*)(Vt[0] + voff + 8);
                        vl.u[0] = *(const uint2*)(Vt[1] + voff);
                        vl.u[1] = *(const uint2*)(Vt[1] + voff + 8);
                        o[dh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl.v, ph.v, o[dh], 0, 0, 0);
                        o[dh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh.v, pl.v, o[dh], 0, 0, 0);
                        o[dh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh.v, ph.v, o[dh], 0, 0, 0);
                    }
                }
            }
        }
    }
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    // lane = query row, 4 consecutive d per register quad: split, exchange halves (v_permlane32_swap) -> 8 consecutive d = one 16-B piece
    // of a blocked unit; lanes 0-31 write unit 2p, lanes 32-63 unit 2p + 1 of the head's 8 units
    const int q = q0 + l31;
    if (q < N) {
        const size_t ooff = x3_blk_elem(m_img + q, h * 8, C >> 3);
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int qq = 0; qq < 4; qq += 2) {
                uint32_t a0, a1, b0, b1, la0, la1, lb0, lb1;
                split_bf16x2(o[dh][4 * qq] * inv, o[dh][4 * qq + 1] * inv, a0, la0);
                split_bf16x2(o[dh][4 * qq + 2] * inv, o[dh][4 * qq + 3] * inv, a1, la1);
                split_bf16x2(o[dh][4 * qq + 4] * inv, o[dh][4 * qq + 5] * inv, b0, lb0);
                split_bf16x2(o[dh][4 * qq + 6] * inv, o[dh][4 * qq + 7] * inv, b1, lb1);
                const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                *(uint4*)(out_hi + ooff + (size_t)(dh * 4 + qq + hi) * 256) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
                const auto s0 = __builtin_amdgcn_permlane32_swap(la0, lb0, false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(la1, lb1, false, false);
                *(uint4*)(out_lo + ooff + (size_t)(dh * 4 + qq + hi) * 256) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
            }
    }
}

template <int NKT>
static int launch_x3(const void* qh, const void* ql, void* oh, void* ol, int B, int N, int H, float scale, hipStream_t st) {
    constexpr int NPAD = NKT * 32;
    const size_t lds = 2 * ((size_t)NPAD * 128 + 64 * (NPAD + 4) * 2);
    auto kern = attention_x3_blk_kernel<NKT>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(B * H), dim3(NKT * 64), lds, st, (const bf16_t*)qh, (const bf16_t*)ql, (bf16_t*)oh, (bf16_t*)ol, N, H, scale);
    WHMR_CHECK_LAUNCH();
    return 0;
}

int whmr_attention_blk16_x3_launch(const void* qh, const void* ql, void* oh, void* ol, int B, int N, int H, float scale, hipStream_t st, int lab);      // attention_blk16.hip
static int g_x3_old = 0;       // whmr_set_option(120, 1): the round-3 kernel above instead of the persistent 16-row-tile one (A/B, tests)
extern "C" int whmr_attention_x3_set_variant(int variant) { g_x3_old = variant; return 0; }

extern "C" int whmr_attention_blk_x3(const void* qkv_hi, const void* qkv_lo, void* out_hi, void* out_lo, int B, int N, int H, float scale, void* stream) {
    if (B <= 0 || N <= 64 || N > 256 || H <= 0 || scale <= 0.f || !qkv_lo || !out_lo) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    if (!(g_x3_old & 1) && N <= 208) return whmr_attention_blk16_x3_launch(qkv_hi, qkv_lo, out_hi, out_lo, B, N, H, scale, st, g_x3_old >> 4);
    switch ((N + 31) / 32) {
        case 3: return launch_x3<3>(qkv_hi, qkv_lo, out_hi, out_lo, B, N, H, scale, st);
        case 4: return launch_x3<4>(qkv_hi, qkv_lo, out_hi, out_lo, B, N, H, scale, st);
        case 5: return launch_x3<5>(qkv_hi, qkv_lo, out_hi, out_lo, B, N, H, scale, st);
        case 6: return launch_x3<6>(qkv_hi, qkv_lo, out_hi, out_lo, B, N, H, scale, st);
        case 7: return launch_x3<7>(qkv_hi, qkv_lo, out_hi, out_lo, B, N, H, scale, st);
        case 8: return launch_x3<8>(qkv_hi, qkv_lo, out_hi, out_lo, B, N, H, scale, st);
    }
    return (int)hipErrorInvalidValue;
}
