// Multi-head self-attention core:  O = softmax(scale * Q K^T) V  per (image, head).
// Replaces vit.py:102-111 (q*scale; q@k^T; softmax; @v; transpose/reshape) -- the qkv / proj Linears are GEMMs.
// Input is the qkv GEMM output [B, N, 3, H, d] (exactly the reshape at vit.py:102), output [B, N, H*d].
//
// bf16 kernel (d = 64, N <= 256): one workgroup per (b, h), one wave per 32 query rows (N=196 -> 7 waves, N=192 -> 6).
//   K [N,64] sits in LDS (XOR-swizzled 16-B chunks), V sits TRANSPOSED in LDS (Vt[d][key], key-contiguous).
//   S^T = K.Q^T on v_mfma_f32_32x32x16_bf16 ("swapped QK^T"): each lane then owns one query column, so the
//   softmax is in-register (+1 cross-half shuffle) and P never touches LDS: the C/D register order of S^T is used
//   directly as the k-slot order of the P.V MFMA, with V fragments read from Vt in the matching key permutation.
// fp32 kernel (any d, N <= 256): VALU reference-order arithmetic for the parity mode and the tiny Tz-head block.
#include "common.h"

#define LOG2E 1.4426950408889634f

template <int NKT>
__global__ __launch_bounds__(NKT * 64) void attention_bf16_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                                   int N, int H, float scale_in) {
    float scale = scale_in;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NPAD = NKT * 32;
    constexpr int VS = NPAD + 4;                 // Vt row stride (elements): dword stride = 2*odd -> conflict-free b64 reads
    char* Ks = smem;                             // [NPAD][64] bf16, 128 B rows, chunk ^= (row>>1)&7
    bf16_t* Vt = (bf16_t*)(smem + NPAD * 128);   // [64][VS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int C = H * 64, ld = 3 * C;
    const bf16_t* base = qkv + (size_t)b * N * ld + h * 64;

    int probe = 0;
    if (scale < 0.f) { probe = (int)(-scale); scale = 0.125f; }
    // ---- stage K (swizzled) and V (transposed) into LDS
    if (probe != 1)
    for (int c = tid; c < NPAD * 8; c += NKT * 64) {
        const int key = c >> 3, ch = c & 7;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (key < N) v = *(const uint4*)(base + (size_t)key * ld + C + ch * 8);
        *(uint4*)(Ks + key * 128 + ((ch ^ ((key >> 1) & 7)) << 4)) = v;
    }
    if (probe != 1)
    for (int c = tid; c < (NPAD / 2) * 8; c += NKT * 64) {
        const int kp = c >> 3, ch = c & 7;
        const int k0 = 2 * kp;
        uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0;
        if (k0 < N) v0 = *(const uint4*)(base + (size_t)k0 * ld + 2 * C + ch * 8);
        if (k0 + 1 < N) v1 = *(const uint4*)(base + (size_t)(k0 + 1) * ld + 2 * C + ch * 8);
        const uint32_t a[4] = {v0.x, v0.y, v0.z, v0.w}, bb[4] = {v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t lo = (a[i] & 0xffffu) | (bb[i] << 16);
            const uint32_t hi2 = (a[i] >> 16) | (bb[i] & 0xffff0000u);
            *(uint32_t*)(Vt + (ch * 8 + 2 * i) * VS + k0) = lo;
            *(uint32_t*)(Vt + (ch * 8 + 2 * i + 1) * VS + k0) = hi2;
        }
    }

    // ---- Q fragments straight from global memory (B operand: lane holds Q[q = l31][d = kk*16 + hi*8 ..])
    const int q0 = wave * 32;
    int qrow = q0 + l31;
    if (qrow > N - 1) qrow = N - 1;
    bf16x8_t qf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[kk] = *(const bf16x8_t*)(base + (size_t)qrow * ld + kk * 16 + hi * 8);
    __syncthreads();
    if (probe == 2) { if (qf[0][0] == 12345) out[0] = 1; return; }

    // ---- S^T = K . Q^T : s[kt][r] = S[q = l31][key = kt*32 + (r&3) + 8*(r>>2) + 4*hi].  kk is the OUTER loop so that
    // consecutive MFMAs hit different accumulators (7 independent chains) instead of one dependent chain per tile.
    f32x16_t s[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        bf16x8_t kf[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            const int key = kt * 32 + l31;
            kf[kt] = *(const bf16x8_t*)(Ks + key * 128 + (((kk * 2 + hi) ^ ((key >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kt], qf[kk], s[kt], 0, 0, 0);
    }

    // ---- softmax over keys (in-lane over kt, r; one exchange with the other half-wave).  Only the last key tile can
    // hold padded keys; the scale is folded into the exp2 argument (scale > 0, so the max commutes with it).
    const float sc = scale * LOG2E;
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (kt == NKT - 1 && key >= N) s[kt][r] = -INFINITY;
            mx = fmaxf(mx, s[kt][r]);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * sc;
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f(fmaf(s[kt][r], sc, -mx));
            s[kt][r] = e;
            sum += e;
        }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;

    // ---- O = P . V : A = P (k-slot hi*8+i <-> key 16j + 4hi + (i&3) + 8(i>>2)), B = Vt in the same key order
    f32x16_t o[2];
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dh][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            union { bf16x8_t v; uint32_t u[4]; } pf;
#pragma unroll
            for (int i = 0; i < 4; ++i) pf.u[i] = pack_bf16x2(s[kt][8 * j + 2 * i], s[kt][8 * j + 2 * i + 1]);
#pragma unroll
            for (int dh = 0; dh < 2; ++dh) {
                const bf16_t* vrow = Vt + (dh * 32 + l31) * VS + kt * 32 + 16 * j + 4 * hi;
                union { bf16x8_t v; uint2 u[2]; } vf;
                vf.u[0] = *(const uint2*)(vrow);
                vf.u[1] = *(const uint2*)(vrow + 8);
                o[dh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf.v, vf.v, o[dh], 0, 0, 0);
            }
        }
    }

    // ---- normalise rows (1/sum lives in the lane whose l31 == query row) and store
    bf16_t* obase = out + (size_t)b * N * C + h * 64;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ql = (r & 3) + 8 * (r >> 2) + 4 * hi;
        const float iv = __shfl(inv, ql, 64);
        const int q = q0 + ql;
        if (q < N) {
            obase[(size_t)q * C + l31] = f32_to_bf16(o[0][r] * iv);
            obase[(size_t)q * C + 32 + l31] = f32_to_bf16(o[1][r] * iv);
        }
    }
}

// ---- chunked variant: keys are processed two 32-key tiles at a time with an online softmax, so only 2 score tiles (32
// registers) are live instead of NKT (112 at N = 196): < 128 VGPRs -> 4 waves / SIMD -> TWO workgroups per CU (the
// single-pass kernel above allocates 176 and runs one 7-wave workgroup per CU, i.e. 768 workgroups in 3 rounds).
// O is accumulated TRANSPOSED (O^T = V^T . P^T, i.e. the two MFMA operands above swapped): each lane then owns one query
// column of O^T, so the running rescale exp2(m_old - m_new) and the final 1/l are lane-local; the tile is transposed back
// through LDS (the K region, free after the last key tile) for 128-B row stores.
template <int NKT>
__global__ __launch_bounds__(NKT * 64, 4) void attention_bf16_chunk_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                                           int N, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NPAD = NKT * 32;
    constexpr int VS = NPAD + 4;
    char* Ks = smem;
    bf16_t* Vt = (bf16_t*)(smem + NPAD * 128);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int C = H * 64, ld = 3 * C;
    const bf16_t* base = qkv + (size_t)b * N * ld + h * 64;
    for (int c = tid; c < NPAD * 8; c += NKT * 64) {
        const int key = c >> 3, ch = c & 7;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (key < N) v = *(const uint4*)(base + (size_t)key * ld + C + ch * 8);
        *(uint4*)(Ks + key * 128 + ((ch ^ ((key >> 1) & 7)) << 4)) = v;
    }
    for (int c = tid; c < (NPAD / 2) * 8; c += NKT * 64) {
        const int kp = c >> 3, ch = c & 7;
        const int k0 = 2 * kp;
        uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0;
        if (k0 < N) v0 = *(const uint4*)(base + (size_t)k0 * ld + 2 * C + ch * 8);
        if (k0 + 1 < N) v1 = *(const uint4*)(base + (size_t)(k0 + 1) * ld + 2 * C + ch * 8);
        const uint32_t a[4] = {v0.x, v0.y, v0.z, v0.w}, bb[4] = {v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *(uint32_t*)(Vt + (ch * 8 + 2 * i) * VS + k0) = (a[i] & 0xffffu) | (bb[i] << 16);
            *(uint32_t*)(Vt + (ch * 8 + 2 * i + 1) * VS + k0) = (a[i] >> 16) | (bb[i] & 0xffff0000u);
        }
    }
    const int q0 = wave * 32;
    int qrow = q0 + l31;
    if (qrow > N - 1) qrow = N - 1;
    bf16x8_t qf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[kk] = *(const bf16x8_t*)(base + (size_t)qrow * ld + kk * 16 + hi * 8);
    __syncthreads();

    const float sc = scale * LOG2E;
    float m = -INFINITY, l = 0.f;                 // running max (already in exp2 units) and this half-wave's partial row sum
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
                    const bf16x8_t kf = *(const bf16x8_t*)(Ks + key * 128 + (((kk * 2 + hi) ^ ((key >> 1) & 7)) << 4));
                    s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], s[t], 0, 0, 0);
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
        const float alpha = __builtin_amdgcn_exp2f(m - m_new);          // 0 on the first chunk (m = -inf)
        float ps = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (t < nt) {
                    const float e = __builtin_amdgcn_exp2f(fmaf(s[t][r], sc, -m_new));
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
                    union { bf16x8_t v; uint32_t u[4]; } pf;
#pragma unroll
                    for (int i = 0; i < 4; ++i) pf.u[i] = pack_bf16x2(s[t][8 * j + 2 * i], s[t][8 * j + 2 * i + 1]);
#pragma unroll
                    for (int dh = 0; dh < 2; ++dh) {
                        const bf16_t* vrow = Vt + (dh * 32 + l31) * VS + (c0 + t) * 32 + 16 * j + 4 * hi;
                        union { bf16x8_t v; uint2 u[2]; } vf;
                        vf.u[0] = *(const uint2*)(vrow);
                        vf.u[1] = *(const uint2*)(vrow + 8);
                        o[dh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.v, pf.v, o[dh], 0, 0, 0);
                    }
                }
            }
        }
    }
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    // ---- transpose the wave's [32 queries x 64 d] tile through LDS (rows of 136 B: conflict-free 8-B writes) and store rows
    __syncthreads();                               // every wave is done with K / Vt
    constexpr int TS = 136;
    char* tile = smem + wave * (32 * TS);
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *(uint2*)(tile + l31 * TS + (dh * 32 + 8 * q + 4 * hi) * 2) =
                make_uint2(pack_bf16x2(o[dh][4 * q] * inv, o[dh][4 * q + 1] * inv), pack_bf16x2(o[dh][4 * q + 2] * inv, o[dh][4 * q + 3] * inv));
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    bf16_t* obase = out + (size_t)b * N * C + h * 64;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + (lane >> 3), ch = lane & 7;
        const int q = q0 + row;
        const uint4 v = *(const uint4*)(tile + row * TS + ch * 16);
        if (q < N) *(uint4*)(obase + (size_t)q * C + ch * 8) = v;
    }
}

// fp32, reference operation order: q*scale, dot over d, softmax(expf), weighted sum over keys.
__global__ __launch_bounds__(256) void attention_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                            int N, int H, int d, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ds = d + 1;
    float* Ks = (float*)smem;            // [N][d+1]
    float* Vs = Ks + (size_t)N * ds;     // [N][d+1]
    float* qs = Vs + (size_t)N * ds;     // [4][d]
    float* ps = qs + 4 * d;              // [4][N]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int C = H * d, ld = 3 * C;
    const float* base = qkv + (size_t)b * N * ld + h * d;
    for (int i = tid; i < N * d; i += 256) {
        const int key = i / d, dd = i - key * d;
        Ks[key * ds + dd] = base[(size_t)key * ld + C + dd];
        Vs[key * ds + dd] = base[(size_t)key * ld + 2 * C + dd];
    }
    __syncthreads();
    float* q = qs + wave * d;
    float* p = ps + wave * N;
    for (int row = wave; row < N; row += 4) {
        for (int dd = lane; dd < d; dd += 64) q[dd] = base[(size_t)row * ld + dd] * scale;
        __builtin_amdgcn_wave_barrier();
        float sv[4];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key = lane + 64 * i;
            float a = -INFINITY;
            if (key < N) {
                a = 0.f;
                const float* kr = Ks + key * ds;
                for (int dd = 0; dd < d; ++dd) a = fmaf(q[dd], kr[dd], a);
            }
            sv[i] = a;
            mx = fmaxf(mx, a);
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key = lane + 64 * i;
            if (key < N) { sv[i] = expf(sv[i] - mx); sum += sv[i]; }
        }
        sum = wave_sum(sum);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key = lane + 64 * i;
            if (key < N) p[key] = sv[i] / sum;
        }
        __builtin_amdgcn_wave_barrier();
        for (int dd = lane; dd < d; dd += 64) {
            float a = 0.f;
            for (int key = 0; key < N; ++key) a = fmaf(p[key], Vs[key * ds + dd], a);
            out[((size_t)b * N + row) * C + h * d + dd] = a;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

static int g_attn_chunked = 1;      // whmr_attention_set_variant: 1 = chunked online-softmax kernel (default), 0 = single pass
extern "C" int whmr_attention_set_variant(int chunked) { g_attn_chunked = chunked; return 0; }

template <int NKT>
static int launch_bf16(const void* qkv, void* out, int B, int N, int H, float scale, hipStream_t st) {
    constexpr int NPAD = NKT * 32;
    const size_t lds = (size_t)NPAD * 128 + 64 * (NPAD + 4) * 2;
    if (g_attn_chunked && NKT >= 3 && scale > 0.f)
        hipLaunchKernelGGL((attention_bf16_chunk_kernel<NKT>), dim3(B * H), dim3(NKT * 64), lds, st, (const bf16_t*)qkv,
                           (bf16_t*)out, N, H, scale);
    else
    hipLaunchKernelGGL((attention_bf16_kernel<NKT>), dim3(B * H), dim3(NKT * 64), lds, st, (const bf16_t*)qkv,
                       (bf16_t*)out, N, H, scale);
    WHMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int whmr_attention(const void* qkv, void* out, int B, int N, int H, int d, float scale, int is_bf16,
                              void* stream) {
    if (B <= 0 || N <= 0 || N > 256 || H <= 0 || d <= 0) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    if (is_bf16) {
        if (d != 64) return (int)hipErrorInvalidValue;
        switch ((N + 31) / 32) {
            case 1: return launch_bf16<1>(qkv, out, B, N, H, scale, st);
            case 2: return launch_bf16<2>(qkv, out, B, N, H, scale, st);
            case 3: return launch_bf16<3>(qkv, out, B, N, H, scale, st);
            case 4: return launch_bf16<4>(qkv, out, B, N, H, scale, st);
            case 5: return launch_bf16<5>(qkv, out, B, N, H, scale, st);
            case 6: return launch_bf16<6>(qkv, out, B, N, H, scale, st);
            case 7: return launch_bf16<7>(qkv, out, B, N, H, scale, st);
            case 8: return launch_bf16<8>(qkv, out, B, N, H, scale, st);
        }
        return (int)hipErrorInvalidValue;
    }
    const size_t lds = ((size_t)2 * N * (d + 1) + 4 * d + 4 * N) * sizeof(float);
    if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)attention_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(attention_f32_kernel, dim3(B * H), dim3(256), lds, st, (const float*)qkv, (float*)out, N, H, d, scale);
    WHMR_CHECK_LAUNCH();
    return 0;
}
