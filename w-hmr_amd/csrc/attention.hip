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
// BLK: qkv and out are in the blocked layout of gemm_blk.hip ([rows/32][cols/8][32][8] bf16 over the token rows m = b*N + n): the
// staging loops walk keys fastest (32 consecutive rows of one 16-B column unit are contiguous), and O leaves straight from the
// accumulators in the same (lane = row, 8 consecutive columns) ownership the GEMM epilogue uses -- no LDS transpose.
__device__ __forceinline__ size_t blk_elem(int m, int col8, int ld8) { return ((size_t)(m >> 5) * ld8 + col8) * 256 + (m & 31) * 8; }

template <int NKT, bool BLK = false>
__global__ __launch_bounds__(NKT * 64, 4) void attention_bf16_chunk_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                                           int N, int H, float scale, float* __restrict__ lse, int abl = 0) {
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
    const int ld8 = ld >> 3, m_img = b * N;
    bf16x8_t qf[4];
    if constexpr (BLK) {
        // Every global load of the workgroup's prologue is ISSUED before the first one is consumed: K (4 pieces per thread: NPAD * 8 = 4 x the
        // workgroup), V (2 x 2) and this lane's Q fragments (4) -- 12 x 16 B in flight per thread, ONE memory round trip.  As load -> LDS-store
        // loops hipcc waits vmcnt(0) in every iteration and sinks the Q loads behind the barrier: 4 + 2 + 1 dependent round trips, most of a
        // workgroup's lifetime (round 3: 24.9 -> see DESIGN 6).
        uint4 kreg[4], vreg[2][2];
        const int qr = (wave * 32 + l31 > N - 1) ? N - 1 : wave * 32 + l31;
        if (!(abl & 2)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = tid + i * (NKT * 64), key = c % NPAD, ch = c / NPAD;
                kreg[i] = make_uint4(0, 0, 0, 0);
                if (key < N) kreg[i] = *(const uint4*)(qkv + blk_elem(m_img + key, (C + h * 64) / 8 + ch, ld8));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = tid + i * (NKT * 64), kp = c % (NPAD / 2), ch = c / (NPAD / 2), k0 = 2 * kp;
                vreg[i][0] = vreg[i][1] = make_uint4(0, 0, 0, 0);
                if (k0 < N) vreg[i][0] = *(const uint4*)(qkv + blk_elem(m_img + k0, (2 * C + h * 64) / 8 + ch, ld8));
                if (k0 + 1 < N) vreg[i][1] = *(const uint4*)(qkv + blk_elem(m_img + k0 + 1, (2 * C + h * 64) / 8 + ch, ld8));
            }
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) qf[kk] = *(const bf16x8_t*)(qkv + blk_elem(m_img + qr, (h * 64) / 8 + kk * 2 + hi, ld8));
        if (!(abl & 2)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = tid + i * (NKT * 64), key = c % NPAD, ch = c / NPAD;
                *(uint4*)(Ks + key * 128 + ((ch ^ ((key >> 1) & 7)) << 4)) = kreg[i];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = tid + i * (NKT * 64), kp = c % (NPAD / 2), ch = c / (NPAD / 2), k0 = 2 * kp;
                const uint4 v0 = vreg[i][0], v1 = vreg[i][1];
                const uint32_t a[4] = {v0.x, v0.y, v0.z, v0.w}, bb[4] = {v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    *(uint32_t*)(Vt + (ch * 8 + 2 * e) * VS + k0) = (a[e] & 0xffffu) | (bb[e] << 16);
                    *(uint32_t*)(Vt + (ch * 8 + 2 * e + 1) * VS + k0) = (a[e] >> 16) | (bb[e] & 0xffff0000u);
                }
            }
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) asm volatile("" :: "v"(qf[kk]));        // the Q fragments have landed before the barrier (no sinking)
    } else {
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
    }
    const int q0 = wave * 32;
    int qrow = q0 + l31;
    if (qrow > N - 1) qrow = N - 1;
    if constexpr (!BLK) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) qf[kk] = *(const bf16x8_t*)(base + (size_t)qrow * ld + kk * 16 + hi * 8);
    }
    __syncthreads();

    const float sc = scale * LOG2E;
    float m = -INFINITY, l = 0.f;                 // running max (already in exp2 units) and this half-wave's partial row sum
    f32x16_t o[2];                                // O^T: o[dh][r] = O[q = l31][d = dh*32 + (r&3) + 8*(r>>2) + 4*hi]
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dh][r] = 0.f;
#pragma unroll 1
    for (int c0 = (abl & 4) ? NKT : 0; c0 < NKT; c0 += 2) {
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
    // training: log2-domain log-sum-exp per query, P = exp2(s * scale * log2e - lse) in the backward kernel
    if (lse && hi == 0 && q0 + l31 < N) lse[((size_t)b * H + h) * N + q0 + l31] = m + __log2f(l);
    if constexpr (BLK) {
        // lane = query row, 4 consecutive d per register quad: pack, exchange halves (v_permlane32_swap) -> 8 consecutive d = one 16-B
        // piece of a blocked unit; lanes 0-31 write unit 2p, lanes 32-63 unit 2p + 1 of the head's 8 units
        const int q = q0 + l31;
        if (q < N) {
            bf16_t* orow = out + blk_elem(m_img + q, h * 8, C >> 3);
#pragma unroll
            for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                for (int qq = 0; qq < 4; qq += 2) {
                    const uint32_t a0 = pack_bf16x2(o[dh][4 * qq] * inv, o[dh][4 * qq + 1] * inv), a1 = pack_bf16x2(o[dh][4 * qq + 2] * inv, o[dh][4 * qq + 3] * inv);
                    const uint32_t b0 = pack_bf16x2(o[dh][4 * qq + 4] * inv, o[dh][4 * qq + 5] * inv), b1 = pack_bf16x2(o[dh][4 * qq + 6] * inv, o[dh][4 * qq + 7] * inv);
                    const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                    const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                    *(uint4*)(orow + (size_t)(dh * 4 + qq + hi) * 256) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
                }
        }
        return;
    }
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

// ---- backward of the attention core (bf16, d = 64, N <= 256): dQ, dK, dV from the saved qkv, O, the log-sum-exp written by
// the chunked forward kernel and dO.  One workgroup per (image, head), wave j owns KEY tile j (32 keys):
//   dK_j, dV_j are complete sums over the queries inside the wave (accumulated TRANSPOSED, [d x keys], lane = key);
//   dQ needs a sum over key tiles = over waves: in step s wave j works on query tile (j + s) % NKT, so every query tile is
//   touched by exactly one wave per step and its fp32 accumulator in LDS is updated with plain loads/stores between two
//   barriers -- deterministic, no atomics.
// Per (key tile j, query tile i), all on v_mfma_f32_32x32x16_bf16:
//   S^T = K_j Q_i^T, dP^T = V_j dO_i^T  (lane = query)  ->  dS^T = scale * P^T o (dP^T - D)  ->  dQ_i^T += K_j^T dS^T
//   S   = Q_i K_j^T, dP   = dO_i V_j^T  (lane = key)    ->  P, dS                            ->  dV_j^T += dO_i^T P,  dK_j^T += Q_i^T dS
// (the same operand registers serve S and S^T with the MFMA operands swapped).  The "transposed A" operands (K_j^T, dO_i^T,
// Q_i^T: rows = d, k = keys / queries) are read from the row-major LDS tiles with ds_read_b64_tr_b16.
__device__ __forceinline__ uint32_t swz_addr(int row, int col) {      // byte offset of element (row, col) in a [rows][64] bf16 tile
    return row * 128 + ((((col >> 3) ^ ((row >> 1) & 7))) << 4) + (col & 7) * 2;
}
// A operand "X^T": lane -> m = c0 + (lane & 31) (a column of X), k-slots hi*8 + e <-> X rows R0 + 4*hi + (e & 3) + 8*(e >> 2).  The reads are ISSUED only
// (round 4): the caller waits once for a whole group (tr_wait ties the registers to the wait, so that no consumer can be scheduled above it) -- as
// a read + wait per fragment the backward kernel paid 12 LDS round trips per (key tile, query tile) step.
struct tr_raw { uint2 a, b; };
__device__ __forceinline__ void tr_issue(tr_raw& f, uint32_t lds_base, int R0, int c0, int lane) {
    const int g = lane >> 4, i = lane & 15;
    const int col = c0 + 16 * (g & 1) + 4 * (i & 3);
    const int row = R0 + 4 * (g >> 1) + (i >> 2);
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3" : "=&v"(f.a), "=&v"(f.b) : "v"(lds_base + swz_addr(row, col)), "v"(lds_base + swz_addr(row + 8, col)) : "memory");
}
__device__ __forceinline__ void tr_wait(tr_raw& f0, tr_raw& f1, tr_raw& f2, tr_raw& f3) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f0.a), "+v"(f0.b), "+v"(f1.a), "+v"(f1.b), "+v"(f2.a), "+v"(f2.b), "+v"(f3.a), "+v"(f3.b) :: "memory");
}
__device__ __forceinline__ bf16x8_t tr_value(const tr_raw& f) {
    union { bf16x8_t v; uint32_t u[4]; } o;
    o.u[0] = f.a.x; o.u[1] = f.a.y; o.u[2] = f.b.x; o.u[3] = f.b.y;
    return o.v;
}

#ifndef ATT_LAB
#define ATT_LAB 0          // tools/lab/attn_bwd_lab.hip only: s_memtime stamps of workgroup 300 (per wave: staging, the six phases of step 2, loop end, kernel end)
#endif
#if ATT_LAB
__device__ unsigned long long att_lab_stamps[8 * 16];
#define ATT_STAMP(i) do { if (blockIdx.x == 300 && lane == 0) att_lab_stamps[wave * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ATT_STAMP(i) do { } while (0)
#endif

template <int NKT>
__global__ __launch_bounds__(NKT * 64) void attention_bwd_bf16_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                                       const float* __restrict__ dout, const float* __restrict__ lse,
                                                                       bf16_t* __restrict__ dqkv, int N, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NPAD = NKT * 32;
    constexpr int DQS = 68;                                  // fp32 row stride of the dQ accumulator (16-B aligned, conflict-free)
    char* Qs = smem;                                         // [NPAD][64] bf16, swizzled rows (like Ks of the forward kernel)
    char* dOs = Qs + NPAD * 128;
    char* Ks = dOs + NPAD * 128;
    float* dQa = (float*)(Ks + NPAD * 128);                  // [NPAD][DQS]
    float* lseS = dQa + NPAD * DQS;                          // [NPAD]
    float* Ds = lseS + NPAD;                                 // [NPAD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int C = H * 64, ld = 3 * C;
    const bf16_t* base = qkv + (size_t)b * N * ld + h * 64;
    const float* dob = dout + (size_t)b * N * C + h * 64;
    const bf16_t* ob = o + (size_t)b * N * C + h * 64;
    ATT_STAMP(0);
    // ---- stage Q, K (bf16 rows), dO (fp32 -> bf16), zero the dQ accumulator, per-query lse and D = sum_d dO * O
    // (round 4) D rides along: the 8 lanes of a row hold its 8 chunks of dO (fp32, before the rounding) and fetch the matching chunks of O -- one
    // coalesced pass instead of a second, row-per-THREAD walk over dO and O (64 cache lines per wave instruction).  (Requesting all 20 loads of a
    // thread's four chunks before the first use changes nothing: the pass is bound by the CU's fetch rate for 128-B rows scattered over 4.6 KB
    // strides, 15 000-17 000 of a workgroup's 49 000 cycles: tools/lab/attn_bwd_lab.hip.)
    for (int c = tid; c < NPAD * 8; c += NKT * 64) {
        const int row = c >> 3, ch = c & 7;
        uint4 q = make_uint4(0, 0, 0, 0), k = q, d = q;
        float dsum = 0.f, ls = INFINITY;                     // padded queries: P = exp2(-inf) = 0
        if (row < N) {
            q = *(const uint4*)(base + (size_t)row * ld + ch * 8);
            k = *(const uint4*)(base + (size_t)row * ld + C + ch * 8);
            const float4 a = *(const float4*)(dob + (size_t)row * C + ch * 8), bb = *(const float4*)(dob + (size_t)row * C + ch * 8 + 4);
            const uint4 ov = *(const uint4*)(ob + (size_t)row * C + ch * 8);
            if (ch == 0) ls = lse[((size_t)b * H + h) * N + row];
            d = make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(bb.x, bb.y), pack_bf16x2(bb.z, bb.w));
            dsum = a.x * __uint_as_float(ov.x << 16) + a.y * __uint_as_float(ov.x & 0xffff0000u) +
                   a.z * __uint_as_float(ov.y << 16) + a.w * __uint_as_float(ov.y & 0xffff0000u);
            dsum += bb.x * __uint_as_float(ov.z << 16) + bb.y * __uint_as_float(ov.z & 0xffff0000u) +
                    bb.z * __uint_as_float(ov.w << 16) + bb.w * __uint_as_float(ov.w & 0xffff0000u);
        }
        dsum += __shfl_xor(dsum, 1, 64);
        dsum += __shfl_xor(dsum, 2, 64);
        dsum += __shfl_xor(dsum, 4, 64);
        const int off = row * 128 + ((ch ^ ((row >> 1) & 7)) << 4);
        *(uint4*)(Qs + off) = q;
        *(uint4*)(Ks + off) = k;
        *(uint4*)(dOs + off) = d;
        if (ch == 0) { lseS[row] = ls; Ds[row] = dsum; }
    }
    for (int c = tid; c < NPAD * DQS / 4; c += NKT * 64) ((float4*)dQa)[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    // this wave's key tile: K_j / V_j row fragments (lane = key, k = d) straight from global memory
    const int k0 = wave * 32;
    int krow = k0 + l31;
    const bool key_ok_lane = krow < N;                       // lane-as-key validity (S orientation)
    if (krow > N - 1) krow = N - 1;
    bf16x8_t kf[4], vf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        kf[kk] = *(const bf16x8_t*)(base + (size_t)krow * ld + C + kk * 16 + hi * 8);
        vf[kk] = *(const bf16x8_t*)(base + (size_t)krow * ld + 2 * C + kk * 16 + hi * 8);
    }
    __syncthreads();
    const uint32_t qs_l = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)Qs;
    const uint32_t dos_l = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)dOs;
    const uint32_t ks_l = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)Ks;
    const float sc = scale * LOG2E;
    f32x16_t dvt[2], dkt[2];                                 // dV_j^T, dK_j^T: [d = dh*32 + (r&3) + 8(r>>2) + 4hi][key = l31]
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dvt[dh][r] = 0.f; dkt[dh][r] = 0.f; }

    // K_j^T fragments of the dQ product: the wave's own key tile, the same in every step -- read once (they were re-read, one LDS round trip each,
    // in all NKT steps)
    tr_raw ktr[2][2];
#pragma unroll
    for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
        for (int dh = 0; dh < 2; ++dh) tr_issue(ktr[j2][dh], ks_l, k0 + 16 * j2, dh * 32, lane);
    tr_wait(ktr[0][0], ktr[0][1], ktr[1][0], ktr[1][1]);
    bf16x8_t ktf[2][2];
#pragma unroll
    for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
        for (int dh = 0; dh < 2; ++dh) ktf[j2][dh] = tr_value(ktr[j2][dh]);

    ATT_STAMP(1);
#pragma unroll 1
    for (int step = 0; step < NKT; ++step) {
        const int it = (wave + step) % NKT, q0 = it * 32;
        if (step == 2) ATT_STAMP(2);
        bf16x8_t qf[4], dof[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int row = q0 + l31;
            const int off = row * 128 + (((kk * 2 + hi) ^ ((row >> 1) & 7)) << 4);
            qf[kk] = *(const bf16x8_t*)(Qs + off);
            dof[kk] = *(const bf16x8_t*)(dOs + off);
        }

        // ---- lane = query orientation: dS^T -> dQ_i^T partial
        {
            f32x16_t st, dpt;
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[r] = 0.f; dpt[r] = 0.f; }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kk], qf[kk], st, 0, 0, 0);
                dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[kk], dof[kk], dpt, 0, 0, 0);
            }
            const float ls = lseS[q0 + l31], dq_ = Ds[q0 + l31];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const float p = key < N ? __builtin_amdgcn_exp2f(fmaf(st[r], sc, -ls)) : 0.f;
                st[r] = scale * p * (dpt[r] - dq_);                                  // dS^T
            }
            if (step == 2) ATT_STAMP(3);                                             // S^T / dP^T MFMAs + exponentials done
            f32x16_t dq[2];
#pragma unroll
            for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                for (int r = 0; r < 16; ++r) dq[dh][r] = 0.f;
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) {
                union { bf16x8_t v; uint32_t u[4]; } sf;
#pragma unroll
                for (int e = 0; e < 4; ++e) sf.u[e] = pack_bf16x2(st[8 * j2 + 2 * e], st[8 * j2 + 2 * e + 1]);
#pragma unroll
                for (int dh = 0; dh < 2; ++dh)
                    dq[dh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf[j2][dh], sf.v, dq[dh], 0, 0, 0);
            }
            if (step == 2) ATT_STAMP(4);                                             // dQ MFMAs issued
            float* arow = dQa + (q0 + l31) * DQS;
#pragma unroll
            for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4* ap = (float4*)(arow + dh * 32 + 8 * q + 4 * hi);
                    float4 a = *ap;
                    a.x += dq[dh][4 * q]; a.y += dq[dh][4 * q + 1]; a.z += dq[dh][4 * q + 2]; a.w += dq[dh][4 * q + 3];
                    *ap = a;
                }
        }
        if (step == 2) ATT_STAMP(5);                                                 // dQ accumulated in LDS
        // ---- lane = key orientation: P, dS -> dV_j^T, dK_j^T
        {
            // the transposed dO_i / Q_i fragments of the dV / dK products depend on nothing this step computes: requested here, in front of the S / dP
            // MFMAs and the exponentials (LDS returns in order: the compiler's own counted waits stay valid, merely stricter), awaited once in
            // front of their MFMAs
            tr_raw dotr[2][2], qtr[2][2];
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
                for (int dh = 0; dh < 2; ++dh) {
                    tr_issue(dotr[j2][dh], dos_l, q0 + 16 * j2, dh * 32, lane);
                    tr_issue(qtr[j2][dh], qs_l, q0 + 16 * j2, dh * 32, lane);
                }
            f32x16_t s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[kk], kf[kk], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dof[kk], vf[kk], dp, 0, 0, 0);
            }
            // per-query lse / D of the 16 queries a lane's registers hold: four aligned float4 reads each (as per-register conditionals they were 32
            // scalar LDS reads inside 16 exec-masked blocks)
            float4 ls4[4], d4[4];
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                ls4[k4] = *(const float4*)(lseS + q0 + 8 * k4 + 4 * hi);
                d4[k4] = *(const float4*)(Ds + q0 + 8 * k4 + 4 * hi);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float lsq = (r & 3) == 0 ? ls4[r >> 2].x : (r & 3) == 1 ? ls4[r >> 2].y : (r & 3) == 2 ? ls4[r >> 2].z : ls4[r >> 2].w;
                const float dq_ = (r & 3) == 0 ? d4[r >> 2].x : (r & 3) == 1 ? d4[r >> 2].y : (r & 3) == 2 ? d4[r >> 2].z : d4[r >> 2].w;
                float p = __builtin_amdgcn_exp2f(fmaf(s[r], sc, -lsq));
                p = key_ok_lane ? p : 0.f;
                s[r] = p;                                                            // P
                dp[r] = scale * p * (dp[r] - dq_);                                   // dS
            }
            if (step == 2) ATT_STAMP(6);                                             // S / dP MFMAs + exponentials done
            tr_wait(dotr[0][0], dotr[0][1], dotr[1][0], dotr[1][1]);
            tr_wait(qtr[0][0], qtr[0][1], qtr[1][0], qtr[1][1]);
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) {
                union { bf16x8_t v; uint32_t u[4]; } pf, sf;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pf.u[e] = pack_bf16x2(s[8 * j2 + 2 * e], s[8 * j2 + 2 * e + 1]);
                    sf.u[e] = pack_bf16x2(dp[8 * j2 + 2 * e], dp[8 * j2 + 2 * e + 1]);
                }
#pragma unroll
                for (int dh = 0; dh < 2; ++dh) {
                    dvt[dh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_value(dotr[j2][dh]), pf.v, dvt[dh], 0, 0, 0);
                    dkt[dh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_value(qtr[j2][dh]), sf.v, dkt[dh], 0, 0, 0);
                }
            }
        }
        if (step == 2) ATT_STAMP(7);                                                 // dV / dK MFMAs issued
        __syncthreads();                                     // the next step's owner of this query tile sees the update
        if (step == 2) ATT_STAMP(8);
    }
    ATT_STAMP(9);
    // ---- dQ rows of query tile `wave` (fp32 accumulator -> bf16), 16 lanes per row
    bf16_t* dqb = dqkv + (size_t)b * N * ld + h * 64;
#pragma unroll
    for (int itr = 0; itr < 8; ++itr) {
        const int row = wave * 32 + itr * 4 + (lane >> 4), c4 = (lane & 15) * 4;
        if (row < N) {
            const float4 a = *(const float4*)(dQa + row * DQS + c4);
            *(uint2*)(dqb + (size_t)row * ld + c4) = make_uint2(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w));
        }
    }
    // ---- dK_j, dV_j: transpose [d x keys] -> [keys][d] through LDS (Q / dO tiles are dead after the last barrier)
    constexpr int TS = 136;
    char* tile = smem + wave * (2 * 32 * TS);
#pragma unroll
    for (int which = 0; which < 2; ++which)
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x16_t& a = which ? dvt[dh] : dkt[dh];
                *(uint2*)(tile + which * 32 * TS + l31 * TS + (dh * 32 + 8 * q + 4 * hi) * 2) =
                    make_uint2(pack_bf16x2(a[4 * q], a[4 * q + 1]), pack_bf16x2(a[4 * q + 2], a[4 * q + 3]));
            }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int which = 0; which < 2; ++which)
#pragma unroll
        for (int itr = 0; itr < 4; ++itr) {
            const int row = itr * 8 + (lane >> 3), ch = lane & 7;
            const int key = k0 + row;
            const uint4 v = *(const uint4*)(tile + which * 32 * TS + row * TS + ch * 16);
            if (key < N) *(uint4*)(dqb + (size_t)key * ld + (1 + which) * C + ch * 8) = v;
        }
    ATT_STAMP(10);
}

// ---- fp32 attention on the matrix pipes (parity mode, d = 64, N <= 256): v_mfma_f32_32x32x2_f32 is exact f32 arithmetic, so this is the VALU
// kernel below at ~10x its rate.  One workgroup per (image, head), wave w owns queries 32 w .. 32 w + 31 and walks the key tiles with an online
// softmax.  S^T tile = K_t . Q^T (A operand = K rows from LDS, one float per lane and step; B operand = the wave's Q row, 32 registers, scaled by
// scale * log2 e) lands in the C layout with lane = query, so row max / sum are lane-local (+ one exchange between the half-waves), and the
// probabilities feed the second product straight from the accumulator registers: step r of O^T += V_t^T . P_t^T pairs the keys the two half-waves
// hold in register r ((r & 3) + 8 (r >> 2) + 4 hi), whose V rows are the A operand.  K rows are padded to 66 floats and V rows to 72 so that both
// fragment reads are conflict-free; O goes back through LDS for 256-B row stores.
#define AF_KLD 66
#define AF_VLD 72
template <int NKT>
__global__ __launch_bounds__(NKT * 64) void attention_f32_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ out, int N, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NPAD = NKT * 32;
    float* Ks = (float*)smem;                     // [NPAD][AF_KLD]
    float* Vs = Ks + NPAD * AF_KLD;               // [NPAD][AF_VLD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int C = H * 64, ld = 3 * C;
    const float* base = qkv + (size_t)b * N * ld + h * 64;
    for (int c = tid; c < NPAD * 16; c += NKT * 64) {
        const int key = c >> 4, ch = (c & 15) * 4;
        float4 k = make_float4(0.f, 0.f, 0.f, 0.f), v = k;
        if (key < N) {
            k = *(const float4*)(base + (size_t)key * ld + C + ch);
            v = *(const float4*)(base + (size_t)key * ld + 2 * C + ch);
        }
        float* kd = Ks + key * AF_KLD + ch;
        kd[0] = k.x; kd[1] = k.y; kd[2] = k.z; kd[3] = k.w;                     // rows are 8-byte aligned only (66 floats)
        *(float4*)(Vs + key * AF_VLD + ch) = v;
    }
    const int q0 = wave * 32;
    const int qrow = min(q0 + l31, N - 1);
    const float sc = scale * LOG2E;
    float q[32];                                   // q[s] = Q[qrow][2 s + hi] * scale * log2(e)
#pragma unroll
    for (int s4 = 0; s4 < 16; ++s4) {
        // lanes of one half-wave need the even (hi = 0) or odd (hi = 1) elements: read pairs, keep one
        const float4 t = *(const float4*)(base + (size_t)qrow * ld + s4 * 4);
        q[2 * s4] = (hi ? t.y : t.x) * sc;
        q[2 * s4 + 1] = (hi ? t.w : t.z) * sc;
    }
    __syncthreads();
    float m = -INFINITY, l = 0.f;
    f32x16_t o[2];
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dh][r] = 0.f;
#pragma unroll 1
    for (int t = 0; t < NKT; ++t) {
        f32x16_t sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
        const float* kr = Ks + (t * 32 + l31) * AF_KLD + hi;
#pragma unroll
        for (int s = 0; s < 32; ++s) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[2 * s], q[s], sacc, 0, 0, 0);
        // sacc[r] = S^T[key = 32 t + (r & 3) + 8 (r >> 2) + 4 hi][query = l31]  (already in log2 units)
        float cm = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (key >= N) sacc[r] = -INFINITY;
            cm = fmaxf(cm, sacc[r]);
        }
        cm = fmaxf(cm, __shfl_xor(cm, 32, 64));
        const float m_new = fmaxf(m, cm);
        const float alpha = __builtin_amdgcn_exp2f(m - m_new);                  // 0 on the first tile (m = -inf)
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sacc[r] = __builtin_amdgcn_exp2f(sacc[r] - m_new); ps += sacc[r]; }
        l = fmaf(l, alpha, ps);
        m = m_new;
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dh][r] *= alpha;
        // O^T[d][q] += sum over the tile's keys of V[key][d] P[key][q]: step r pairs keys (r & 3) + 8 (r >> 2) (+ 4 in the upper half-wave)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float* vr = Vs + (t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * AF_VLD + l31;
            o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[0], sacc[r], o[0], 0, 0, 0);
            o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[32], sacc[r], o[1], 0, 0, 0);
        }
    }
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    // o[dh][r] = O[q = l31][d = 32 dh + (r & 3) + 8 (r >> 2) + 4 hi]: through LDS (K / V are free now) for 256-B row stores
    __syncthreads();
    float* tile = (float*)smem + wave * (32 * 65);
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int r = 0; r < 16; ++r) tile[l31 * 65 + dh * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi] = o[dh][r] * inv;
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float* ob = out + (size_t)b * N * C + h * 64;
    for (int row = 0; row < 32; ++row) {
        const int qq = q0 + row;
        if (qq < N) ob[(size_t)qq * C + lane] = tile[row * 65 + lane];
    }
}

template <int NKT>
static int launch_f32_mfma(const float* qkv, float* out, int B, int N, int H, float scale, hipStream_t st) {
    constexpr size_t lds = (size_t)NKT * 32 * (AF_KLD + AF_VLD) * sizeof(float);
    auto kern = attention_f32_mfma_kernel<NKT>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(B * H), dim3(NKT * 64), lds, st, qkv, out, N, H, scale);
    WHMR_CHECK_LAUNCH();
    return 0;
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

static int g_attn_chunked = 1;      // whmr_attention_set_variant: bit 0: 1 = chunked online-softmax kernel (default), 0 = single pass
static int g_attn_abl = 0;          // bits 1-2 (timing probes only, wrong results): 2 = skip the K / V staging, 4 = skip the key loop
static int g_attn_f32_mfma = 1;     // bit 3 SET switches the fp32 attention back to the VALU kernel (A/B, tests)
static int g_attn_blk_old = 0;      // bit 4 SET: the blocked bf16 attention on the round-2 kernel (one workgroup per (image, head), 32-row tiles) instead of
                                    // the persistent 16-row-tile kernel of attention_blk16.hip (A/B, tests)
extern "C" int whmr_attention_set_variant(int v) { g_attn_chunked = v & 1; g_attn_abl = v & 6; g_attn_f32_mfma = !(v & 8); g_attn_blk_old = (v >> 4) & 1; return 0; }
int whmr_attention_blk16_launch(const void* qkv, void* out, int B, int N, int H, float scale, hipStream_t st, int abl);      // attention_blk16.hip

template <int NKT>
static int launch_bf16(const void* qkv, void* out, int B, int N, int H, float scale, hipStream_t st, float* lse = nullptr) {
    constexpr int NPAD = NKT * 32;
    const size_t lds = (size_t)NPAD * 128 + 64 * (NPAD + 4) * 2;
    if ((g_attn_chunked || lse) && NKT >= 3 && scale > 0.f)
        hipLaunchKernelGGL((attention_bf16_chunk_kernel<NKT>), dim3(B * H), dim3(NKT * 64), lds, st, (const bf16_t*)qkv,
                           (bf16_t*)out, N, H, scale, lse);
    else
    hipLaunchKernelGGL((attention_bf16_kernel<NKT>), dim3(B * H), dim3(NKT * 64), lds, st, (const bf16_t*)qkv,
                       (bf16_t*)out, N, H, scale);
    WHMR_CHECK_LAUNCH();
    return 0;
}

template <int NKT>
static int launch_bf16_blk(const void* qkv, void* out, int B, int N, int H, float scale, hipStream_t st) {
    constexpr int NPAD = NKT * 32;
    const size_t lds = (size_t)NPAD * 128 + 64 * (NPAD + 4) * 2;
    hipLaunchKernelGGL((attention_bf16_chunk_kernel<NKT, true>), dim3(B * H), dim3(NKT * 64), lds, st, (const bf16_t*)qkv, (bf16_t*)out, N, H, scale,
                       (float*)nullptr, g_attn_abl);
    WHMR_CHECK_LAUNCH();
    return 0;
}

// Same attention core on the BLOCKED token layout of whmr_gemm_blk: qkv [ceil(B*N/32)][3*H*8][32][8], out [ceil(B*N/32)][H*8][32][8] (bf16, d = 64).
extern "C" int whmr_attention_blk(const void* qkv, void* out, int B, int N, int H, float scale, void* stream) {
    if (B <= 0 || N <= 64 || N > 256 || H <= 0 || scale <= 0.f) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    if (!g_attn_blk_old && N <= 240) return whmr_attention_blk16_launch(qkv, out, B, N, H, scale, st, g_attn_abl);
    switch ((N + 31) / 32) {
        case 3: return launch_bf16_blk<3>(qkv, out, B, N, H, scale, st);
        case 4: return launch_bf16_blk<4>(qkv, out, B, N, H, scale, st);
        case 5: return launch_bf16_blk<5>(qkv, out, B, N, H, scale, st);
        case 6: return launch_bf16_blk<6>(qkv, out, B, N, H, scale, st);
        case 7: return launch_bf16_blk<7>(qkv, out, B, N, H, scale, st);
        case 8: return launch_bf16_blk<8>(qkv, out, B, N, H, scale, st);
    }
    return (int)hipErrorInvalidValue;
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
    if (d == 64 && N > 32 && N <= 256 && scale > 0.f && g_attn_f32_mfma) {       // matrix-pipe form (same exact-f32 arithmetic, ~10x the rate)
        switch ((N + 31) / 32) {
            case 2: return launch_f32_mfma<2>((const float*)qkv, (float*)out, B, N, H, scale, st);
            case 3: return launch_f32_mfma<3>((const float*)qkv, (float*)out, B, N, H, scale, st);
            case 4: return launch_f32_mfma<4>((const float*)qkv, (float*)out, B, N, H, scale, st);
            case 5: return launch_f32_mfma<5>((const float*)qkv, (float*)out, B, N, H, scale, st);
            case 6: return launch_f32_mfma<6>((const float*)qkv, (float*)out, B, N, H, scale, st);
            case 7: return launch_f32_mfma<7>((const float*)qkv, (float*)out, B, N, H, scale, st);
            case 8: return launch_f32_mfma<8>((const float*)qkv, (float*)out, B, N, H, scale, st);
        }
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

template <int NKT>
static int launch_bwd(const void* qkv, const void* o, const float* dout, const float* lse, void* dqkv, int B, int N, int H, float scale,
                      hipStream_t st) {
    constexpr int NPAD = NKT * 32;
    const size_t lds = (size_t)3 * NPAD * 128 + (size_t)NPAD * 68 * 4 + (size_t)2 * NPAD * 4;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)attention_bwd_bf16_kernel<NKT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL((attention_bwd_bf16_kernel<NKT>), dim3(B * H), dim3(NKT * 64), lds, st, (const bf16_t*)qkv, (const bf16_t*)o, dout, lse,
                       (bf16_t*)dqkv, N, H, scale);
    WHMR_CHECK_LAUNCH();
    return 0;
}

// Forward with the per-query log-sum-exp kept for the backward pass (bf16, d = 64, 64 < N <= 256): lse [B, H, N] fp32.
extern "C" int whmr_attention_fwd_train(const void* qkv, void* out, float* lse, int B, int N, int H, int d, float scale, void* stream) {
    if (B <= 0 || N <= 64 || N > 256 || H <= 0 || d != 64 || !lse || scale <= 0.f) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    switch ((N + 31) / 32) {
        case 3: return launch_bf16<3>(qkv, out, B, N, H, scale, st, lse);
        case 4: return launch_bf16<4>(qkv, out, B, N, H, scale, st, lse);
        case 5: return launch_bf16<5>(qkv, out, B, N, H, scale, st, lse);
        case 6: return launch_bf16<6>(qkv, out, B, N, H, scale, st, lse);
        case 7: return launch_bf16<7>(qkv, out, B, N, H, scale, st, lse);
        case 8: return launch_bf16<8>(qkv, out, B, N, H, scale, st, lse);
    }
    return (int)hipErrorInvalidValue;
}

// dqkv [B, N, 3, H, 64] bf16 from qkv (same layout), o = forward output [B, N, H*64] bf16, dout = its gradient (fp32), lse.
extern "C" int whmr_attention_bwd(const void* qkv, const void* o, const float* dout, const float* lse, void* dqkv, int B, int N, int H, int d,
                                  float scale, void* stream) {
    if (B <= 0 || N <= 64 || N > 224 || H <= 0 || d != 64) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    switch ((N + 31) / 32) {
        case 3: return launch_bwd<3>(qkv, o, dout, lse, dqkv, B, N, H, scale, st);
        case 4: return launch_bwd<4>(qkv, o, dout, lse, dqkv, B, N, H, scale, st);
        case 5: return launch_bwd<5>(qkv, o, dout, lse, dqkv, B, N, H, scale, st);
        case 6: return launch_bwd<6>(qkv, o, dout, lse, dqkv, B, N, H, scale, st);
        case 7: return launch_bwd<7>(qkv, o, dout, lse, dqkv, B, N, H, scale, st);
    }
    return (int)hipErrorInvalidValue;
}


// ---- fp32 backward of the attention core (parity mode of the ViT, the 5-token timm Block of the Tz head: any head dim, N <= 256): autograd of
// vit.py:102-111 / timm Attention.  One workgroup per (image, head), deterministic (fixed loop orders, no atomics):
//   phase 1, one wave per query row i:  s_ij = scale q_i.k_j -> p_ij = softmax_j;  dp_ij = dO_i.v_j;  D_i = sum_j p_ij dp_ij;
//            ds_ij = scale p_ij (dp_ij - D_i);  dQ_i = sum_j ds_ij k_j;  the rows p_i, ds_i go to a global scratch [2][N][N] of this (b, h)
//   phase 2, one wave per key row j:    dK_j = sum_i ds_ij q_i,  dV_j = sum_i p_ij dO_i
// K, V (phase 1) and Q, dO (phase 2) sit in LDS with rows padded to d + 1 floats (conflict-free when lane = row).
__global__ __launch_bounds__(256) void attention_bwd_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ dout, float* __restrict__ dqkv,
                                                                float* __restrict__ scratch, int N, int H, int d, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ds1 = d + 1;
    float* A = (float*)smem;                     // [N][d+1]: K, then Q
    float* Bm = A + (size_t)N * ds1;             // [N][d+1]: V, then dO
    float* rowbuf = Bm + (size_t)N * ds1;        // [4 waves][N]: ds row of the wave's current query
    float* qbuf = rowbuf + 4 * N;                // [4 waves][2][d]: q_i and dO_i of the wave's current query
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int C = H * d, ld = 3 * C;
    const float* base = qkv + (size_t)b * N * ld + h * d;
    const float* dob = dout + (size_t)b * N * C + h * d;
    float* dbase = dqkv + (size_t)b * N * ld + h * d;
    float* P = scratch + (size_t)blockIdx.x * 2 * N * N;
    float* DS = P + (size_t)N * N;
    for (int e = tid; e < N * d; e += 256) {
        const int r = e / d, c = e - r * d;
        A[r * ds1 + c] = base[(size_t)r * ld + C + c];
        Bm[r * ds1 + c] = base[(size_t)r * ld + 2 * C + c];
    }
    __syncthreads();
    float* myrow = rowbuf + wave * N;
    float* myq = qbuf + wave * 2 * d;
    constexpr int KPL = 4;                       // keys per lane (N <= 256)
    for (int i = wave; i < N; i += 4) {
        for (int c = lane; c < d; c += 64) { myq[c] = base[(size_t)i * ld + c]; myq[d + c] = dob[(size_t)i * C + c]; }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float s[KPL], dp[KPL];
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < KPL; ++t) {
            const int j = lane + 64 * t;
            float a = 0.f, g = 0.f;
            if (j < N) {
                const float* kr = A + j * ds1;
                const float* vr = Bm + j * ds1;
                for (int c = 0; c < d; ++c) { a = fmaf(myq[c], kr[c], a); g = fmaf(myq[d + c], vr[c], g); }
                a *= scale;
                mx = fmaxf(mx, a);
            }
            s[t] = a; dp[t] = g;
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < KPL; ++t) {
            const int j = lane + 64 * t;
            s[t] = j < N ? expf(s[t] - mx) : 0.f;
            sum += s[t];
        }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
        float dsum = 0.f;
#pragma unroll
        for (int t = 0; t < KPL; ++t) { s[t] *= inv; dsum = fmaf(s[t], dp[t], dsum); }
        dsum = wave_sum(dsum);
#pragma unroll
        for (int t = 0; t < KPL; ++t) {
            const int j = lane + 64 * t;
            if (j < N) {
                const float dsv = scale * s[t] * (dp[t] - dsum);
                myrow[j] = dsv;
                P[(size_t)i * N + j] = s[t];
                DS[(size_t)i * N + j] = dsv;
            }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int c = lane; c < d; c += 64) {
            float acc = 0.f;
            for (int j = 0; j < N; ++j) acc = fmaf(myrow[j], A[j * ds1 + c], acc);
            dbase[(size_t)i * ld + c] = acc;
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();                             // phase-1 scratch rows of this block are complete and visible to the block
    for (int e = tid; e < N * d; e += 256) {
        const int r = e / d, c = e - r * d;
        A[r * ds1 + c] = base[(size_t)r * ld + c];
        Bm[r * ds1 + c] = dob[(size_t)r * C + c];
    }
    __syncthreads();
    for (int j = wave; j < N; j += 4) {
        for (int c = lane; c < d; c += 64) {
            float ak = 0.f, av = 0.f;
            for (int i = 0; i < N; ++i) {
                ak = fmaf(DS[(size_t)i * N + j], A[i * ds1 + c], ak);
                av = fmaf(P[(size_t)i * N + j], Bm[i * ds1 + c], av);
            }
            dbase[(size_t)j * ld + C + c] = ak;
            dbase[(size_t)j * ld + 2 * C + c] = av;
        }
    }
}

// qkv [B, N, 3, H, d] fp32, dout [B, N, H*d] fp32 -> dqkv [B, N, 3, H, d] fp32; scratch >= B*H*2*N*N floats.
extern "C" int whmr_attention_bwd_f32(const float* qkv, const float* dout, float* dqkv, float* scratch, int B, int N, int H, int d, float scale,
                                      void* stream) {
    if (B <= 0 || N <= 0 || N > 256 || H <= 0 || d <= 0) return (int)hipErrorInvalidValue;
    const size_t lds = ((size_t)2 * N * (d + 1) + 4 * N + 8 * d) * sizeof(float);
    if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)attention_bwd_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(attention_bwd_f32_kernel, dim3(B * H), dim3(256), lds, (hipStream_t)stream, qkv, dout, dqkv, scratch, N, H, d, scale);
    WHMR_CHECK_LAUNCH();
    return 0;
}
