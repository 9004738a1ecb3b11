// whmr_gemm_blk_desc: bf16 MFMA GEMM on BLOCKED operand layouts, two wave groups in ping-pong.  The bf16 inference path of the ViT
// (vit.py:61-140: qkv / proj / fc1 + GELU / fc2, vit.py:157 patch embed) runs on this kernel; the row-major kernel of
// gemm_bf16_big.hip stays for the convolutions, the training graph and odd shapes.
//
// Layout ("blocked"): a [R, C] matrix is stored as [R/32][C/E][32][E] with E = 8 (bf16) or 4 (fp32), i.e. 512-byte units of
// 32 rows x 16 bytes.  That unit is at once
//   * what one half-wave of an MFMA 32x32x16 operand fetch reads (lane = row, 8 consecutive k),
//   * what one half-wave of the (operand-swapped) MFMA result owns (lane = row, 4 fp32 / 8 bf16 consecutive columns),
//   * a contiguous 512 B of global memory AND of LDS.
// Consequences: global_load_lds copies whole 1-KiB runs (no swizzle: the ds_read_b128 fragment reads of a linear image are
// conflict-free), and the epilogue stores straight from the accumulators in 1-KiB wave stores -- no LDS transpose, no barrier.
//
// Schedule (measured in tools/lab/gemm_lab.hip): 8 waves = 2 wave rows x 4 wave columns, wave tile (32 MI) x 64.  The two
// wave rows are two GROUPS (one wave of each per SIMD) that run one s_barrier apart.  Work unit = half a K tile (32 deep):
//   MEM(h)  : ds_read this wave's fragments of half tile h (2 MI + 4 reads), issue its share of the LDS-DMA of half tile h + 3
//             (ring of 4 half-tile slots, counted vmcnt: two younger groups stay in flight), wait
//   MFMA(h) : 4 MI MFMAs straight from registers
// with ONE s_barrier per half tile and the two groups walking each slot in opposite order (group 0: MFMA then MEM, group 1: MEM then
// MFMA): on every SIMD one wave feeds the matrix pipe while the other talks to LDS and the texture addresser.  In a lock-step loop all
// 8 waves issue their DMA at the same time and the matrix pipes idle for the ~1000 clk the L1 needs to take 64 KiB (qkv main loop
// 48 -> 36-38 us in the lab).
#include <type_traits>
#include "common.h"
#include "gemm_blk.h"

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

template <int N> __device__ __forceinline__ void blk_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void blk_wait_lgkmcnt() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
// fragment read hidden from hipcc's waitcnt bookkeeping (valid after the counted wait + sched_barrier that follows it)
template <int OFF> __device__ __forceinline__ bf16x8_t blk_lds_read128(uint32_t addr) {
    bf16x8_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

template <int MI0, int MI1>
struct blk_cfg {
    static constexpr int MB = MI0 + MI1;                 // A row blocks (32 rows) per tile
    static constexpr int BM = MB * 32, BN = 256;
    static constexpr int SLOT = (MB + 8) * 2048;         // one half K tile (32 deep) of A and W: 2 KiB per row block
    static constexpr int HU = (MB + 8) * 2;              // 1-KiB DMA units per half tile
    static constexpr int HUPW = (HU + 7) / 8;            // units per wave (waves >= HU % 8 issue one less when HU % 8 != 0)
    static constexpr int BIAS_OFF = 4 * SLOT;            // [256] floats behind the ring, then [256] floats of the LayerNorm-fold column sums
    static constexpr int STAT_OFF = BIAS_OFF + 2048;     // LayerNorm folding: [BM][4][2] floats -- row statistics (consumer) / per-wave-column partial sums (producer)
    static constexpr int LDS = 4 * SLOT + 2048 + BM * 32;
    static constexpr int MIMAX = MI0 > MI1 ? MI0 : MI1;
};

// SCHED 1: one barrier per half tile, groups in opposite order within a slot;  SCHED 0: two barriers per half tile (MEM | MFMA rendezvous)
template <int MI0, int MI1, int EPI, int SCHED>
__global__ __launch_bounds__(512, 2) void gemm_blk_kernel(const whmr_gemm_blk_desc p) {
    using cfg = blk_cfg<MI0, MI1>;
    constexpr int MB = cfg::MB, BM = cfg::BM, BN = cfg::BN, SLOT = cfg::SLOT, HU = cfg::HU, HUPW = cfg::HUPW, NJ = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;              // wm = group
    const int l31 = lane & 31, hi = lane >> 5;
    const int tiles_n = p.N / BN, tiles_m = (p.M + BM - 1) / BM;
    const int lid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = lid / tiles_n, tn = lid % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int KC = p.K >> 3;                              // 16-B chunks per row
    const int H = p.K >> 5;                               // half K tiles
    const int rb_last = ((p.M + 31) >> 5) - 1;            // last valid row block
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_t*)smem;

    // this tile's bias slice -> LDS (one float per thread, in flight under the whole main loop)
    if (tid < BN) ((float*)(smem + cfg::BIAS_OFF))[tid] = p.bias ? p.bias[n0 + tid] : 0.f;
    else if (tid < 2 * BN && p.stats_in) ((float*)(smem + cfg::BIAS_OFF))[tid] = p.colsum[n0 + tid - BN];
    if (p.stats_in) {
        // consumer of a folded LayerNorm: this tile's row statistics (K/256 partial (sum, sum of squares) pairs per row, written by the producer
        // GEMM's column tiles) -> LDS now, so that the epilogue finds them without a global round trip
        const int S3 = p.K >> 8;
        for (int r = tid; r < BM; r += 512) {
            int m = m0 + r;
            if (m > rb_last * 32 + 31) m = rb_last * 32 + 31;
            for (int t = 0; t < S3; ++t)
                *(float2*)(smem + cfg::STAT_OFF + (r * 4 + t) * 8) = *(const float2*)(p.stats_in + ((size_t)m * S3 + t) * 2);
        }
    }

    // ---- DMA units of this wave: u = wave + 8 i -> row block u >> 1 (A blocks first, then the 8 W blocks), 1-KiB half u & 1
    const bool dma_full = (HU % 8 == 0) || (wave < HU % 8);          // this wave issues HUPW units (else HUPW - 1)
    const char* hsrc[HUPW];
#pragma unroll
    for (int i = 0; i < HUPW; ++i) {
        int u = wave + 8 * i;
        if (u >= HU) u = HU - 1;                          // never issued (dma_full is false); keeps the address valid
        const int b = u >> 1, half = u & 1;
        if (b < MB) {
            int rb = (m0 >> 5) + b;
            if (rb > rb_last) rb = rb_last;               // M tail: re-read the last block (its results are not stored)
            hsrc[i] = (const char*)p.A + ((size_t)rb * KC) * 512 + half * 1024 + lane * 16;
        } else {
            hsrc[i] = (const char*)p.W + ((size_t)((n0 >> 5) + b - MB) * KC) * 512 + half * 1024 + lane * 16;
        }
    }
    auto hstage = [&](int h) {
        const int slot = h & 3;
#pragma unroll
        for (int i = 0; i < HUPW; ++i) {
            if (i < HUPW - 1 || dma_full)
                __builtin_amdgcn_global_load_lds((gbl_void_t*)(hsrc[i] + (size_t)h * 2048), (lds_void_t*)(smem + slot * SLOT + (wave + 8 * i) * 1024), 16, 0, 0);
        }
    };
    // own DMA groups still allowed in flight: `young` groups of (HUPW or HUPW - 1) loads
    auto wait_dma = [&](int young) {
        if (young >= 2) { if (dma_full) blk_wait_vmcnt<2 * HUPW>(); else blk_wait_vmcnt<2 * (HUPW - 1)>(); }
        else if (young == 1) { if (dma_full) blk_wait_vmcnt<HUPW>(); else blk_wait_vmcnt<HUPW - 1>(); }
        else blk_wait_vmcnt<0>();
    };

    f32x16_t acc[cfg::MIMAX][NJ];
#pragma unroll
    for (int i = 0; i < cfg::MIMAX; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    hstage(0);
    if (H > 1) hstage(1);
    if (H > 2) hstage(2);
    wait_dma(H > 2 ? 2 : H - 1);
    __builtin_amdgcn_s_barrier();

    // ONE barrier per half K tile; the two groups walk a slot in opposite order:
    //   slot k:   group 0: MFMA(k), MEM(k+1)      group 1: MEM(k+1), MFMA(k+1)
    // so the first half of a slot is MFMA (g0) beside MEM (g1) and the second half the reverse, with no rendezvous in the middle (a slot
    // costs MEM + MFMA, not 2 x max(MEM, MFMA) + a second barrier: qkv 53.9 -> 47.2 us in the lab).  Hazards: MEM(x) of both groups
    // lies in slot x-1: it reads ring slot x & 3 (DMA issued in slot x-4, own share awaited in slot x-2, then a barrier) and refills ring
    // slot (x-1) & 3, last read in slot x-2 by MEM(x-1) -- whose ds_reads are drained (lgkmcnt(0)) before the barrier that ends that slot.
    auto main_loop = [&](auto miw_tag) {
        constexpr int MIW = decltype(miw_tag)::value;
        const uint32_t a_b = lds0 + (wm * MI0) * 2048 + hi * 512 + l31 * 16;
        const uint32_t b_b = lds0 + (MB + wn * 2) * 2048 + hi * 512 + l31 * 16;
        bf16x8_t fa[MIW][2], fb[NJ][2];
        auto MEM = [&](int x) {
            const uint32_t sa = a_b + (x & 3) * SLOT, sb = b_b + (x & 3) * SLOT;
            fb[0][0] = blk_lds_read128<0>(sb); fb[1][0] = blk_lds_read128<2048>(sb);
            fa[0][0] = blk_lds_read128<0>(sa);
            if constexpr (MIW > 1) fa[1][0] = blk_lds_read128<2048>(sa);
            if constexpr (MIW > 2) fa[2][0] = blk_lds_read128<4096>(sa);
            if constexpr (MIW > 3) fa[3][0] = blk_lds_read128<6144>(sa);
            if constexpr (MIW > 4) fa[4][0] = blk_lds_read128<8192>(sa);
            fb[0][1] = blk_lds_read128<1024>(sb); fb[1][1] = blk_lds_read128<2048 + 1024>(sb);
            fa[0][1] = blk_lds_read128<1024>(sa);
            if constexpr (MIW > 1) fa[1][1] = blk_lds_read128<2048 + 1024>(sa);
            if constexpr (MIW > 2) fa[2][1] = blk_lds_read128<4096 + 1024>(sa);
            if constexpr (MIW > 3) fa[3][1] = blk_lds_read128<6144 + 1024>(sa);
            if constexpr (MIW > 4) fa[4][1] = blk_lds_read128<8192 + 1024>(sa);
            if (x + 3 < H) hstage(x + 3);
            wait_dma(H - 2 - x);                                       // own share of half tile x + 1 has landed (x + 2, x + 3 may fly)
            blk_wait_lgkmcnt<0>();
            __builtin_amdgcn_sched_barrier(0);
        };
        auto MFMA = [&]() {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < MIW; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j][kk], fa[i][kk], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        if constexpr (SCHED == 1) {
            MEM(0);
            if (wm == 1) MFMA();                                           // group 1 is half a slot ahead
            for (int k = 0; k < H; ++k) {
                __builtin_amdgcn_s_barrier();
                if (wm == 0) MFMA();                                       // MFMA(k)
                if (k + 1 < H) {
                    MEM(k + 1);
                    if (wm == 1) MFMA();                                   // MFMA(k + 1)
                }
            }
        } else {
            // two barriers per half tile: MEM(h) of one group beside MFMA(h) of the other, rendezvous after each; group 1 one barrier behind
            if (wm == 1) __builtin_amdgcn_s_barrier();
            for (int h = 0; h < H; ++h) {
                MEM(h);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_s_setprio(1);
                MFMA();
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_s_barrier();
            }
            if (wm == 0) __builtin_amdgcn_s_barrier();
        }
    };
    if constexpr (MI0 == MI1) {
        main_loop(std::integral_constant<int, MI0>{});
    } else {
        if (wm == 0) main_loop(std::integral_constant<int, MI0>{});
        else main_loop(std::integral_constant<int, MI1>{});
    }

    // ---- epilogue: straight from the accumulators (lane = row l31 of a 32-row block, 4 consecutive columns per register quad)
    const int miw = wm == 0 ? MI0 : MI1;
    const int rb0 = (m0 >> 5) + (wm == 0 ? 0 : MI0);                   // first row block of this wave
    const int nb0 = n0 + wn * 64;
    const float* sBias = (const float*)(smem + cfg::BIAS_OFF) + wn * 64;
    if constexpr (EPI == 0 || EPI == 1) {
        const int NC8 = p.N >> 3;
        const bool fold = p.stats_in != nullptr;                        // LayerNorm folded into this GEMM: per-row (rstd, rstd * mean)
        float rs[cfg::MIMAX], rm[cfg::MIMAX];
#pragma unroll
        for (int i = 0; i < cfg::MIMAX; ++i) { rs[i] = 1.f; rm[i] = 0.f; }
        if (fold) {
            const int S3 = p.K >> 8;                                     // partial pairs per row (one per 256-column tile of the producer)
            const float invC = 1.0f / (float)p.K;
#pragma unroll
            for (int i = 0; i < cfg::MIMAX; ++i) {
                if (i >= miw) continue;
                const float2* sp = (const float2*)(smem + cfg::STAT_OFF) + ((wm == 0 ? 0 : MI0 * 32) + i * 32 + l31) * 4;
                float sx = 0.f, sxx = 0.f;
                for (int t = 0; t < S3; ++t) { const float2 v = sp[t]; sx += v.x; sxx += v.y; }
                const float mean = sx * invC;
                const float var = fmaxf(fmaf(-mean, mean, sxx * invC), 0.f);
                rs[i] = 1.0f / sqrtf(var + p.ln_eps);
                rm[i] = rs[i] * mean;
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float4 bq[4], cq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bq[q] = *(const float4*)(sBias + j * 32 + 8 * q + 4 * hi);
                cq[q] = fold ? *(const float4*)(sBias + BN + j * 32 + 8 * q + 4 * hi) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int i = 0; i < cfg::MIMAX; ++i) {
                if (i >= miw || rb0 + i > rb_last) continue;
                uint32_t pk[4][2];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // plain: acc + bias;  folded LayerNorm: rstd * acc + (bias' - rstd * mean * colsum)   (rs = 1, rm = 0, cq = 0 when not folded)
                    f32x2_t v0 = {fmaf(acc[i][j][4 * q], rs[i], fmaf(-rm[i], cq[q].x, bq[q].x)), fmaf(acc[i][j][4 * q + 1], rs[i], fmaf(-rm[i], cq[q].y, bq[q].y))};
                    f32x2_t v1 = {fmaf(acc[i][j][4 * q + 2], rs[i], fmaf(-rm[i], cq[q].z, bq[q].z)), fmaf(acc[i][j][4 * q + 3], rs[i], fmaf(-rm[i], cq[q].w, bq[q].w))};
                    if constexpr (EPI == 1) { v0 = gelu_fast2(v0); v1 = gelu_fast2(v1); }
                    pk[q][0] = pack_bf16x2(v0.x, v0.y); pk[q][1] = pack_bf16x2(v1.x, v1.y);
                }
                char* rowp = (char*)p.C + ((size_t)(rb0 + i) * NC8 + ((nb0 + j * 32) >> 3)) * 512 + l31 * 16;
#pragma unroll
                for (int q = 0; q < 4; q += 2) {
                    // half exchange: lanes 0-31 end up with columns 8q..8q+7 (unit q), lanes 32-63 with 8(q+1)..8(q+1)+7 (unit q+1)
                    const auto r0 = __builtin_amdgcn_permlane32_swap(pk[q][0], pk[q + 1][0], false, false);
                    const auto r1 = __builtin_amdgcn_permlane32_swap(pk[q][1], pk[q + 1][1], false, false);
                    *(uint4*)(rowp + (q + hi) * 512) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
                }
            }
        }
    } else {
        const int NC4 = p.N >> 2;
        const bool emit = p.xhat != nullptr;                            // also write bf16(C) as the next GEMM's operand + row partial sums
        float sx[cfg::MIMAX], sxx[cfg::MIMAX];
#pragma unroll
        for (int i = 0; i < cfg::MIMAX; ++i) { sx[i] = 0.f; sxx[i] = 0.f; }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float4 bq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bq[q] = *(const float4*)(sBias + j * 32 + 8 * q + 4 * hi);
#pragma unroll
            for (int i = 0; i < cfg::MIMAX; ++i) {
                if (i >= miw || rb0 + i > rb_last) continue;
                const size_t off = ((size_t)(rb0 + i) * NC4 + ((nb0 + j * 32) >> 2) + hi) * 512 + l31 * 16;
                float4 rv[4];
                if constexpr (EPI == 2) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) rv[q] = *(const float4*)((const char*)p.res + off + q * 1024);
                } else {                                                // EPI 3: row-major residual, row = m % res_rows (pos embed, vit.py:320)
                    const int m = (rb0 + i) * 32 + l31;
                    const float* rr = p.res + (size_t)(m % p.res_rows) * p.N + nb0 + j * 32 + 4 * hi;
#pragma unroll
                    for (int q = 0; q < 4; ++q) rv[q] = *(const float4*)(rr + 8 * q);
                }
                uint32_t pk[4][2];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 o;
                    o.x = acc[i][j][4 * q] + bq[q].x + rv[q].x; o.y = acc[i][j][4 * q + 1] + bq[q].y + rv[q].y;
                    o.z = acc[i][j][4 * q + 2] + bq[q].z + rv[q].z; o.w = acc[i][j][4 * q + 3] + bq[q].w + rv[q].w;
                    *(float4*)((char*)p.C + off + q * 1024) = o;
                    if (emit) {
                        // explicit order / explicit fma: every tile instantiation must produce the same bits for a row (batch-independence tests)
                        sx[i] += o.x; sx[i] += o.y; sx[i] += o.z; sx[i] += o.w;
                        sxx[i] = fmaf(o.x, o.x, sxx[i]); sxx[i] = fmaf(o.y, o.y, sxx[i]); sxx[i] = fmaf(o.z, o.z, sxx[i]); sxx[i] = fmaf(o.w, o.w, sxx[i]);
                        pk[q][0] = pack_bf16x2(o.x, o.y); pk[q][1] = pack_bf16x2(o.z, o.w);
                    }
                }
                if (emit) {
                    char* rowp = (char*)p.xhat + ((size_t)(rb0 + i) * (p.N >> 3) + ((nb0 + j * 32) >> 3)) * 512 + l31 * 16;
#pragma unroll
                    for (int q = 0; q < 4; q += 2) {
                        const auto r0 = __builtin_amdgcn_permlane32_swap(pk[q][0], pk[q + 1][0], false, false);
                        const auto r1 = __builtin_amdgcn_permlane32_swap(pk[q][1], pk[q + 1][1], false, false);
                        *(uint4*)(rowp + (q + hi) * 512) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
                    }
                }
            }
        }
        if (emit) {      // block-uniform
            // per row: (sum x, sum x^2) of the stored fp32 values over this tile's 256 columns = the 4 wave columns combined through LDS in a
            // fixed order (deterministic); one pair per row and column tile goes to stats_out [rows][N/256][2]
            float2* sRed = (float2*)(smem + cfg::STAT_OFF);
#pragma unroll
            for (int i = 0; i < cfg::MIMAX; ++i) {
                if (i >= miw) continue;
                const float a = sx[i] + __shfl_xor(sx[i], 32, 64), b = sxx[i] + __shfl_xor(sxx[i], 32, 64);
                if (hi == 0) sRed[((wm == 0 ? 0 : MI0 * 32) + i * 32 + l31) * 4 + wn] = make_float2(a, b);
            }
            __syncthreads();
            const int S3 = p.N >> 8;
            for (int r = tid; r < BM; r += 512) {
                if ((m0 >> 5) + (r >> 5) > rb_last) continue;
                const float2 v0 = sRed[r * 4], v1 = sRed[r * 4 + 1], v2 = sRed[r * 4 + 2], v3 = sRed[r * 4 + 3];
                *(float2*)(p.stats_out + ((size_t)(m0 + r) * S3 + tn) * 2) = make_float2((v0.x + v1.x) + (v2.x + v3.x), (v0.y + v1.y) + (v2.y + v3.y));
            }
        }
    }
}

static int g_blk_sched = 1;

template <int MI0, int MI1, int EPI, int SCHED>
static int launch_blk_s(const whmr_gemm_blk_desc& p, hipStream_t st) {
    using cfg = blk_cfg<MI0, MI1>;
    auto kern = gemm_blk_kernel<MI0, MI1, EPI, SCHED>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, cfg::LDS);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const int tiles = ((p.M + cfg::BM - 1) / cfg::BM) * (p.N / cfg::BN);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), cfg::LDS, st, p);
    WHMR_CHECK_LAUNCH();
    return 0;
}

template <int MI0, int MI1, int EPI>
static int launch_blk(const whmr_gemm_blk_desc& p, hipStream_t st) {
    return g_blk_sched ? launch_blk_s<MI0, MI1, EPI, 1>(p, st) : launch_blk_s<MI0, MI1, EPI, 0>(p, st);
}

template <int MI0, int MI1>
static int launch_blk_epi(const whmr_gemm_blk_desc& p, hipStream_t st) {
    switch (p.epi) {
        case 0: return launch_blk<MI0, MI1, 0>(p, st);
        case 1: return launch_blk<MI0, MI1, 1>(p, st);
        case 2: return launch_blk<MI0, MI1, 2>(p, st);
        case 3: return launch_blk<MI0, MI1, 3>(p, st);
    }
    return (int)hipErrorInvalidValue;
}

// Tile heights (x 256 columns): the wave rows own MI0 and MI1 row blocks.  The chooser minimises (rounds over 256 CUs) x (tile rows).
static const int kBlkTiles[][2] = {{4, 4}, {5, 5}, {4, 3}, {3, 3}, {3, 2}, {2, 2}, {5, 4}, {2, 1}};

extern "C" int whmr_gemm_blk_tile(const whmr_gemm_blk_desc* pp, int tile, void* stream) {
    const whmr_gemm_blk_desc& p = *pp;
    if (p.M <= 0 || p.N <= 0 || (p.N % 256) || p.K < 32 || (p.K % 32) || p.epi < 0 || p.epi > 3) return (int)hipErrorInvalidValue;
    if ((p.epi >= 2) && !p.res) return (int)hipErrorInvalidValue;
    if (p.xhat && (p.epi < 2 || !p.stats_out)) return (int)hipErrorInvalidValue;
    if (p.stats_in && (p.epi >= 2 || !p.colsum || (p.K % 256) || p.K > 1024)) return (int)hipErrorInvalidValue;
    if (p.xhat && p.N > 1024) return (int)hipErrorInvalidValue;
    if (p.epi == 3 && p.res_rows <= 0) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    switch (tile) {
        case 0x44: return launch_blk_epi<4, 4>(p, st);      // 256 x 256
        case 0x55: return launch_blk_epi<5, 5>(p, st);      // 320 x 256
        case 0x43: return launch_blk_epi<4, 3>(p, st);      // 224 x 256
        case 0x33: return launch_blk_epi<3, 3>(p, st);      // 192 x 256
        case 0x32: return launch_blk_epi<3, 2>(p, st);      // 160 x 256
        case 0x22: return launch_blk_epi<2, 2>(p, st);      // 128 x 256
        case 0x54: return launch_blk_epi<5, 4>(p, st);      // 288 x 256
        case 0x21: return launch_blk_epi<2, 1>(p, st);      // 96 x 256: ViT-L at 32 crops (6144 tokens) x N = 1024 is exactly 256 such tiles
    }
    return (int)hipErrorInvalidValue;
}

static int g_blk_force[4] = {0, 0, 0, 0};                   // whmr_set_option keys 110..113: tile for N = 2304 / (768, K <= 1024) / 3072 / (768, K > 1024)
extern "C" int whmr_gemm_blk_set_tile(int slot, int tile) {
    if (slot == 4) { g_blk_sched = tile; return 0; }                 // A/B: main-loop schedule (0 two barriers per half tile, 1 one barrier)
    if (slot < 0 || slot > 3) return (int)hipErrorInvalidValue;
    g_blk_force[slot] = tile;
    return 0;
}

extern "C" int whmr_gemm_blk(const whmr_gemm_blk_desc* pp, void* stream) {
    const whmr_gemm_blk_desc& p = *pp;
    if (p.tile) return whmr_gemm_blk_tile(pp, p.tile, stream);
    const int slot = p.N == 2304 ? 0 : p.N == 3072 ? 2 : p.N == 768 ? (p.K <= 1024 ? 1 : 3) : -1;
    if (slot >= 0 && g_blk_force[slot]) return whmr_gemm_blk_tile(pp, g_blk_force[slot], stream);
    // Cost model (in-model A/B on three boxes, tools/blk_ab.py; ViT-B 224^2 batch 64): with ONE round of tiles (<= 256) a launch lasts one
    // tile, so the lowest tile wins (N = 768: 160 rows / 237 tiles beats 192 / 198 by 4 us per launch).  With two or more rounds every CU
    // is busy all the time, the chip sits at its power limit and what counts is total CU time incl. the per-tile prologue / epilogue
    // (~64 rows' worth): qkv on 441 tiles of 256 rows beats 504 tiles of 224 by 9 us per launch although the latter fills two rounds.
    long best_cost = -1;
    int best = 0x44;
    const int tiles_n = p.N / 256;
    for (const auto& t : kBlkTiles) {
        const int bm = 32 * (t[0] + t[1]);
        const long tiles = (long)((p.M + bm - 1) / bm) * tiles_n;
        const long rounds = (tiles + 255) / 256;
        long cost = rounds * (bm + 64) * 256;
        if (rounds >= 2) { const long thr = tiles * (bm + 64) * 5 / 4; if (thr > cost) cost = thr; }
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = (t[0] << 4) | t[1]; }
    }
    return whmr_gemm_blk_tile(pp, best, stream);
}
