// Kernel template + launchers of the blocked-layout MFMA GEMM, shared by gemm_blk.hip (bf16 operands) and gemm_blk_x3.hip (split-bf16
// operands: X3 = true).  See gemm_blk.hip for the layout and the schedule.
//
// X3 ("bf16x3", the parity-grade numerics of the ViT inference path): every operand is a PAIR of blocked bf16 matrices, x = x_hi + x_lo with
// x_hi = bf16(x), x_lo = bf16(x - x_hi) (16 significand bits), and a product is three MFMAs per fragment pair,
//     a.w ~= a_hi.w_hi + a_lo.w_hi + a_hi.w_lo          (the dropped a_lo.w_lo term is 2^-16 of the product),
// accumulated in fp32 -- fp32-grade results at a third of the bf16 MFMA rate, several times the exact-f32 MFMA path.  The ring slot keeps
// its size: a slot is a 16-deep K slice with the hi and the lo 1-KiB unit of a row block side by side where the bf16 kernel keeps two
// consecutive 16-deep units, so the fragment reads are IDENTICAL (fa[i][0] = hi, fa[i][1] = lo) and only the DMA source addresses and the
// MFMA phase differ: 24 MFMAs (MI = 4) against the same 12 ds_reads + DMA share per phase.  In the bf16 kernel a wave's MEM phase (12 fragment
// reads + its 4 LDS-DMA pieces, each ~100-185 cycles to issue inside a busy phase: MI355X_MICROARCH.md, + the counted waits) outlasts its partner's
// 512-cycle MFMA phase; here the MFMA phase is 768 cycles and covers it, so the loop is MFMA-bound.
#pragma once
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

// WD ("W direct"): the weight fragments never touch LDS -- each wave loads its own 64 columns of W straight from global memory / L2 into
// registers (a blocked 1-KiB unit IS an MFMA operand: one coalesced global_load_dwordx4 per fragment), three half tiles deep; the ring holds only A
// and a wave issues 2 LDS-DMA pieces + 4 register loads per phase instead of 4 pieces, and reads 8 fragments from LDS instead of 12.  Measured
// 3-4 % SLOWER than the LDS-staged loop: the texture path carries the same bytes either way, and two half tiles of prefetch distance do
// not cover a global load's latency under load (a fourth register set does not fit).  Kept as schedule 2 for A/B runs; bit-identical results.
template <int MI0, int MI1, bool WD = false>
struct blk_cfg {
    static constexpr int MB = MI0 + MI1;                 // A row blocks (32 rows) per tile
    static constexpr int BM = MB * 32, BN = 256;
    static constexpr int NBL = WD ? 0 : 8;               // W row blocks staged through LDS
    static constexpr int SLOT = (MB + NBL) * 2048;       // one half K tile (32 deep) of A (and W): 2 KiB per row block
    static constexpr int HU = (MB + NBL) * 2;            // 1-KiB DMA units per half tile
    static constexpr int HUPW = (HU + 7) / 8;            // units per wave (waves >= HU % 8 issue one less when HU % 8 != 0)
    static constexpr int BIAS_OFF = 4 * SLOT;            // [256] floats behind the ring, then [256] floats of the LayerNorm-fold column sums
    static constexpr int STAT_OFF = BIAS_OFF + 2048;     // LayerNorm folding: [BM][4][2] floats -- row statistics (consumer) / per-wave-column partial sums (producer)
    static constexpr int LDS = 4 * SLOT + 2048 + BM * 32;
    static constexpr int MIMAX = MI0 > MI1 ? MI0 : MI1;
};

// SCHED 1: one barrier per half tile, groups in opposite order within a slot;  SCHED 0: two barriers per half tile (MEM | MFMA rendezvous)
template <int MI0, int MI1, int EPI, int SCHED, bool X3 = false, bool WD = false>
__global__ __launch_bounds__(512, 2) void gemm_blk_kernel(const whmr_gemm_blk_desc p) {
    using cfg = blk_cfg<MI0, MI1, WD>;
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
    const int H = X3 ? (p.K >> 4) : (p.K >> 5);           // ring slots to walk: half K tiles (32 deep); X3: 16-deep slices, hi | lo units side by side
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
            hsrc[i] = (const char*)((X3 && half) ? p.A_lo : p.A) + ((size_t)rb * KC) * 512 + (X3 ? 0 : half * 1024) + lane * 16;
        } else {
            hsrc[i] = (const char*)((X3 && half) ? p.W_lo : p.W) + ((size_t)((n0 >> 5) + b - MB) * KC) * 512 + (X3 ? 0 : half * 1024) + lane * 16;
        }
    }
    auto hstage = [&](int h) {
        const int slot = h & 3;
#pragma unroll
        for (int i = 0; i < HUPW; ++i) {
            if (i < HUPW - 1 || dma_full)
                __builtin_amdgcn_global_load_lds((gbl_void_t*)(hsrc[i] + (size_t)h * (X3 ? 1024 : 2048)), (lds_void_t*)(smem + slot * SLOT + (wave + 8 * i) * 1024), 16, 0, 0);
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

    if constexpr (!WD) {
        hstage(0);
        if (H > 1) hstage(1);
        if (H > 2) hstage(2);
        wait_dma(H > 2 ? 2 : H - 1);
        __builtin_amdgcn_s_barrier();
    }

    // ONE barrier per half K tile; the two groups walk a slot in opposite order:
    //   slot k:   group 0: MFMA(k), MEM(k+1)      group 1: MEM(k+1), MFMA(k+1)
    // so the first half of a slot is MFMA (g0) beside MEM (g1) and the second half the reverse, with no rendezvous in the middle (a slot
    // costs MEM + MFMA, not 2 x max(MEM, MFMA) + a second barrier: qkv 53.9 -> 47.2 us in the lab).  Hazards: MEM(x) of both groups
    // lies in slot x-1: it reads ring slot x & 3 (DMA issued in slot x-4, own share awaited in slot x-2, then a barrier) and refills ring
    // slot (x-1) & 3, last read in slot x-2 by MEM(x-1) -- whose ds_reads are drained (lgkmcnt(0)) before the barrier that ends that slot.
    // W direct: the wave's 64 weight columns, as an SGPR base + the lane's 16-B offset
    const uint32_t lane16 = lane * 16;
    const uint64_t wbase = (uint64_t)(uintptr_t)p.W + ((uint64_t)((n0 >> 5) + wn * NJ) * KC) * 512;
    const uint64_t wbase_lo = X3 ? (uint64_t)(uintptr_t)p.W_lo + ((uint64_t)((n0 >> 5) + wn * NJ) * KC) * 512 : wbase;
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
            if constexpr (X3) {
                // [.][0] = hi, [.][1] = lo of the same 16-deep slice: w_hi.a_lo, w_lo.a_hi (the small terms first), then w_hi.a_hi
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int i = 0; i < MIW; ++i)
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j][t == 1 ? 1 : 0], fa[i][t == 0 ? 1 : 0], acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int i = 0; i < MIW; ++i)
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j][kk], fa[i][kk], acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        if constexpr (WD) {
            // ---- W direct: fbq[set] = the wave's W fragments of half tile h (set = h % 3), requested two half tiles ahead of their MFMA phase.
            // vmcnt bookkeeping: a wave's loads in issue order are ... B(x+1) DMA(x+2) | B(x+2) DMA(x+3) at the end of MEM(x); everything older
            // (DMA(x+1): its share of the next LDS slot, B(x): the operands of MFMA(x)) must have landed.
            static_assert(SCHED == 1, "W direct runs the one-barrier schedule");
            constexpr int NB4 = 2 * NJ;                                    // B loads per half tile and wave
            bf16x8_t fbq[3][NJ][2];
            // The loads are inline asm (SGPR base + the lane's 16-B offset), invisible to hipcc's waitcnt pass like the fragment ds_reads above: seen
            // as ordinary loads it drains vmcnt to ZERO in front of every MFMA phase and in front of every reload of a set (the loads of a set
            // cross the loop's back edge), i.e. it waits for the DMA three half tiles ahead.  The counted waits of MEMW cover them.
            auto gload = [](bf16x8_t& dst, uint64_t base, uint32_t voff, auto off_tag) {
                constexpr int OFF = decltype(off_tag)::value;
                const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base), hi32 = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
                const uint64_t sb = ((uint64_t)hi32 << 32) | lo;
                asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sb), "n"(OFF));
            };
            auto loadB = [&](auto set_tag, int h) {
                constexpr int SET = decltype(set_tag)::value;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if constexpr (X3) {
                        gload(fbq[SET][j][0], wbase + (uint64_t)j * KC * 512 + (uint64_t)h * 1024, lane16, std::integral_constant<int, 0>{});
                        gload(fbq[SET][j][1], wbase_lo + (uint64_t)j * KC * 512 + (uint64_t)h * 1024, lane16, std::integral_constant<int, 0>{});
                    } else {
                        const uint64_t b = wbase + (uint64_t)j * KC * 512 + (uint64_t)h * 2048;
                        gload(fbq[SET][j][0], b, lane16, std::integral_constant<int, 0>{});
                        gload(fbq[SET][j][1], b, lane16, std::integral_constant<int, 1024>{});
                    }
                }
            };
            auto wait_young = [&](int nd, int nb) {                         // leave nd DMA groups + nb B groups (the youngest loads) in flight
                constexpr int D1 = HUPW, D0 = HUPW - 1;
                if (nb >= 2) {
                    if (nd >= 2) { if (dma_full) blk_wait_vmcnt<2 * D1 + 2 * NB4>(); else blk_wait_vmcnt<2 * D0 + 2 * NB4>(); }
                    else if (nd == 1) { if (dma_full) blk_wait_vmcnt<D1 + 2 * NB4>(); else blk_wait_vmcnt<D0 + 2 * NB4>(); }
                    else blk_wait_vmcnt<2 * NB4>();
                } else if (nb == 1) {
                    if (nd >= 1) { if (dma_full) blk_wait_vmcnt<D1 + NB4>(); else blk_wait_vmcnt<D0 + NB4>(); }
                    else blk_wait_vmcnt<NB4>();
                } else blk_wait_vmcnt<0>();
            };
            auto wait_wd = [&](int x) {                                    // end of MEM(x): allow the loads issued in MEM(x-1) and MEM(x) only
                const int nd = H - 2 - x, nb = H - 1 - x;                   // younger DMA groups (x+2, x+3) / B groups (x+1, x+2) that exist
                wait_young(nd < 0 ? 0 : nd > 2 ? 2 : nd, nb < 0 ? 0 : nb > 2 ? 2 : nb);
            };
            auto MEMW = [&](auto set_tag, int x) {                          // MEM(x) with set = x % 3: A fragments from LDS, B(x+2) -> set (x+2) % 3
                constexpr int SET = decltype(set_tag)::value;
                const uint32_t sa = a_b + (x & 3) * SLOT;
                fa[0][0] = blk_lds_read128<0>(sa);
                if constexpr (MIW > 1) fa[1][0] = blk_lds_read128<2048>(sa);
                if constexpr (MIW > 2) fa[2][0] = blk_lds_read128<4096>(sa);
                if constexpr (MIW > 3) fa[3][0] = blk_lds_read128<6144>(sa);
                if constexpr (MIW > 4) fa[4][0] = blk_lds_read128<8192>(sa);
                fa[0][1] = blk_lds_read128<1024>(sa);
                if constexpr (MIW > 1) fa[1][1] = blk_lds_read128<2048 + 1024>(sa);
                if constexpr (MIW > 2) fa[2][1] = blk_lds_read128<4096 + 1024>(sa);
                if constexpr (MIW > 3) fa[3][1] = blk_lds_read128<6144 + 1024>(sa);
                if constexpr (MIW > 4) fa[4][1] = blk_lds_read128<8192 + 1024>(sa);
                if (x + 2 < H) loadB(std::integral_constant<int, (SET + 2) % 3>{}, x + 2);
                if (x + 3 < H) hstage(x + 3);
                wait_wd(x);
                blk_wait_lgkmcnt<0>();
                __builtin_amdgcn_sched_barrier(0);
            };
            auto MFMAW = [&](auto set_tag) {
                constexpr int SET = decltype(set_tag)::value;
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (X3) {
#pragma unroll
                    for (int t = 0; t < 3; ++t)
#pragma unroll
                        for (int i = 0; i < MIW; ++i)
#pragma unroll
                            for (int j = 0; j < NJ; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fbq[SET][j][t == 1 ? 1 : 0], fa[i][t == 0 ? 1 : 0], acc[i][j], 0, 0, 0);
                } else {
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                        for (int i = 0; i < MIW; ++i)
#pragma unroll
                            for (int j = 0; j < NJ; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fbq[SET][j][kk], fa[i][kk], acc[i][j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            auto body = [&](auto set_tag, int k) {                          // slot k: group 0: MFMA(k), MEM(k+1);  group 1: MEM(k+1), MFMA(k+1)
                constexpr int SET = decltype(set_tag)::value;
                using next = std::integral_constant<int, (SET + 1) % 3>;
                __builtin_amdgcn_s_barrier();
                if (wm == 0) MFMAW(set_tag);
                if (k + 1 < H) {
                    MEMW(next{}, k + 1);
                    if (wm == 1) MFMAW(next{});
                }
            };
            // prologue in the steady-state issue order: DMA(0) | B(0) DMA(1) | B(1) DMA(2); then DMA(0) must have landed (barrier: everyone's share)
            hstage(0);
            loadB(std::integral_constant<int, 0>{}, 0);
            if (H > 1) { hstage(1); loadB(std::integral_constant<int, 1>{}, 1); }
            if (H > 2) hstage(2);
            wait_young(H > 2 ? 2 : H - 1, H > 1 ? 2 : 1);
            __builtin_amdgcn_s_barrier();
            MEMW(std::integral_constant<int, 0>{}, 0);
            if (wm == 1) MFMAW(std::integral_constant<int, 0>{});
            for (int k = 0; k < H; k += 3) {
                body(std::integral_constant<int, 0>{}, k);
                if (k + 1 < H) body(std::integral_constant<int, 1>{}, k + 1);
                if (k + 2 < H) body(std::integral_constant<int, 2>{}, k + 2);
            }
        } else if constexpr (SCHED == 1) {
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
                uint32_t pk[4][2], pl[4][2];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // plain: acc + bias;  folded LayerNorm: rstd * acc + (bias' - rstd * mean * colsum)   (rs = 1, rm = 0, cq = 0 when not folded)
                    f32x2_t v0 = {fmaf(acc[i][j][4 * q], rs[i], fmaf(-rm[i], cq[q].x, bq[q].x)), fmaf(acc[i][j][4 * q + 1], rs[i], fmaf(-rm[i], cq[q].y, bq[q].y))};
                    f32x2_t v1 = {fmaf(acc[i][j][4 * q + 2], rs[i], fmaf(-rm[i], cq[q].z, bq[q].z)), fmaf(acc[i][j][4 * q + 3], rs[i], fmaf(-rm[i], cq[q].w, bq[q].w))};
                    if constexpr (X3) {
                        // parity-grade epilogue: erf GELU like nn.GELU (vit.py:66-68; gelu_as: 8.7e-7 from float64, the fp32 framework GELU 1.2e-6), result split into hi / lo
                        if constexpr (EPI == 1) { v0.x = gelu_as(v0.x); v0.y = gelu_as(v0.y); v1.x = gelu_as(v1.x); v1.y = gelu_as(v1.y); }
                        split_bf16x2(v0.x, v0.y, pk[q][0], pl[q][0]); split_bf16x2(v1.x, v1.y, pk[q][1], pl[q][1]);
                    } else {
                        if constexpr (EPI == 1) { v0 = gelu_fast2(v0); v1 = gelu_fast2(v1); }
                        pk[q][0] = pack_bf16x2(v0.x, v0.y); pk[q][1] = pack_bf16x2(v1.x, v1.y);
                    }
                }
                const size_t roff = ((size_t)(rb0 + i) * NC8 + ((nb0 + j * 32) >> 3)) * 512 + l31 * 16;
                char* rowp = (char*)p.C + roff;
#pragma unroll
                for (int q = 0; q < 4; q += 2) {
                    // half exchange: lanes 0-31 end up with columns 8q..8q+7 (unit q), lanes 32-63 with 8(q+1)..8(q+1)+7 (unit q+1)
                    const auto r0 = __builtin_amdgcn_permlane32_swap(pk[q][0], pk[q + 1][0], false, false);
                    const auto r1 = __builtin_amdgcn_permlane32_swap(pk[q][1], pk[q + 1][1], false, false);
                    *(uint4*)(rowp + (q + hi) * 512) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
                }
                if constexpr (X3) {
                    char* rowl = (char*)p.C_lo + roff;
#pragma unroll
                    for (int q = 0; q < 4; q += 2) {
                        const auto r0 = __builtin_amdgcn_permlane32_swap(pl[q][0], pl[q + 1][0], false, false);
                        const auto r1 = __builtin_amdgcn_permlane32_swap(pl[q][1], pl[q + 1][1], false, false);
                        *(uint4*)(rowl + (q + hi) * 512) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
                    }
                }
            }
        }
    } else {
        const int NC4 = p.N >> 2;
        const bool emit = p.xhat != nullptr;                            // also write bf16(C) as the next GEMM's operand + row partial sums
        float sx[cfg::MIMAX], sxx[cfg::MIMAX], sh[cfg::MIMAX];
#pragma unroll
        for (int i = 0; i < cfg::MIMAX; ++i) { sx[i] = 0.f; sxx[i] = 0.f; sh[i] = 0.f; }
        if (emit && (p.shift || p.shift_stats || p.shift_out)) {
            // Per-row SHIFT of the folded LayerNorm: the bf16 operand copy and the partial sums are taken of (x - s_m), s_m = the row's mean one
            // residual step earlier (its previous shift + the mean of its previous shifted statistics).  LN(x) = ((x - s) - mean(x - s)) * rstd is
            // exact for ANY s, so the consumer's formula does not change -- but the value that gets rounded to bf16 is now centred, so the
            // rounding error is relative to the row's spread instead of its offset (a trained ViT's token offsets / massive channels).
            const int S3 = p.N >> 8;
            const float invN = 1.0f / (float)p.N;
#pragma unroll
            for (int i = 0; i < cfg::MIMAX; ++i) {
                if (i >= miw || rb0 + i > rb_last) continue;
                const int m = (rb0 + i) * 32 + l31;
                float s = p.shift ? p.shift[m] : 0.f;
                if (p.shift_stats) {
                    float t = 0.f;
                    for (int u = 0; u < S3; ++u) t += p.shift_stats[((size_t)m * S3 + u) * 2];
                    s = fmaf(t, invN, s);
                }
                sh[i] = s;
                if (p.shift_out && tn == 0 && wn == 0 && hi == 0) p.shift_out[m] = s;
            }
        }
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
                uint32_t pk[4][2], pl[4][2];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 o;
                    o.x = acc[i][j][4 * q] + bq[q].x + rv[q].x; o.y = acc[i][j][4 * q + 1] + bq[q].y + rv[q].y;
                    o.z = acc[i][j][4 * q + 2] + bq[q].z + rv[q].z; o.w = acc[i][j][4 * q + 3] + bq[q].w + rv[q].w;
                    *(float4*)((char*)p.C + off + q * 1024) = o;
                    if (emit) {
                        // explicit order / explicit fma: every tile instantiation must produce the same bits for a row (batch-independence tests)
                        o.x -= sh[i]; o.y -= sh[i]; o.z -= sh[i]; o.w -= sh[i];
                        sx[i] += o.x; sx[i] += o.y; sx[i] += o.z; sx[i] += o.w;
                        sxx[i] = fmaf(o.x, o.x, sxx[i]); sxx[i] = fmaf(o.y, o.y, sxx[i]); sxx[i] = fmaf(o.z, o.z, sxx[i]); sxx[i] = fmaf(o.w, o.w, sxx[i]);
                        if constexpr (X3) { split_bf16x2(o.x, o.y, pk[q][0], pl[q][0]); split_bf16x2(o.z, o.w, pk[q][1], pl[q][1]); }
                        else { pk[q][0] = pack_bf16x2(o.x, o.y); pk[q][1] = pack_bf16x2(o.z, o.w); }
                    }
                }
                if (emit) {
                    const size_t xoff = ((size_t)(rb0 + i) * (p.N >> 3) + ((nb0 + j * 32) >> 3)) * 512 + l31 * 16;
                    char* rowp = (char*)p.xhat + xoff;
#pragma unroll
                    for (int q = 0; q < 4; q += 2) {
                        const auto r0 = __builtin_amdgcn_permlane32_swap(pk[q][0], pk[q + 1][0], false, false);
                        const auto r1 = __builtin_amdgcn_permlane32_swap(pk[q][1], pk[q + 1][1], false, false);
                        *(uint4*)(rowp + (q + hi) * 512) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
                    }
                    if constexpr (X3) {                                     // the centred row as an operand PAIR (bf16x3 fold)
                        char* rowl = (char*)p.xhat_lo + xoff;
#pragma unroll
                        for (int q = 0; q < 4; q += 2) {
                            const auto r0 = __builtin_amdgcn_permlane32_swap(pl[q][0], pl[q + 1][0], false, false);
                            const auto r1 = __builtin_amdgcn_permlane32_swap(pl[q][1], pl[q + 1][1], false, false);
                            *(uint4*)(rowl + (q + hi) * 512) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
                        }
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

template <int MI0, int MI1, int EPI, int SCHED, bool X3, bool WD = false>
static int launch_blk_s(const whmr_gemm_blk_desc& p, hipStream_t st) {
    using cfg = blk_cfg<MI0, MI1, WD>;
    auto kern = gemm_blk_kernel<MI0, MI1, EPI, SCHED, X3, WD>;
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

template <int MI0, int MI1, int EPI, bool X3>
static int launch_blk(const whmr_gemm_blk_desc& p, hipStream_t st, int sched) {
    // Only the split-bf16 form (X3) with W through LDS and the one-barrier schedule is compiled since round 5.  The body still carries its plain-bf16
    // branch (the bf16 kernel moved to 16x16x32: gemm_blk16_impl.h), the two-barrier schedule and the W-direct main loop (measured 3-4 % slower,
    // DESIGN 6) behind `if constexpr`; none of them is instantiated.
    static_assert(X3, "the 32x32x16 blocked kernel is the split-bf16 kernel");
    (void)sched;
    return launch_blk_s<MI0, MI1, EPI, 1, true>(p, st);
}

template <int MI0, int MI1, bool X3>
static int launch_blk_epi(const whmr_gemm_blk_desc& p, hipStream_t st, int sched) {
    switch (p.epi) {
        case 0: return launch_blk<MI0, MI1, 0, X3>(p, st, sched);
        case 1: return launch_blk<MI0, MI1, 1, X3>(p, st, sched);
        case 2: return launch_blk<MI0, MI1, 2, X3>(p, st, sched);
        case 3: return launch_blk<MI0, MI1, 3, X3>(p, st, sched);
    }
    return (int)hipErrorInvalidValue;
}

// Tile heights (x 256 columns): the wave rows own MI0 and MI1 row blocks.
template <bool X3>
static int blk_launch_tile(const whmr_gemm_blk_desc& p, int tile, hipStream_t st, int sched) {
    switch (tile) {
        case 0x44: return launch_blk_epi<4, 4, X3>(p, st, sched);      // 256 x 256
        case 0x55: return launch_blk_epi<5, 5, X3>(p, st, sched);      // 320 x 256
        case 0x43: return launch_blk_epi<4, 3, X3>(p, st, sched);      // 224 x 256
        case 0x33: return launch_blk_epi<3, 3, X3>(p, st, sched);      // 192 x 256
        case 0x32: return launch_blk_epi<3, 2, X3>(p, st, sched);      // 160 x 256
        case 0x22: return launch_blk_epi<2, 2, X3>(p, st, sched);      // 128 x 256
        case 0x54: return launch_blk_epi<5, 4, X3>(p, st, sched);      // 288 x 256
        case 0x21: return launch_blk_epi<2, 1, X3>(p, st, sched);      // 96 x 256: ViT-L at 32 crops (6144 tokens) x N = 1024 is exactly 256 such tiles
    }
    return (int)hipErrorInvalidValue;
}

// split-operand launcher (gemm_blk_x3.hip; library-internal), reached through whmr_gemm_blk / whmr_gemm_blk_tile when the descriptor carries lo halves
__attribute__((visibility("hidden"))) int blk_x3_launch_tile(const whmr_gemm_blk_desc* pp, int tile, void* stream, int sched);
