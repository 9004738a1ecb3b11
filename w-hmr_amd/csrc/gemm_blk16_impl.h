// bf16 kernel of the blocked-layout MFMA GEMM on v_mfma_f32_16x16x32_bf16 (gemm_blk.hip has the layout and the schedule; the split-bf16
// variant keeps the 32x32x16 kernel of gemm_blk_impl.h).
//
// Why this instruction: the launches that keep all 256 CUs busy run at the chip's POWER limit (zero-filled operands are 13-22 % faster,
// DESIGN 6), and a loop of nothing but MFMAs sustains 1.89 PF with 16x16x32 against 1.73-1.75 PF with 32x32x16 on random data
// (tools/lab/mfma_power.hip) -- the 16x16 form moves fewer operand bits per MAC through the register file.  Same fragments per half tile (a
// blocked 32-row unit is read as two 16-row halves x four 8-deep K chunks instead of 32 rows x two chunks), same accumulator count, same
// ring, same schedule; measured on the ViT-B 224^2 batch-64 forward, interleaved on one box: 3.085 -> 2.973 ms (tools/lab/mfma16_ab.sh; qkv
// 50.9 -> 48.7 us, fc1 + GELU 70.5 -> 66.6, proj / fc2 50.8 -> 49.1).
//
// Fragment maps (cdna_hip_programming.md "Fragment layout"): lane l = (g = l >> 4, c = l & 15) holds K chunk g (8 consecutive k) of row c of
// each operand; D[row = 4 g + r][col = c] for r < 4, row = index in the FIRST operand.  The W fragment is the first operand and the A
// fragment the second, so a lane owns ONE row m of A's 16-row half and 4 columns n per 16x16 tile.  The 32 columns of a W block are placed
// in the LDS unit so that the two tiles nh = 0, 1 of a block give lane group g the columns 8 g + 4 nh + r -- together the 8 consecutive
// columns of blocked unit g: the epilogue stores 16 B per lane with no cross-lane traffic at all (the 32x32 kernel needed a permlane32
// swap per register pair).  LDS slot s = 16 nh + i of a unit holds column 8 (i >> 2) + 4 nh + (i & 3); the permutation is applied on the
// DMA's SOURCE side (a lane of the LDS-DMA picks its own 16 bytes inside the same contiguous 512-B unit), so the global access pattern and
// the conflict-free 256-B-per-16-lanes fragment reads are those of the plain layout.
#pragma once
#include <type_traits>
#include "common.h"
#include "gemm_blk.h"

// ---- lab instrumentation (tools/gemm_stamps.py; NEVER defined in the product build): per tile and wave group, s_memrealtime (100 MHz, one clock for
// the whole chip) at kernel entry / first half tile landed / main loop done / epilogue stores issued / stores drained, the CU it ran on and the
// shader-clock count of the tile.  The kernels gain a trailing pointer parameter; the C ABI is unchanged (whmr_debug_blk_stamps sets the buffer).
#ifdef WHMR_BLK_STAMPS
#define BLK16_STAMP_PARAM , unsigned long long* stamps
#define BLK16_STAMP_PASS , stamps
#define BLK16_STAMP(slot)                                                                                                    \
    do {                                                                                                                     \
        if (stamps && lane == 0 && (wave & 3) == 0)                                                                          \
            stamps[((size_t)lid * 2 + (wave >> 2)) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime();                         \
    } while (0)
#define BLK16_STAMP_RAW(slot, v)                                                                                             \
    do {                                                                                                                     \
        if (stamps && lane == 0 && (wave & 3) == 0) stamps[((size_t)lid * 2 + (wave >> 2)) * 8 + (slot)] = (v);              \
    } while (0)
unsigned long long* blk_stamp_next(int tiles);
#define BLK16_STAMP_ARG(tiles) , blk_stamp_next(tiles)
#else
#define BLK16_STAMP_PARAM
#define BLK16_STAMP_PASS
#define BLK16_STAMP(slot) do { } while (0)
#define BLK16_STAMP_RAW(slot, v) do { } while (0)
#define BLK16_STAMP_ARG(tiles)
#endif

typedef __attribute__((address_space(3))) void lds16_void_t;
typedef const __attribute__((address_space(1))) void gbl16_void_t;

template <int N> __device__ __forceinline__ void b16_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void b16_wait_lgkmcnt() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
// fragment read hidden from hipcc's waitcnt bookkeeping (valid after the counted wait + sched_barrier that follows it)
template <int OFF> __device__ __forceinline__ bf16x8_t b16_lds_read128(uint32_t addr) {
    bf16x8_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

template <int MI0, int MI1>
struct blk16_cfg {
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

// One barrier per half tile, the two wave groups in opposite order within a slot.  (Round 3-4 also carried a two-barrier schedule and the same tile on
// FOUR waves, one per SIMD -- tile id 0x144, bit-identical, same K slope, worse epilogue: profiles/r04_fw4_ab.txt.  Both are gone from the build.)
// lid_in >= 0: the logical tile id is given by the caller (the persistent chain kernel below walks a tile list); < 0: one tile per workgroup, XCD-aware order
template <int MI0, int MI1, int EPI, int SCHED, int NW>
__device__ __forceinline__ void gemm_blk16_body(const whmr_gemm_blk_desc& p, char* smem, int lid_in BLK16_STAMP_PARAM) {
    using cfg = blk16_cfg<MI0, MI1>;
    constexpr int MB = cfg::MB, BM = cfg::BM, BN = cfg::BN, SLOT = cfg::SLOT, HU = cfg::HU, NJ = NW == 8 ? 2 : 4, NT = NW * 64;
    constexpr int HUPW = (HU + NW - 1) / NW;             // DMA units per wave (waves >= HU % NW issue one less when HU % NW != 0)
    static_assert(NW == 8 && SCHED == 1, "eight waves, one barrier per half tile");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = NW == 8 ? wave >> 2 : wave >> 1, wn = NW == 8 ? wave & 3 : wave & 1;     // NW 8: wm = group
    const int l15 = lane & 15, g = lane >> 4;             // row inside a 16-row half / K chunk of the operand fragments = column group of the results
    const int tiles_n = p.N / BN, tiles_m = (p.M + BM - 1) / BM;
    const int lid = lid_in >= 0 ? lid_in : xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = lid / tiles_n, tn = lid % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    BLK16_STAMP(0);
    BLK16_STAMP_RAW(5, ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492));    // XCC_ID | HW_ID
    BLK16_STAMP_RAW(6, __builtin_amdgcn_s_memtime());
    const int KC = p.K >> 3;                              // 16-B chunks per row
    const int H = p.K >> 5;                               // half K tiles (32 deep) = ring slots to walk
    const int rb_last = ((p.M + 31) >> 5) - 1;            // last valid row block
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds16_void_t*)smem;

    // this tile's bias slice -> LDS (one float per thread, in flight under the whole main loop)
    for (int t = tid; t < 2 * BN; t += NT) {
        if (t < BN) ((float*)(smem + cfg::BIAS_OFF))[t] = p.bias ? p.bias[n0 + t] : 0.f;
        else if (p.stats_in) ((float*)(smem + cfg::BIAS_OFF))[t] = p.colsum[n0 + t - BN];
    }
    if (p.stats_in) {
        // consumer of a folded LayerNorm: this tile's row statistics (K/256 partial (sum, sum of squares) pairs per row, written by the producer
        // GEMM's column tiles) -> LDS now, so that the epilogue finds them without a global round trip
        const int S3 = p.K >> 8;
        for (int r = tid; r < BM; r += NT) {
            int m = m0 + r;
            if (m > rb_last * 32 + 31) m = rb_last * 32 + 31;
            for (int t = 0; t < S3; ++t)
                *(float2*)(smem + cfg::STAT_OFF + (r * 4 + t) * 8) = *(const float2*)(p.stats_in + ((size_t)m * S3 + t) * 2);
        }
    }

    // ---- DMA units of this wave: u = wave + 8 i -> row block u >> 1 (A blocks first, then the 8 W blocks), 1-KiB half u & 1 (two 8-deep chunks).
    // A units are copied as they are; inside a W chunk LDS slot s (= lane & 31) receives column 8 ((s & 15) >> 2) + 4 (s >> 4) + (s & 3).
    const bool dma_full = (HU % NW == 0) || (wave < HU % NW);        // this wave issues HUPW units (else HUPW - 1)
    const int wslot = lane & 31;
    const int wsrc = (lane & 32) * 16 + (8 * ((wslot & 15) >> 2) + 4 * (wslot >> 4) + (wslot & 3)) * 16;
    const char* hsrc[HUPW];
#pragma unroll
    for (int i = 0; i < HUPW; ++i) {
        int u = wave + NW * i;
        if (u >= HU) u = HU - 1;                          // never issued (dma_full is false); keeps the address valid
        const int b = u >> 1, half = u & 1;
        if (b < MB) {
            int rb = (m0 >> 5) + b;
            if (rb > rb_last) rb = rb_last;               // M tail: re-read the last block (its results are not stored)
            hsrc[i] = (const char*)p.A + ((size_t)rb * KC) * 512 + half * 1024 + lane * 16;
        } else {
            hsrc[i] = (const char*)p.W + ((size_t)((n0 >> 5) + b - MB) * KC) * 512 + half * 1024 + wsrc;
        }
    }
    auto hpiece = [&](int h, int i) {                     // piece i of half tile h
        __builtin_amdgcn_global_load_lds((gbl16_void_t*)(hsrc[i] + (size_t)h * 2048), (lds16_void_t*)(smem + (h & 3) * SLOT + (wave + NW * i) * 1024), 16, 0, 0);
    };
    auto hstage = [&](int h) {
#pragma unroll
        for (int i = 0; i < HUPW; ++i) {
            if (i < HUPW - 1 || dma_full) hpiece(h, i);
        }
    };
    // own DMA groups still allowed in flight: `young` groups of (HUPW or HUPW - 1) loads
    auto wait_dma = [&](int young) {
        if (young >= 2) { if (dma_full) b16_wait_vmcnt<2 * HUPW>(); else b16_wait_vmcnt<2 * (HUPW - 1)>(); }
        else if (young == 1) { if (dma_full) b16_wait_vmcnt<HUPW>(); else b16_wait_vmcnt<HUPW - 1>(); }
        else b16_wait_vmcnt<0>();
    };

    // acc[i][j][mh][nh]: rows 16 mh + l15 of row block i, columns 8 g + 4 nh + (0..3) of column block j
    f32x4_t acc[cfg::MIMAX][NJ][2][2];
#pragma unroll
    for (int i = 0; i < cfg::MIMAX; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[i][j][a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    hstage(0);
    if (H > 1) hstage(1);
    if (H > 2) hstage(2);
    wait_dma(H > 2 ? 2 : H - 1);
    __builtin_amdgcn_s_barrier();
    BLK16_STAMP(1);

    // ONE barrier per half K tile; the two groups walk a slot in opposite order:
    //   slot k:   group 0: MFMA(k), MEM(k+1)      group 1: MEM(k+1), MFMA(k+1)
    // (hazards: gemm_blk_impl.h -- the ring and the order of its accesses are unchanged)
    auto main_loop = [&](auto miw_tag) {
        constexpr int MIW = decltype(miw_tag)::value;
        const uint32_t a_b = lds0 + (wm * MI0) * 2048 + g * 512 + l15 * 16;
        const uint32_t b_b = lds0 + (MB + wn * NJ) * 2048 + g * 512 + l15 * 16;
        bf16x8_t fa[MIW][2], fb[NJ][2];                    // [.][mh] rows 16 mh.. of the A block;  [.][nh] LDS slots 16 nh.. of the W block
        auto MEM = [&](int x) {
            const uint32_t sa = a_b + (x & 3) * SLOT, sb = b_b + (x & 3) * SLOT;
            fb[0][0] = b16_lds_read128<0>(sb); fb[1][0] = b16_lds_read128<2048>(sb);
            fa[0][0] = b16_lds_read128<0>(sa);
            if constexpr (MIW > 1) fa[1][0] = b16_lds_read128<2048>(sa);
            if constexpr (MIW > 2) fa[2][0] = b16_lds_read128<4096>(sa);
            if constexpr (MIW > 3) fa[3][0] = b16_lds_read128<6144>(sa);
            if constexpr (MIW > 4) fa[4][0] = b16_lds_read128<8192>(sa);
            fb[0][1] = b16_lds_read128<256>(sb); fb[1][1] = b16_lds_read128<2048 + 256>(sb);
            fa[0][1] = b16_lds_read128<256>(sa);
            if constexpr (MIW > 1) fa[1][1] = b16_lds_read128<2048 + 256>(sa);
            if constexpr (MIW > 2) fa[2][1] = b16_lds_read128<4096 + 256>(sa);
            if constexpr (MIW > 3) fa[3][1] = b16_lds_read128<6144 + 256>(sa);
            if constexpr (MIW > 4) fa[4][1] = b16_lds_read128<8192 + 256>(sa);
            if (x + 3 < H) hstage(x + 3);
            wait_dma(H - 2 - x);                                       // own share of half tile x + 1 has landed (x + 2, x + 3 may fly)
            b16_wait_lgkmcnt<0>();
            __builtin_amdgcn_sched_barrier(0);
        };
        auto MFMA = [&]() {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int i = 0; i < MIW; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int b = 0; b < 2; ++b)
                            acc[i][j][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j][b], fa[i][a], acc[i][j][a][b], 0, 0, 0);
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

    BLK16_STAMP(2);
    // ---- epilogue: straight from the accumulators.  Lane (g, l15) owns rows 16 mh + l15 of its row blocks and the 8 consecutive columns
    // 8 g .. 8 g + 7 of each column block (nh = 0: the first four, nh = 1: the last four) = 16 bytes of bf16 unit g / two fp32 units 2 g, 2 g + 1.
    const int miw = wm == 0 ? MI0 : MI1;
    const int rb0 = (m0 >> 5) + (wm == 0 ? 0 : MI0);                   // first row block of this wave
    const int rw0 = wm == 0 ? 0 : MI0 * 32;                           // its first row inside the tile
    const int nb0 = n0 + wn * (NJ * 32);
    const float* sBias = (const float*)(smem + cfg::BIAS_OFF) + wn * (NJ * 32);
    if constexpr (EPI == 0 || EPI == 1) {
        const int NC8 = p.N >> 3;
        const bool fold = p.stats_in != nullptr;                        // LayerNorm folded into this GEMM: per-row (rstd, rstd * mean)
        float rs[cfg::MIMAX][2], rm[cfg::MIMAX][2];
#pragma unroll
        for (int i = 0; i < cfg::MIMAX; ++i) { rs[i][0] = rs[i][1] = 1.f; rm[i][0] = rm[i][1] = 0.f; }
        if (fold) {
            const int S3 = p.K >> 8;                                     // partial pairs per row (one per 256-column tile of the producer)
            const float invC = 1.0f / (float)p.K;
#pragma unroll
            for (int i = 0; i < cfg::MIMAX; ++i) {
                if (i >= miw) continue;
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const float2* sp = (const float2*)(smem + cfg::STAT_OFF) + (rw0 + i * 32 + 16 * a + l15) * 4;
                    float sx = 0.f, sxx = 0.f;
                    for (int t = 0; t < S3; ++t) { const float2 v = sp[t]; sx += v.x; sxx += v.y; }
                    const float mean = sx * invC;
                    const float var = fmaxf(fmaf(-mean, mean, sxx * invC), 0.f);
                    rs[i][a] = 1.0f / sqrtf(var + p.ln_eps);
                    rm[i][a] = rs[i][a] * mean;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float4 bq[2], cq[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                bq[b] = *(const float4*)(sBias + j * 32 + 8 * g + 4 * b);
                cq[b] = fold ? *(const float4*)(sBias + BN + j * 32 + 8 * g + 4 * b) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int i = 0; i < cfg::MIMAX; ++i) {
                if (i >= miw || rb0 + i > rb_last) continue;
                char* unitp = (char*)p.C + ((size_t)(rb0 + i) * NC8 + ((nb0 + j * 32) >> 3) + g) * 512 + l15 * 16;
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    uint32_t pk[4];
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        // plain: acc + bias;  folded LayerNorm: rstd * acc + (bias' - rstd * mean * colsum)   (rs = 1, rm = 0, cq = 0 when not folded)
                        const f32x4_t c = acc[i][j][a][b];
                        f32x2_t v0 = {fmaf(c[0], rs[i][a], fmaf(-rm[i][a], cq[b].x, bq[b].x)), fmaf(c[1], rs[i][a], fmaf(-rm[i][a], cq[b].y, bq[b].y))};
                        f32x2_t v1 = {fmaf(c[2], rs[i][a], fmaf(-rm[i][a], cq[b].z, bq[b].z)), fmaf(c[3], rs[i][a], fmaf(-rm[i][a], cq[b].w, bq[b].w))};
                        if constexpr (EPI == 1) { v0 = gelu_fast2(v0); v1 = gelu_fast2(v1); }
                        pk[2 * b] = pack_bf16x2(v0.x, v0.y); pk[2 * b + 1] = pack_bf16x2(v1.x, v1.y);
                    }
                    *(uint4*)(unitp + a * 256) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                }
            }
        }
    } else {
        const int NC4 = p.N >> 2;
        const bool emit = p.xhat != nullptr;                            // also write bf16(C) as the next GEMM's operand + row partial sums
        // row partial sums per PAIR of column blocks (64 columns: one wave column of the eight-wave kernel) -- the four-wave body keeps two of
        // them per row so that the sums are taken in the same order whatever kernel ran the tile
        float sx[cfg::MIMAX][2][NJ / 2], sxx[cfg::MIMAX][2][NJ / 2], sh[cfg::MIMAX][2];
#pragma unroll
        for (int i = 0; i < cfg::MIMAX; ++i)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                sh[i][a] = 0.f;
#pragma unroll
                for (int q = 0; q < NJ / 2; ++q) { sx[i][a][q] = 0.f; sxx[i][a][q] = 0.f; }
            }
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
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int m = (rb0 + i) * 32 + 16 * a + l15;
                    float s = p.shift ? p.shift[m] : 0.f;
                    if (p.shift_stats) {
                        float t = 0.f;
                        for (int u = 0; u < S3; ++u) t += p.shift_stats[((size_t)m * S3 + u) * 2];
                        s = fmaf(t, invN, s);
                    }
                    sh[i][a] = s;
                    if (p.shift_out && tn == 0 && wn == 0 && g == 0) p.shift_out[m] = s;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float4 bq[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) bq[b] = *(const float4*)(sBias + j * 32 + 8 * g + 4 * b);
#pragma unroll
            for (int i = 0; i < cfg::MIMAX; ++i) {
                if (i >= miw || rb0 + i > rb_last) continue;
                // fp32 units 2 g (nh = 0) and 2 g + 1 (nh = 1) of this column block; row 16 a + l15
                const size_t off = ((size_t)(rb0 + i) * NC4 + ((nb0 + j * 32) >> 2) + 2 * g) * 512 + l15 * 16;
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    float4 rv[2];
                    if constexpr (EPI == 2) {
#pragma unroll
                        for (int b = 0; b < 2; ++b) rv[b] = *(const float4*)((const char*)p.res + off + b * 512 + a * 256);
                    } else {                                            // EPI 3: row-major residual, row = m % res_rows (pos embed, vit.py:320)
                        const int m = (rb0 + i) * 32 + 16 * a + l15;
                        const float* rr = p.res + (size_t)(m % p.res_rows) * p.N + nb0 + j * 32 + 8 * g;
#pragma unroll
                        for (int b = 0; b < 2; ++b) rv[b] = *(const float4*)(rr + 4 * b);
                    }
                    uint32_t pk[4];
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const f32x4_t c = acc[i][j][a][b];
                        float4 o;
                        o.x = c[0] + bq[b].x + rv[b].x; o.y = c[1] + bq[b].y + rv[b].y;
                        o.z = c[2] + bq[b].z + rv[b].z; o.w = c[3] + bq[b].w + rv[b].w;
                        *(float4*)((char*)p.C + off + b * 512 + a * 256) = o;
                        if (emit) {
                            // explicit order / explicit fma: every tile instantiation must produce the same bits for a row (batch-independence tests)
                            o.x -= sh[i][a]; o.y -= sh[i][a]; o.z -= sh[i][a]; o.w -= sh[i][a];
                            float& ax = sx[i][a][j >> 1];
                            float& axx = sxx[i][a][j >> 1];
                            ax += o.x; ax += o.y; ax += o.z; ax += o.w;
                            axx = fmaf(o.x, o.x, axx); axx = fmaf(o.y, o.y, axx);
                            axx = fmaf(o.z, o.z, axx); axx = fmaf(o.w, o.w, axx);
                            pk[2 * b] = pack_bf16x2(o.x, o.y); pk[2 * b + 1] = pack_bf16x2(o.z, o.w);
                        }
                    }
                    if (emit)
                        *(uint4*)((char*)p.xhat + ((size_t)(rb0 + i) * (p.N >> 3) + ((nb0 + j * 32) >> 3) + g) * 512 + (16 * a + l15) * 16) =
                            make_uint4(pk[0], pk[1], pk[2], pk[3]);
                }
            }
        }
        if (emit) {      // block-uniform
            // per row: (sum x, sum x^2) of the stored fp32 values over this tile's 256 columns = the 4 column groups of a wave (lanes 16 apart,
            // fixed tree), then the 4 wave columns through LDS in a fixed order (deterministic); one pair per row and column tile goes to
            // stats_out [rows][N/256][2]
            float2* sRed = (float2*)(smem + cfg::STAT_OFF);
#pragma unroll
            for (int i = 0; i < cfg::MIMAX; ++i) {
                if (i >= miw) continue;
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int q = 0; q < NJ / 2; ++q) {
                        float u = sx[i][a][q] + __shfl_xor(sx[i][a][q], 16, 64), v = sxx[i][a][q] + __shfl_xor(sxx[i][a][q], 16, 64);
                        u += __shfl_xor(u, 32, 64); v += __shfl_xor(v, 32, 64);
                        if (g == 0) sRed[(rw0 + i * 32 + 16 * a + l15) * 4 + wn * (NJ / 2) + q] = make_float2(u, v);
                    }
            }
            __syncthreads();
            const int S3 = p.N >> 8;
            for (int r = tid; r < BM; r += NT) {
                if ((m0 >> 5) + (r >> 5) > rb_last) continue;
                const float2 v0 = sRed[r * 4], v1 = sRed[r * 4 + 1], v2 = sRed[r * 4 + 2], v3 = sRed[r * 4 + 3];
                *(float2*)(p.stats_out + ((size_t)(m0 + r) * S3 + tn) * 2) = make_float2((v0.x + v1.x) + (v2.x + v3.x), (v0.y + v1.y) + (v2.y + v3.y));
            }
        }
    }
#ifdef WHMR_BLK_STAMPS
    BLK16_STAMP(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BLK16_STAMP(4);
    BLK16_STAMP_RAW(7, __builtin_amdgcn_s_memtime());
#endif
}

template <int MI0, int MI1, int EPI, int SCHED>
__global__ __launch_bounds__(512, 2) void gemm_blk16_kernel(const whmr_gemm_blk_desc p BLK16_STAMP_PARAM) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm_blk16_body<MI0, MI1, EPI, SCHED, 8>(p, smem, -1 BLK16_STAMP_PASS);
}

#ifndef WHMR_BLK_STAMPS
// ---- PILOT (round 6; off by default, WHMR_BLK_CHAIN=1): fc1 -> fc2 of one transformer layer as ONE persistent launch.
// grid = the CU count; workgroup w walks items w, w + grid, ... of the static list [fc1 tiles (XCD-aware order) | fc2 tiles]; an fc1 tile ARRIVES on the
// counter of its row panel when all its stores are visible device-wide (per-thread release fence at agent scope, barrier, one atomic add); an fc2 tile
// WAITS until the fc1 panels that hold its rows have seen all their column tiles, then takes an acquire fence (the XCD's L2 must not serve stale lines
// of the hidden activations: the buffer is reused every layer).  Every wait targets EARLIER items of the list and a workgroup walks its items in order,
// so the list cannot deadlock once every workgroup has started; a workgroup that has not been scheduled yet (foreign kernels on its CU) is what the
// BOUNDED spin covers: after `spin_limit` ticks of the 100 MHz counter the waiter raises *err and goes on (wrong results, flagged -- never a hang).
// The same tile bodies as the two launches: bit-identical outputs.
template <int A0, int A1, int B0, int B1>
__global__ __launch_bounds__(512, 2) void gemm_blk16_chain_kernel(const whmr_gemm_blk_desc p1, const whmr_gemm_blk_desc p2, unsigned* __restrict__ cnt,
                                                                  int* __restrict__ err, unsigned long long spin_limit, int lab) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM1 = (A0 + A1) * 32, BM2 = (B0 + B1) * 32;
    const int tn1 = p1.N / 256, tm1 = (p1.M + BM1 - 1) / BM1, n1 = tn1 * tm1;
    const int tn2 = p2.N / 256, tm2 = (p2.M + BM2 - 1) / BM2, n2 = tn2 * tm2;
    // (two loops, not one loop with a branch: with both tile bodies inside one loop the two descriptors' ~100 scalar registers stay live together, spill to
    //  vector lanes and push the kernel to 256 VGPRs + scratch; the walk order is the same -- every fc1 item of the list precedes every fc2 item)
    int item = blockIdx.x;
    for (; item < n1; item += gridDim.x) {
        const int lid = xcd_remap(item, n1);
        gemm_blk16_body<A0, A1, 1, 1, 8>(p1, smem, lid);
        // (lab != 0, tools/r6_chain_ab.py only: workgroup-scope fences -- NOT correct across the XCD-private L2s; it prices the agent-scope fences)
        if (lab) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");                   // this wave's stores of the tile: complete and written back
        __syncthreads();                                                          // (also: the next item's prologue overwrites the bias / statistics area and the ring)
        if (threadIdx.x == 0) __hip_atomic_fetch_add(&cnt[lid / tn1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (; item < n1 + n2; item += gridDim.x) {
        const int lid = xcd_remap(item - n1, n2);
        const int r0 = (lid / tn2) * BM2;
        int r1 = r0 + BM2 - 1;
        if (r1 > p2.M - 1) r1 = p2.M - 1;
        if (threadIdx.x == 0) {
            int bad = 0;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            for (int pn = r0 / BM1; pn <= r1 / BM1 && !bad; ++pn) {
                while (__hip_atomic_load(&cnt[pn], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)tn1) {
                    __builtin_amdgcn_s_sleep(8);
                    if (__builtin_amdgcn_s_memrealtime() - t0 > spin_limit) { bad = 1; break; }
                }
            }
            if (bad) *err = 1;
        }
        __syncthreads();
        if (lab) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        gemm_blk16_body<B0, B1, 2, 1, 8>(p2, smem, lid);
        __syncthreads();
    }
}

template <int A0, int A1, int B0, int B1>
static int launch_blk16_chain(const whmr_gemm_blk_desc& p1, const whmr_gemm_blk_desc& p2, unsigned* cnt, int* err, int grid, int lab, hipStream_t st) {
    using c1 = blk16_cfg<A0, A1>;
    using c2 = blk16_cfg<B0, B1>;
    constexpr int LDS = c1::LDS > c2::LDS ? c1::LDS : c2::LDS;
    auto kern = gemm_blk16_chain_kernel<A0, A1, B0, B1>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const int panels = (p1.M + c1::BM - 1) / c1::BM;
    hipError_t e = hipMemsetAsync(cnt, 0, sizeof(unsigned) * panels, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, st, p1, p2, cnt, err, 2000000ull /* 20 ms */, lab);
    WHMR_CHECK_LAUNCH();
    return 0;
}
#endif

template <int MI0, int MI1, int EPI, int SCHED>
static int launch_blk16_s(const whmr_gemm_blk_desc& p, hipStream_t st) {
    using cfg = blk16_cfg<MI0, MI1>;
    auto kern = gemm_blk16_kernel<MI0, MI1, EPI, SCHED>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, cfg::LDS);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const int tiles = ((p.M + cfg::BM - 1) / cfg::BM) * (p.N / cfg::BN);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), cfg::LDS, st, p BLK16_STAMP_ARG(tiles));
    WHMR_CHECK_LAUNCH();
    return 0;
}

template <int MI0, int MI1>
static int launch_blk16_epi(const whmr_gemm_blk_desc& p, hipStream_t st) {
    switch (p.epi) {
        case 0: return launch_blk16_s<MI0, MI1, 0, 1>(p, st);
        case 1: return launch_blk16_s<MI0, MI1, 1, 1>(p, st);
        case 2: return launch_blk16_s<MI0, MI1, 2, 1>(p, st);
        case 3: return launch_blk16_s<MI0, MI1, 3, 1>(p, st);
    }
    return (int)hipErrorInvalidValue;
}

// Tile heights (x 256 columns): the wave rows own MI0 and MI1 row blocks.
static int blk16_launch_tile(const whmr_gemm_blk_desc& p, int tile, hipStream_t st) {
    switch (tile) {
        case 0x44: return launch_blk16_epi<4, 4>(p, st);      // 256 x 256
        case 0x55: return launch_blk16_epi<5, 5>(p, st);      // 320 x 256
        case 0x43: return launch_blk16_epi<4, 3>(p, st);      // 224 x 256
        case 0x33: return launch_blk16_epi<3, 3>(p, st);      // 192 x 256
        case 0x32: return launch_blk16_epi<3, 2>(p, st);      // 160 x 256
        case 0x22: return launch_blk16_epi<2, 2>(p, st);      // 128 x 256
        case 0x54: return launch_blk16_epi<5, 4>(p, st);      // 288 x 256
        case 0x21: return launch_blk16_epi<2, 1>(p, st);      // 96 x 256: ViT-L at 32 crops (6144 tokens) x N = 1024 is exactly 256 such tiles
    }
    return (int)hipErrorInvalidValue;
}
