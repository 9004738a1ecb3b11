// Generic large-wave-tile bf16 MFMA GEMM for gfx950 (v_mfma_f32_32x32x16_bf16), templated on the block tile
// BM x BN x BK and the wave grid WM x WN (wave tile (BM/WM) x (BN/WN), at least 64 wide so that LDS fragment traffic
// stays at <= 24 B/clk/SIMD).  Instantiated as:
//   <128,256,32,1,4>  4 waves, 48 KiB LDS  -> 2-3 independent blocks per CU: one block's (memory-bound) epilogue
//                     overlaps the other's MFMA main loop.  Default for the ViT shapes.
//   <256,256,64,2,4>, <192,256,64,2,4>  8 waves, 112-128 KiB LDS, one block per CU (A/B reference points).
// Differences from gemm_bf16.hip (128x128, 64x64 wave tiles): larger wave tile; MFMA operands are swapped
// (D^T = W_tile . A_tile^T) so that each lane owns 4 consecutive output COLUMNS per register quad -- the epilogue packs
// them (bias + activation applied in registers) into 16-B / 8-B LDS writes, and leaves through whole-row 16-B stores.
#include <type_traits>
#include "common.h"
#include "gemm_params.h"

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

template <int BM, int BN, int BK, int WM, int WN, int NS>
struct big_cfg {
    static constexpr int THREADS = 64 * WM * WN;
    static constexpr int CPR = BK / 8;                     // 16-B chunks per tile row
    static constexpr int RP = THREADS / CPR;               // tile rows staged per pass
    static constexpr int AP = BM / RP, BP = BN / RP;       // staging passes (= global_load_lds per thread per K step)
    static constexpr int WTM = BM / WM, WTN = BN / WN;     // wave tile
    static constexpr int MI = WTM / 32, NJ = WTN / 32;
    static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
    static constexpr int CROWS = WM * 32;                  // C rows per epilogue pass
    static constexpr int CLD_F32 = BN + 4;                 // fp32 staging row stride (dwords): conflict-free b128 writes
    static constexpr int CLD_BF16 = BN + 4;                // bf16 staging row stride (elements): (BN+4)/2 dwords = 2 mod 32
    static constexpr int EPI_BYTES = CROWS * CLD_F32 * 4;
    static constexpr int LDS = (NS * STAGE > EPI_BYTES) ? NS * STAGE : EPI_BYTES;
    // bf16-output kernels stage the finished tile as bf16 (bias / activation already applied): all MI passes at once when they
    // fit in 160 KiB, else as many as fit in the main-loop allocation
    static constexpr int EPI16_ROW = BN * 2 + 16;                                        // bytes per staged row: 16-B aligned, 4 dwords of skew per row (conflict-free 8-B writes)
    static constexpr int EPI16_ALL = MI * CROWS * EPI16_ROW;
    static constexpr int LDS16 = (EPI16_ALL <= 160 * 1024 && EPI16_ALL > LDS) ? EPI16_ALL : LDS;
    static constexpr int GP16 = (LDS16 / (CROWS * EPI16_ROW)) < MI ? (LDS16 / (CROWS * EPI16_ROW)) : MI;   // passes staged per group
    static constexpr int G = AP + BP;                      // global_load_lds per thread per K step
    static_assert(BM % RP == 0 && BN % RP == 0, "staging passes must tile the block");
    static_assert(BK == 32 || BK == 64, "BK");
    static_assert(NS >= 2 && NS <= 4, "wait_step handles at most 2 younger groups in flight");
};

// XOR swizzle of the 16-B chunk index inside a tile row, keyed by the row (conflict-free ds_read_b128 fragment reads)
template <int BK> __device__ __forceinline__ int swz(int row) { return BK == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3); }

#ifndef DIRECT_EPI
#define DIRECT_EPI 0
#endif
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_lgkmcnt() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
// Fragment read hidden from hipcc's waitcnt bookkeeping (it would otherwise wait lgkmcnt(0) across the loop back-edge
// and serialise the register double-buffer): destination is valid only after the matching counted wait_lgkmcnt + sched_barrier.
__device__ __forceinline__ bf16x8_t lds_read128(uint32_t addr) {
    bf16x8_t v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}

template <int BM, int BN, int BK, int WM, int WN, int NS, int MINW, int PP, int OUT_BF16, int ACT, bool GATHER>
__global__ __launch_bounds__(64 * WM * WN, MINW) void gemm_bf16_big_kernel(const whmr_gemm p) {
    using cfg = big_cfg<BM, BN, BK, WM, WN, NS>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int AP = cfg::AP, BP = cfg::BP, MI = cfg::MI, NJ = cfg::NJ, CPR = cfg::CPR, RP = cfg::RP;
    // PP == 3: "trimmed" tile.  The LDS / register layout is the BM-row one, but the last 32-row block of the LAST wave row is
    // dead (not multiplied, not stored) and tiles advance by BM - 32 rows: 160 x 256 and 224 x 256 tiles out of the 192 / 256
    // kernels.  Each SIMD hosts one wave of every wave row, so the SIMDs stay balanced (MI + MI - 1 row blocks each).  Lets the
    // chooser fit 12544 x 768 into ONE round of 237 tiles (was 198 of 192 rows: 23 % of the CUs idle) and 12544 x 2304 into two
    // rounds of 224-row tiles (504 of 512 slots).
    constexpr int TRIM = (PP == 3) ? 32 : 0, BME = BM - TRIM;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = TRIM ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);   // provably wave-uniform: the trimmed main loop branches on it around s_barrier
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BME - 1) / BME;
    // Sub-pixel deconv: the phase is the FASTEST tile index, so the 4 phases of one M tile (which gather the same input
    // neighbourhood and write interleaved output pixels) run back to back on one XCD: shared A reads hit its L2 and the
    // interleaved 512-B pixel rows of the output meet in cache before they go to HBM.
    const int nph = (GATHER && p.n_phase > 1) ? p.n_phase : 1;
    int lid = xcd_remap(blockIdx.x, tiles_m * tiles_n * nph);
    const int phase = lid % nph;
    lid /= nph;
    const int tm = lid / tiles_n, tn = lid % tiles_n;
    const int m0 = tm * BME, n0 = tn * BN;
    const int m_end = (m0 + BME < p.M) ? m0 + BME : p.M;      // first row this tile does NOT own
    const bf16_t* __restrict__ A = (const bf16_t*)p.A;
    const bf16_t* __restrict__ W = (const bf16_t*)p.W;
    int PH = p.PH, PW = p.PW;
    int64_t c_off = p.c_off;
    if constexpr (GATHER) {
        if (p.n_phase > 1) {                  // sub-pixel deconv phase of this block (gemm_params.h)
            const int py = phase >> 1, px = phase & 1;
            W += (size_t)phase * p.phase_w_stride;
            PH -= py; PW -= px;
            c_off += py * p.phase_cy + px * p.phase_cx;
        }
    }

    // bf16-output epilogue: this tile's bias slice is fetched NOW (one value per thread, in flight under the whole main loop)
    // and parked in LDS behind the staging area after the loop -- the epilogue then reads it with ds_read instead of paying
    // a global round trip per tile while the matrix pipes idle.
    float bias_early = 0.f;
    if constexpr (OUT_BF16) {
        if (p.bias && tid < BN && n0 + tid < p.N) bias_early = p.bias[n0 + tid];
    }
    // ---- staging geometry: chunk c = tid + THREADS*i -> tile row tid/CPR + RP*i, physical slot tid%CPR
    const int srow = tid / CPR, pc = tid % CPR;
    const bf16_t* a_src[AP];
    int a_y[AP], a_x[AP];
    const bf16_t* b_src[BP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int row = srow + RP * i;
        const int lc = pc ^ swz<BK>(row);
        int m = m0 + row;
        if (m > p.M - 1) m = p.M - 1;
        if constexpr (GATHER) {
            const int ohw = p.OH * p.OW;
            const int b = m / ohw, rem = m - b * ohw;
            const int oy = rem / p.OW, ox = rem - oy * p.OW;
            a_y[i] = oy * p.SH - PH;
            a_x[i] = ox * p.SW - PW;
            a_src[i] = A + (size_t)b * p.IH * p.IW * p.Cin + lc * 8;
        } else {
            a_src[i] = A + (size_t)m * p.lda + lc * 8;
        }
    }
#pragma unroll
    for (int i = 0; i < BP; ++i) {
        const int row = srow + RP * i;
        const int lc = pc ^ swz<BK>(row);
        int nr = n0 + row;
        if (nr > p.N - 1) nr = p.N - 1;
        b_src[i] = W + (size_t)nr * p.K + lc * 8;
    }

    // split-K slice of this block (whmr_gemm_bf16 sets split_k for few-tile, deep-K shapes): raw fp32 partial sums go to
    // workspace slice blockIdx.z; bias / skip / activation happen in splitk_epilogue_kernel.
    const int k_first = p.split_k ? (int)(blockIdx.z * p.split_k) : 0;
    // amask / bmask: which staging passes (64-row slabs for BK = 64) of the A / B tile to issue
    auto stage_sel = [&](int kt, int s, unsigned amask, unsigned bmask) {
        char* sa = smem + s * cfg::STAGE;
        char* sb = sa + cfg::A_BYTES;
        const int k0 = k_first + kt * BK;
        int ky = 0, kx = 0, ci0 = 0;
        if constexpr (GATHER) {
            int tap;
            if (p.epi_flags & 8) {               // chunk-major K order (ci chunk of 64, ky, kx, ci in chunk): see gemm_params.h
                const int ntaps = p.K / p.Cin, blk = k0 >> 6, chunk = blk / ntaps;
                tap = blk - chunk * ntaps;
                ci0 = (chunk << 6) + (k0 & 63);
            } else {
                tap = k0 / p.Cin;
                ci0 = k0 - tap * p.Cin;
            }
            ky = tap / p.KW;
            kx = tap - ky * p.KW;
        }
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            if (!((amask >> i) & 1)) continue;
            const bf16_t* src;
            if constexpr (GATHER) {
                const int iy = a_y[i] + ky, ix = a_x[i] + kx;
                const bool ok = (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
                src = ok ? a_src[i] + ((size_t)iy * p.IW + ix) * p.Cin + ci0 : (const bf16_t*)p.zeros + pc * 8;
            } else {
                src = a_src[i] + k0;
            }
            __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(sa + (wave * 64 + cfg::THREADS * i) * 16), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < BP; ++i) {
            if (!((bmask >> i) & 1)) continue;
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(b_src[i] + k0), (lds_void_t*)(sb + (wave * 64 + cfg::THREADS * i) * 16), 16, 0, 0);
        }
    };
    auto stage = [&](int kt, int s) { stage_sel(kt, s, ~0u, ~0u); };

    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hi = lane >> 5;
    // acc[i][j][r]: output row m = wm*WTM + i*32 + l31, column n = wn*WTN + j*32 + (r&3) + 8*(r>>2) + 4*hi  (swapped operands)
    f32x16_t acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int a_off[MI], a_sw[MI], b_off[NJ], b_sw[NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int ra = wm * cfg::WTM + i * 32 + l31;
        a_off[i] = ra * (BK * 2); a_sw[i] = swz<BK>(ra);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int rb = wn * cfg::WTN + j * 32 + l31;
        b_off[j] = rb * (BK * 2); b_sw[j] = swz<BK>(rb);
    }

    // ---- main loop: NS-deep LDS ring, loads issued NS-1 K steps ahead.  global_load_lds has no register result, so the
    // compiler tracks nothing: completion is enforced by hand with a COUNTED s_waitcnt vmcnt (loads return in order) and a
    // raw s_barrier -- one barrier per K step, never a drain to zero in steady state.
    // Fragments are software-pipelined through two register sets: the ds_reads of sub-step kk+1 are issued before the
    // MFMAs of sub-step kk, so LDS latency hides under the wave's own 8 MFMAs instead of idling the matrix pipe while
    // both waves of a SIMD wait in lockstep.  With NS >= 3 the K-step barrier sits before the LAST sub-step's MFMAs:
    // after it the wave issues the next stage's DMA and the first fragments of step kt+1, then still has MFMAs queued.
    constexpr int KK = BK / 16;
    const int nkt = (p.split_k ? min((int)p.split_k, p.K - k_first) : p.K) / BK;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_t*)smem;
    // MIW = row blocks this wave multiplies (MI, or MI - 1 for the last wave row of a trimmed tile): the counted lgkmcnt waits
    // depend on it, so the loop body is instantiated per value and selected by a wave-uniform branch (same barrier count).
    auto main_loop = [&](auto miw_tag) {
    constexpr int MIW = decltype(miw_tag)::value;
    bf16x8_t af[2][MIW], bfr[2][NJ];
    constexpr int NF = MIW + NJ;          // ds_read_b128 per fragment set
    auto load_frags = [&](int buf, int kk, int set) {
        const uint32_t sa = lds0 + buf * cfg::STAGE;
        const uint32_t sb = sa + cfg::A_BYTES;
        const int c = kk * 2 + hi;
#pragma unroll
        for (int j = 0; j < NJ; ++j) bfr[set][j] = lds_read128(sb + b_off[j] + ((c ^ b_sw[j]) << 4));
#pragma unroll
        for (int i = 0; i < MIW; ++i) af[set][i] = lds_read128(sa + a_off[i] + ((c ^ a_sw[i]) << 4));
    };
    auto mfmas = [&](int set) {
#pragma unroll
        for (int i = 0; i < MIW; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[set][j], af[set][i], acc[i][j], 0, 0, 0);
    };
    // Step `step` has landed once at most (younger groups already in flight) x G loads are outstanding.  `ahead` = how many
    // younger groups can be in flight at the wait: NS-2 when the wait precedes the step's own compute (2-stage scheme),
    // NS-3 for the mid-step wait of the deep scheme (the next stage is issued only after the barrier that follows it).
    auto wait_step = [&](int step, int ahead) {
        int younger = nkt - 1 - step;
        if (younger > ahead) younger = ahead;
        if (younger >= 2) wait_vmcnt<cfg::G * 2>();
        else if (younger == 1) wait_vmcnt<cfg::G>();
        else wait_vmcnt<0>();
    };
    if constexpr (PP == 2) {
        // ---- ping-pong main loop (opt-in tile 259: 256x256x64, 8 waves = two groups of 4, one wave of each group per SIMD).
        // A K tile is computed in 2 phases of 16 MFMAs (two 64x32 quadrants of the wave tile x K = 64); every phase is
        //   [ds_read the fragments the quadrants add | issue 4 global_load_lds | counted vmcnt] s_barrier [MFMAs] s_barrier
        // and group 1 runs one barrier behind group 0, so on every SIMD one wave is in its MFMA cluster while the other reads
        // LDS / issues DMA.  DMA units (16 KiB = 2 loads per thread): S1 = A rows {0-63,128-191} (the first-phase fragments of
        // both groups), S3 = the other A rows, S2a / S2b = B halves; each is re-staged into the region whose last ds_read
        // retired one barrier earlier and awaited one barrier before its first read (the other group issues half of every unit).
        static_assert(BM == 256 && BN == 256 && BK == 64 && WM == 2 && WN == 4 && NS == 2, "ping-pong schedule is written for 256x256x64 / 8 waves");
        bf16x8_t fa[2][4], fb0[4], fb1[4];
        auto rd_a = [&](int buf, int i, int kk) { return lds_read128(lds0 + buf * cfg::STAGE + a_off[i] + ((((kk << 1) + hi) ^ a_sw[i]) << 4)); };
        auto rd_b = [&](int buf, int j, int kk) { return lds_read128(lds0 + buf * cfg::STAGE + cfg::A_BYTES + b_off[j] + ((((kk << 1) + hi) ^ b_sw[j]) << 4)); };
        // prologue: K tiles 0 and 1 (except S3(1)), in the steady-state order
        stage_sel(0, 0, 0x5, 0x0); stage_sel(0, 0, 0x0, 0x3); stage_sel(0, 0, 0x0, 0xc); stage_sel(0, 0, 0xa, 0x0);
        if (nkt > 1) {
            stage_sel(1, 1, 0x5, 0x0); stage_sel(1, 1, 0x0, 0x3);
            wait_vmcnt<6>();
        } else {
            wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
        if (wm == 1) __builtin_amdgcn_s_barrier();            // group 1 runs one barrier behind
        {
            // two phases per K tile (16 MFMAs each).  Phase A reads A rows 0-63 + both B fragments
            // and computes quadrants 1-2; phase B reads A rows 64-127 and computes quadrants 3-4.  DMA: B(t) issues S1, S2a of
            // K tile t+2, A(t+1) issues S2b, S3 of t+2; waits vmcnt(6) in B (S1/S2 of t+1 landed) and vmcnt(8) in A (S3(t)).
            auto quad2 = [&](int i0, int j0, const bf16x8_t* fbx, const bf16x8_t* fby) {
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    acc[i0][j0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fbx[kk], fa[0][kk], acc[i0][j0], 0, 0, 0);
                    acc[i0 + 1][j0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fbx[kk], fa[1][kk], acc[i0 + 1][j0], 0, 0, 0);
                }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    acc[i0][j0 ^ 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fby[kk], fa[0][kk], acc[i0][j0 ^ 1], 0, 0, 0);
                    acc[i0 + 1][j0 ^ 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fby[kk], fa[1][kk], acc[i0 + 1][j0 ^ 1], 0, 0, 0);
                }
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
            };
            for (int t = 0; t < nkt; ++t) {
                const int buf = t & 1;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) fb0[kk] = rd_b(buf, 0, kk);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) { fa[0][kk] = rd_a(buf, 0, kk); fa[1][kk] = rd_a(buf, 1, kk); }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) fb1[kk] = rd_b(buf, 1, kk);
                if (t + 1 < nkt) { stage_sel(t + 1, buf ^ 1, 0x0, 0xc); stage_sel(t + 1, buf ^ 1, 0xa, 0x0); wait_vmcnt<8>(); } else { wait_vmcnt<0>(); }
                wait_lgkmcnt<0>();
                quad2(0, 0, fb0, fb1);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) { fa[0][kk] = rd_a(buf, 2, kk); fa[1][kk] = rd_a(buf, 3, kk); }
                if (t + 2 < nkt) { stage_sel(t + 2, buf, 0x5, 0x0); stage_sel(t + 2, buf, 0x0, 0x3); wait_vmcnt<6>(); } else { wait_vmcnt<0>(); }
                wait_lgkmcnt<0>();
                quad2(2, 1, fb1, fb0);
            }
        }
        if (wm == 0) __builtin_amdgcn_s_barrier();            // re-align the groups
    } else {
#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
        if (t < nkt) stage(t, t);
    if constexpr (NS >= 3) {
        wait_step(0, NS - 2);
        __builtin_amdgcn_s_barrier();
        load_frags(0, 0, 0);
        for (int kt = 0; kt < nkt; ++kt) {
            const int buf = kt % NS;
#pragma unroll
            for (int kk = 0; kk < KK - 1; ++kk) {
                load_frags(buf, kk + 1, (kk + 1) & 1);
                wait_lgkmcnt<NF>();                 // the older set (sub-step kk) has landed; kk+1 stays in flight
                __builtin_amdgcn_sched_barrier(0);
                mfmas(kk & 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (kt + 1 < nkt) {
                wait_step(kt + 1, NS - 3);
                __builtin_amdgcn_s_barrier();       // step kt+1 visible to all; everyone is past step kt-1's buffer
                if (kt + NS - 1 < nkt) stage(kt + NS - 1, (kt + NS - 1) % NS);
                load_frags((kt + 1) % NS, 0, KK & 1);
                wait_lgkmcnt<NF>();
            } else {
                wait_lgkmcnt<0>();
            }
            __builtin_amdgcn_sched_barrier(0);
            mfmas((KK - 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if constexpr (PP == 4) {
        // ---- 2-stage loop with the DMA issue SPREAD behind the MFMAs (one global_load_lds after every second MFMA of sub-steps 0-1)
        // instead of a burst of G loads per wave right after the barrier: the burst keeps all 8 waves in VMEM issue (the texture
        // addresser takes ~16 clk per 1-KiB wave instruction: 64 of them per K step = ~1000 clk with the matrix pipes idle).
        static_assert(NS == 2 && BK == 64, "spread-DMA schedule: 2 stages of BK = 64");
        for (int kt = 0; kt < nkt; ++kt) {
            wait_step(kt, 0);
            __builtin_amdgcn_s_barrier();
            const bool more = kt + 1 < nkt;
            const int nb = (kt + 1) & 1;
            load_frags(kt & 1, 0, 0);
            int d = 0;                            // next DMA unit (0..AP-1: A passes, AP..AP+BP-1: B passes)
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                if (kk + 1 < KK) { load_frags(kt & 1, kk + 1, (kk + 1) & 1); wait_lgkmcnt<NF>(); }
                else wait_lgkmcnt<0>();
                __builtin_amdgcn_sched_barrier(0);
                int n = 0;
#pragma unroll
                for (int i = 0; i < MIW; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[kk & 1][j], af[kk & 1][i], acc[i][j], 0, 0, 0);
                        ++n;
                        constexpr int PER = (cfg::G + 1) / 2;               // DMA units per sub-step (sub-steps 0 and 1 carry all G)
                        constexpr int EVERY = (MIW * NJ) / PER > 0 ? (MIW * NJ) / PER : 1;
                        if (kk < 2 && (n % EVERY) == 0 && d < cfg::G && d < (kk + 1) * PER) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (more) { if (d < AP) stage_sel(kt + 1, nb, 1u << d, 0u); else stage_sel(kt + 1, nb, 0u, 1u << (d - AP)); }
                            __builtin_amdgcn_sched_barrier(0);
                            ++d;
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        for (int kt = 0; kt < nkt; ++kt) {
            wait_step(kt, NS - 2);
            __builtin_amdgcn_s_barrier();        // everyone's step-kt data visible; everyone done reading buffer (kt-1) % NS
            if (kt + NS - 1 < nkt) stage(kt + NS - 1, (kt + NS - 1) % NS);
            load_frags(kt % NS, 0, 0);
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                if (kk + 1 < KK) { load_frags(kt % NS, kk + 1, (kk + 1) & 1); wait_lgkmcnt<NF>(); }
                else wait_lgkmcnt<0>();
                __builtin_amdgcn_sched_barrier(0);
                mfmas(kk & 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    }   // !PP
    };
    if constexpr (TRIM) {
        if (wave / WN == WM - 1) main_loop(std::integral_constant<int, MI - 1>{});
        else main_loop(std::integral_constant<int, MI>{});
    } else {
        main_loop(std::integral_constant<int, MI>{});
    }
    __syncthreads();                         // all fragment reads done before the epilogue reuses the LDS

    if (p.res_row_mod == -12345) {            // timing probe: main loop only (keeps the accumulators live)
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += acc[i][j][r];
        if (t == 12345.678f) ((float*)p.C)[0] = t;
        return;
    }

    // ---- bf16 outputs without a skip tensor (qkv, fc1 + GELU, deconvs, ResNet 3x3 / 1x1 convs): bias + activation in registers
    // (each lane owns 4 consecutive columns per register quad), pack to bf16, park the WHOLE tile in LDS as 8-B pieces, one
    // barrier, then every thread streams 16-B row chunks straight to global memory.  Half the LDS bytes of the fp32 slab
    // below, no per-pass barriers and no arithmetic in the store loop: the fp32-slab epilogue cost ~11 us of a 60 us qkv launch
    // with the global stores removed (tools/alias_probe.py), i.e. it was instruction / LDS bound, not HBM bound.
    if constexpr (OUT_BF16) {
        // bf16 skip tensor (ResNet); a GELU that must follow the skip add stays on the fp32-slab path
        const bool skip16 = p.residual && (p.epi_flags & 1) && !(p.ldr & 7) && p.res_row_mod <= 0 && p.res_row_mod != -2003 && !(ACT == 1 && (p.epi_flags & 2));
        if ((!p.residual || skip16) && !p.row_scale && !p.split_k && !(p.N & 7) && (p.c_mode == 1 || !(p.ldc & 7))) {
            const bool act_late = skip16 && (p.epi_flags & 2);               // ResNet: the skip is added BEFORE the activation -> activate in the store loop
            const bool dual = ACT == 1 && p.C2 && !p.residual && p.c_mode == 0;   // fc1 of the training forward: C2 = pre-activation, C = GELU of it (store loop)
            const bool gelu_mul = ACT == 0 && skip16 && (p.epi_flags & 128);  // residual = pre-activation Z: C = value * gelu'(Z)
            constexpr int ROWB = cfg::EPI16_ROW, GP = cfg::GP16, CPRW = BN / 8, RPI2 = cfg::THREADS / CPRW;
            void* const Cout16 = p.C;
            const bool nostore16 = p.res_row_mod == -2003;
            const bool spatial16 = GATHER && p.c_mode == 1;
            const int ohw16 = spatial16 ? p.OH * p.OW : 1;
            const float rcp_ohw16 = 1.0f / (float)ohw16, rcp_ow16 = spatial16 ? 1.0f / (float)p.OW : 1.0f;
            float* sBias = (float*)(smem + cfg::LDS16);                              // [BN] floats behind the staging area
            if (tid < BN) sBias[tid] = bias_early;
            wait_lgkmcnt<0>();
            __builtin_amdgcn_s_barrier();
            float4 bq[NJ][4];
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) bq[j][q] = *(const float4*)(sBias + wn * cfg::WTN + j * 32 + 8 * q + 4 * hi);
#pragma unroll
            for (int i0 = 0; i0 < MI; i0 += GP) {
                if (i0) { wait_lgkmcnt<0>(); __builtin_amdgcn_s_barrier(); }          // previous group fully read
#pragma unroll
                for (int i = i0; i < i0 + GP && i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float v[4] = {acc[i][j][4 * q] + bq[j][q].x, acc[i][j][4 * q + 1] + bq[j][q].y, acc[i][j][4 * q + 2] + bq[j][q].z,
                                          acc[i][j][4 * q + 3] + bq[j][q].w};
                            if (ACT == 1 && !act_late && !dual) {
                                const f32x2_t g0 = gelu_fast2(f32x2_t{v[0], v[1]}), g1 = gelu_fast2(f32x2_t{v[2], v[3]});
                                v[0] = g0.x; v[1] = g0.y; v[2] = g1.x; v[3] = g1.y;
                            }
                            if (ACT == 2 && !act_late) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                            *(uint2*)(smem + ((i - i0) * cfg::CROWS + wm * 32 + l31) * ROWB + (wn * cfg::WTN + j * 32 + 8 * q + 4 * hi) * 2) =
                                make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                        }
                wait_lgkmcnt<0>();
                __builtin_amdgcn_s_barrier();
                const int npass = (MI - i0) < GP ? (MI - i0) : GP;
                const int nrows = npass * cfg::CROWS;
                auto row_addr = [&](int m) -> size_t {
                    if (spatial16) {
                        int b = (int)((float)m * rcp_ohw16);
                        int rem = m - b * ohw16;
                        if (rem >= ohw16) { ++b; rem -= ohw16; } else if (rem < 0) { --b; rem += ohw16; }
                        int oy = (int)((float)rem * rcp_ow16);
                        int ox = rem - oy * p.OW;
                        if (ox >= p.OW) { ++oy; ox -= p.OW; } else if (ox < 0) { --oy; ox += p.OW; }
                        return (size_t)(c_off + b * p.osb + oy * p.osy + ox * p.osx);
                    }
                    return (size_t)m * p.ldc;
                };
                const int chunk = tid % CPRW, col = n0 + chunk * 8;
                if (dual) {                          // pre-activation to C2, GELU of the bf16-rounded value (what the backward differentiates) to C
#pragma unroll 2
                    for (int lr = tid / CPRW; lr < nrows; lr += RPI2) {
                        const int pi = lr / cfg::CROWS, within = lr - pi * cfg::CROWS;
                        const int m = m0 + (within >> 5) * cfg::WTM + (i0 + pi) * 32 + (within & 31);
                        if (m >= m_end || col >= p.N) continue;
                        const uint4 v = *(const uint4*)(smem + lr * ROWB + chunk * 16);
                        const size_t off = (size_t)m * p.ldc + col;
                        *(uint4*)((bf16_t*)p.C2 + off) = v;
                        const uint32_t a[4] = {v.x, v.y, v.z, v.w};
                        uint32_t o[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const f32x2_t g = gelu_fast2(f32x2_t{__uint_as_float(a[e] << 16), __uint_as_float(a[e] & 0xffff0000u)});
                            o[e] = pack_bf16x2(g.x, g.y);
                        }
                        *(uint4*)((bf16_t*)Cout16 + off) = make_uint4(o[0], o[1], o[2], o[3]);
                    }
                } else if (!skip16) {                // the hot loop (qkv, fc1, deconvs): LDS read -> global store, nothing else
#pragma unroll 4
                    for (int lr = tid / CPRW; lr < nrows; lr += RPI2) {
                        const int pi = lr / cfg::CROWS, within = lr - pi * cfg::CROWS;
                        const int m = m0 + (within >> 5) * cfg::WTM + (i0 + pi) * 32 + (within & 31);
                        if (m >= m_end || col >= p.N) continue;
                        const uint4 v = *(const uint4*)(smem + lr * ROWB + chunk * 16);
                        if (!nostore16 || v.x == 0x12345678u) *(uint4*)((bf16_t*)Cout16 + row_addr(m) + col) = v;
                    }
                } else {                             // + bf16 skip row (one 16-B load), late ReLU (ResNet), re-pack
#pragma unroll 2
                    for (int lr = tid / CPRW; lr < nrows; lr += RPI2) {
                        const int pi = lr / cfg::CROWS, within = lr - pi * cfg::CROWS;
                        const int m = m0 + (within >> 5) * cfg::WTM + (i0 + pi) * 32 + (within & 31);
                        if (m >= m_end || col >= p.N) continue;
                        const uint4 v = *(const uint4*)(smem + lr * ROWB + chunk * 16);
                        // epi_flags bit 2: the skip tensor is laid out like C (scatter mode: the in-place accumulate of a data gradient)
                        const uint4 sk = *(const uint4*)((const bf16_t*)p.residual + ((p.epi_flags & 4) ? row_addr(m) : (size_t)m * p.ldr) + col);
                        const uint32_t a[4] = {v.x, v.y, v.z, v.w}, b[4] = {sk.x, sk.y, sk.z, sk.w};
                        uint32_t o[4];
                        if (gelu_mul) {              // d pre = d hid * gelu'(pre): fc2's data gradient and the GELU backward in one pass
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const f32x2_t d = gelu_grad_fast2(f32x2_t{__uint_as_float(b[e] << 16), __uint_as_float(b[e] & 0xffff0000u)});
                                o[e] = pack_bf16x2(__uint_as_float(a[e] << 16) * d.x, __uint_as_float(a[e] & 0xffff0000u) * d.y);
                            }
                        } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float lo = __uint_as_float(a[e] << 16) + __uint_as_float(b[e] << 16);
                            float hi2 = __uint_as_float(a[e] & 0xffff0000u) + __uint_as_float(b[e] & 0xffff0000u);
                            if (ACT == 2 && act_late) { lo = fmaxf(lo, 0.f); hi2 = fmaxf(hi2, 0.f); }
                            o[e] = pack_bf16x2(lo, hi2);
                        }
                        }
                        *(uint4*)((bf16_t*)Cout16 + row_addr(m) + col) = make_uint4(o[0], o[1], o[2], o[3]);
                    }
                }
            }
            return;
        }
    }

    // ---- epilogue, written for CODE SIZE (a fully unrolled epilogue made this kernel 85-92 KB and every tile paid an
    // instruction-cache miss storm -- +20-30 us per GEMM): MI passes; in pass i every wave parks its RAW acc[i][*] (32 rows
    // x WTN cols) in LDS as 16-B writes; then a rolled loop streams whole rows out: 16-B LDS reads -> bias -> activation ->
    // residual -> convert -> one 16-B global store per thread and row.  Bias / activation / residual code exists once.
    constexpr int CPT = OUT_BF16 ? 8 : 4;                   // columns per thread in the store phase (16 B either way)
    constexpr int TPR = BN / CPT;                           // threads per output row
    constexpr int RPI = cfg::THREADS / TPR;                 // rows per store iteration
    constexpr int NIT = cfg::CROWS / RPI;                   // store iterations per pass
    const int ccol = (tid % TPR) * CPT, rsub = tid / TPR;
    const int ncol = n0 + ccol;
    float* sC = (float*)smem;
    void* const Cout = p.split_k ? (void*)((float*)p.C + (size_t)blockIdx.z * p.M * p.N) : p.C;
    float bias_r[CPT];
#pragma unroll
    for (int e = 0; e < CPT; ++e) bias_r[e] = (p.bias && ncol + e < p.N) ? p.bias[ncol + e] : 0.f;
    const float* __restrict__ res = p.residual;
    bool spatial = false;
    if constexpr (GATHER) spatial = (p.c_mode == 1);
    const bool vec_ok = (ncol + CPT <= p.N) && (spatial || (p.ldc % CPT) == 0);
    // LDS-only barrier: the global stores of a pass must NOT be drained at the pass barrier (a __syncthreads() would wait
    // vmcnt(0), i.e. a full HBM write round trip per pass); only this wave's LDS traffic has to be complete.
    auto lds_barrier = [&]() { wait_lgkmcnt<0>(); __builtin_amdgcn_s_barrier(); };
    auto row_of = [&](int i, int it) { const int lr = it * RPI + rsub; return m0 + (lr >> 5) * cfg::WTM + i * 32 + (lr & 31); };
    auto res_ptr = [&](int m) { return res + (size_t)(p.res_row_mod > 0 ? m % p.res_row_mod : m) * p.ldr + ncol; };
    auto stage_pass = [&](int i) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *(float4*)(sC + (wm * 32 + l31) * cfg::CLD_F32 + wn * cfg::WTN + j * 32 + 8 * q + 4 * hi) =
                    make_float4(acc[i][j][q * 4], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]);
    };
    auto load_raw = [&](int lr, float* v) {          // accumulator + bias
#pragma unroll
        for (int e = 0; e < CPT; e += 4) {
            const float4 f = *(const float4*)(sC + lr * cfg::CLD_F32 + ccol + e);
            v[e] = f.x + bias_r[e]; v[e + 1] = f.y + bias_r[e + 1]; v[e + 2] = f.z + bias_r[e + 2]; v[e + 3] = f.w + bias_r[e + 3];
        }
    };
    auto activate = [&](float* v) {
#pragma unroll
        for (int e = 0; e < CPT; e += 2) {
            if (ACT == 1) { const f32x2_t g = gelu_fast2(f32x2_t{v[e], v[e + 1]}); v[e] = g.x; v[e + 1] = g.y; }
            if (ACT == 2) { v[e] = fmaxf(v[e], 0.f); v[e + 1] = fmaxf(v[e + 1], 0.f); }
        }
    };
    const bool res_bf16 = p.epi_flags & 1, res_first = p.epi_flags & 2;   // ResNet-style epilogue: act(acc + bias + bf16 skip)
    auto store_vec = [&](size_t off, const float* v) {
        if constexpr (OUT_BF16)
            *(uint4*)((bf16_t*)Cout + off) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]),
                                                       pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
        else
            *(float4*)((float*)Cout + off) = make_float4(v[0], v[1], v[2], v[3]);
    };

    // fp32 output + epi_flags bit 8: C2 also receives the split-bf16 operand form of the stored values, [hi | lo | hi] along the channel axis (3 N
    // columns per row / pixel) -- what whmr_split3_bf16 makes of C in a pass of its own; the NEXT convolution of the bf16x3 path multiplies it
    // by [W_hi | W_hi | W_lo].  `off` = the element offset store_vec got (row stride ldc, or the scattered pixel offset): the copy sits at 3 off.
    const bool s3 = !OUT_BF16 && (p.epi_flags & 256);
    auto store_s3 = [&](size_t off, const float* v) {
        if constexpr (!OUT_BF16) {
            const size_t row = off - ncol;                                   // start of the row / pixel
            uint32_t h0, l0, h1, l1;
            split_bf16x2(v[0], v[1], h0, l0);
            split_bf16x2(v[2], v[3], h1, l1);
            if (p.epi_flags & 512) {                                         // two parts only: [hi | lo] (2 N per row / pixel)
                bf16_t* d = (bf16_t*)p.C2 + 2 * row + ncol;
                *(uint2*)d = make_uint2(h0, h1);
                *(uint2*)(d + p.N) = make_uint2(l0, l1);
            } else {
                bf16_t* d = (bf16_t*)p.C2 + 3 * row + ncol;
                *(uint2*)d = make_uint2(h0, h1);
                *(uint2*)(d + p.N) = make_uint2(l0, l1);
                *(uint2*)(d + 2 * p.N) = make_uint2(h0, h1);
            }
        }
    };

    if constexpr (!GATHER && MI <= 4) {
        if (res && n0 + BN <= p.N && (p.ldc % CPT) == 0 && (p.ldr % (res_bf16 ? CPT : 4)) == 0) {   // block-uniform: the two paths have different barrier sequences
            // ---- fast residual path (proj / fc2 / patch-embed; ResNet conv3 + bf16 skip): all residual rows of a pass are
            // prefetched one pass ahead, NIT independent 16-B loads in flight per thread instead of one exposed HBM round
            // trip per row.  rv holds raw bits: CPT fp32 values (CPT/4 x 16 B) or CPT bf16 values (first CPT*2 bytes).
            uint4 rv[NIT][CPT / 4];
            auto prefetch_res = [&](int i) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int m = row_of(i, it);
                    if (res_bf16) {
                        const bf16_t* rp = (const bf16_t*)p.residual + (size_t)(p.res_row_mod > 0 ? m % p.res_row_mod : m) * p.ldr + ncol;
                        if constexpr (CPT == 8) rv[it][0] = (m < m_end) ? *(const uint4*)rp : make_uint4(0, 0, 0, 0);
                        else { const uint2 r = (m < m_end) ? *(const uint2*)rp : make_uint2(0, 0); rv[it][0].x = r.x; rv[it][0].y = r.y; }
                    } else {
#pragma unroll
                        for (int e = 0; e < CPT / 4; ++e)
                            rv[it][e] = (m < m_end) ? *(const uint4*)(res_ptr(m) + 4 * e) : make_uint4(0, 0, 0, 0);
                    }
                }
            };
            prefetch_res(0);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                stage_pass(i);
                lds_barrier();
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int m = row_of(i, it);
                    float v[CPT];
                    load_raw(it * RPI + rsub, v);
                    if (!res_first) activate(v);
                    if (p.row_scale) {                                   // stochastic depth: per-row factor on the branch before the skip is added
                        const float rs = m < m_end ? p.row_scale[m] : 0.f;
#pragma unroll
                        for (int e = 0; e < CPT; ++e) v[e] *= rs;
                    }
                    if (res_bf16) {
                        const uint32_t w[4] = {rv[it][0].x, rv[it][0].y, rv[it][0].z, rv[it][0].w};
#pragma unroll
                        for (int e = 0; e < CPT / 2; ++e) { v[2 * e] += __uint_as_float(w[e] << 16); v[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u); }
                    } else {
#pragma unroll
                        for (int e = 0; e < CPT / 4; ++e) {
                            v[4 * e] += __uint_as_float(rv[it][e].x); v[4 * e + 1] += __uint_as_float(rv[it][e].y);
                            v[4 * e + 2] += __uint_as_float(rv[it][e].z); v[4 * e + 3] += __uint_as_float(rv[it][e].w);
                        }
                    }
                    if (res_first) activate(v);
                    if (m < m_end && (p.res_row_mod != -2003 || v[0] == 12345.678f)) {
                        store_vec((size_t)m * p.ldc + ncol, v);
                        if (s3) store_s3((size_t)m * p.ldc + ncol, v);
                    }
                }
                lds_barrier();                                   // slab consumed: the next pass may overwrite it
                if (i + 1 < MI) prefetch_res(i + 1);
            }
            return;
        }
    }
    const bool nostore = p.res_row_mod == -2003;
    const int ohw_ = spatial ? p.OH * p.OW : 1;
    const float rcp_ohw = 1.0f / (float)ohw_, rcp_ow = spatial ? 1.0f / (float)p.OW : 1.0f;
    // ---- generic path (no residual, conv scatter, N tail): rolled store loop, code exists once
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        stage_pass(i);
        lds_barrier();
#pragma unroll 1
        for (int it = 0; it < NIT; ++it) {
            const int m = row_of(i, it);
            if (m >= m_end || ncol >= p.N) continue;
            size_t crow;
            if (spatial) {
                // (b, oy, ox) of row m through reciprocal multiplies (M < 2^24: exact after one correction) -- the two integer
                // divisions per row were ~1000 instructions per thread and tile on the deconv scatter path
                int b = (int)((float)m * rcp_ohw);
                int rem = m - b * ohw_;
                if (rem >= ohw_) { ++b; rem -= ohw_; } else if (rem < 0) { --b; rem += ohw_; }
                int oy = (int)((float)rem * rcp_ow);
                int ox = rem - oy * p.OW;
                if (ox >= p.OW) { ++oy; ox -= p.OW; } else if (ox < 0) { --oy; ox += p.OW; }
                crow = (size_t)(c_off + b * p.osb + oy * p.osy + ox * p.osx);
            } else {
                crow = (size_t)m * p.ldc;
            }
            float v[CPT];
            load_raw(it * RPI + rsub, v);
            if (!res_first) activate(v);
            if (p.row_scale) {
                const float rs = p.row_scale[m];
#pragma unroll
                for (int e = 0; e < CPT; ++e) v[e] *= rs;
            }
            if (res) {                              // fp32 or bf16 skip tensor, added after (ViT) or before (ResNet) the activation
                const size_t roff = (size_t)(p.res_row_mod > 0 ? m % p.res_row_mod : m) * p.ldr + ncol;
                if (res_bf16 && vec_ok && (p.ldr % CPT) == 0) {          // one 16-B (8-B) load of CPT bf16 skip values
                    uint32_t w[CPT / 2];
                    if constexpr (CPT == 8) { const uint4 r = *(const uint4*)((const bf16_t*)p.residual + roff); w[0] = r.x; w[1] = r.y; w[2] = r.z; w[3] = r.w; }
                    else { const uint2 r = *(const uint2*)((const bf16_t*)p.residual + roff); w[0] = r.x; w[1] = r.y; }
#pragma unroll
                    for (int e = 0; e < CPT / 2; ++e) { v[2 * e] += __uint_as_float(w[e] << 16); v[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u); }
                } else {
#pragma unroll 1
                    for (int e = 0; e < CPT; ++e) {
                        if (ncol + e >= p.N) break;
                        v[e] += res_bf16 ? bf16_to_f32(((const bf16_t*)p.residual)[roff + e]) : res[roff + e];
                    }
                }
            }
            if (res_first) activate(v);
            if (nostore) {                           // timing probe: the whole epilogue except the global stores
                if (v[0] == 12345.678f) store_vec(crow + ncol, v);
            } else if (vec_ok) {
                store_vec(crow + ncol, v);
                if (s3) store_s3(crow + ncol, v);
            } else {
#pragma unroll 1
                for (int e = 0; e < CPT; ++e) {
                    if (ncol + e >= p.N) break;
                    if (OUT_BF16) ((bf16_t*)Cout)[crow + ncol + e] = f32_to_bf16(v[e]);
                    else ((float*)Cout)[crow + ncol + e] = v[e];
                }
            }
        }
        lds_barrier();
    }
}

template <int BM, int BN, int BK, int WM, int WN, int NS, int MINW, int PP, int OUT_BF16, int ACT, bool GATHER>
static int launch_big(const whmr_gemm& p, hipStream_t st) {
    using cfg = big_cfg<BM, BN, BK, WM, WN, NS>;
    auto kern = gemm_bf16_big_kernel<BM, BN, BK, WM, WN, NS, MINW, PP, OUT_BF16, ACT, GATHER>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, OUT_BF16 ? cfg::LDS16 + BN * 4 : cfg::LDS);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    constexpr int BME = BM - (PP == 3 ? 32 : 0);
    const int tiles = ((p.M + BME - 1) / BME) * ((p.N + BN - 1) / BN);
    hipLaunchKernelGGL(kern, dim3(tiles * (GATHER && p.n_phase > 1 ? p.n_phase : 1), 1, p.split_k ? (unsigned)((p.K + p.split_k - 1) / p.split_k) : 1), dim3(cfg::THREADS), OUT_BF16 ? cfg::LDS16 + BN * 4 : cfg::LDS, st, p);
    WHMR_CHECK_LAUNCH();
    return 0;
}

template <int BM, int BN, int BK, int WM, int WN, int NS, int MINW, int PP, bool GATHER>
static int launch_big_act(const whmr_gemm& p, hipStream_t st) {
    if (p.out_bf16) {
        switch (p.act) {
            case 0: return launch_big<BM, BN, BK, WM, WN, NS, MINW, PP, 1, 0, GATHER>(p, st);
            case 1: return launch_big<BM, BN, BK, WM, WN, NS, MINW, PP, 1, 1, GATHER>(p, st);
            case 2: return launch_big<BM, BN, BK, WM, WN, NS, MINW, PP, 1, 2, GATHER>(p, st);
        }
    } else {
        switch (p.act) {
            case 0: return launch_big<BM, BN, BK, WM, WN, NS, MINW, PP, 0, 0, GATHER>(p, st);
            case 1: return launch_big<BM, BN, BK, WM, WN, NS, MINW, PP, 0, 1, GATHER>(p, st);
            case 2: return launch_big<BM, BN, BK, WM, WN, NS, MINW, PP, 0, 2, GATHER>(p, st);
        }
    }
    return (int)hipErrorInvalidValue;
}

template <int BM, int BN, int BK, int WM, int WN, int NS, int MINW, int PP>
static int launch_big_mode(const whmr_gemm& p, hipStream_t st) {
    return p.a_mode == 1 ? launch_big_act<BM, BN, BK, WM, WN, NS, MINW, PP, true>(p, st)
                         : launch_big_act<BM, BN, BK, WM, WN, NS, MINW, PP, false>(p, st);
}

// tile: 128 -> 128x256x32, 4 waves, 2+ blocks/CU;  256 / 192 -> BM x 256 x 64, 8 waves, 1 block/CU.
extern "C" int whmr_gemm_bf16_big(const whmr_gemm* pp, int tile, void* stream) {
    const whmr_gemm& p = *pp;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || (p.K % 64)) return (int)hipErrorInvalidValue;
    if (p.a_mode == 1 && (p.Cin % 64 || !p.zeros)) return (int)hipErrorInvalidValue;
    if (p.a_mode == 0 && (p.lda % 8)) return (int)hipErrorInvalidValue;
    // the second output / the gelu' product exist on the packed bf16 epilogue only: bf16 output, row-major C with 16-B rows, no split-K, no row factors
    if (p.epi_flags & 256) {
        // split-bf16 copy of an fp32 output: fp32 C with dense rows (ldc == N when not scattered), N a multiple of 4 (8-byte pieces); the split-K route
        // hands the copy to splitk_epilogue_kernel (whmr_gemm_bf16_split clears the flag on the slice launches)
        if (!p.C2 || p.out_bf16 || (p.N & 3) || (p.c_mode == 0 && p.ldc != p.N) || p.split_k) return (int)hipErrorInvalidValue;
    } else
    if (p.C2 && !(p.act == 1 && p.out_bf16 && !p.residual && p.c_mode == 0 && !(p.N & 7) && !(p.ldc & 7) && !p.row_scale && !p.split_k && p.res_row_mod == 0))
        return (int)hipErrorInvalidValue;
    if ((p.epi_flags & 128) && !(p.residual && (p.epi_flags & 1) && !(p.epi_flags & 6) && p.act == 0 && p.out_bf16 && p.c_mode == 0 && !(p.N & 7) && !(p.ldc & 7) &&
                                 !(p.ldr & 7) && !p.row_scale && !p.split_k && p.res_row_mod == 0))
        return (int)hipErrorInvalidValue;
    // a skip tensor addressed like a scattered C exists on the packed bf16 epilogue only (bf16 in, bf16 out, no split-K, no row factors)
    if ((p.epi_flags & 4) && !(p.residual && (p.epi_flags & 1) && p.out_bf16 && p.c_mode == 1 && !(p.N & 7) && !(p.ldr & 7) && p.res_row_mod == 0 &&
                               !p.row_scale && !p.split_k && !(p.act == 1 && (p.epi_flags & 2))))
        return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    switch (tile) {
        case 65: return launch_big_mode<128, 64, 64, 2, 1, 2, 2, 0>(p, st);       // 48 KiB, 2 waves: narrow-N convs (N <= 64), 3 blocks / CU
        case 64: return launch_big_mode<128, 128, 64, 2, 2, 2, 2, 0>(p, st);      // 64 KiB, 4 waves (64x64 wave tiles): 2 blocks / CU
        case 128: return launch_big_mode<128, 256, 32, 1, 4, 3, 2, 0>(p, st);     // 72 KiB: 2 blocks / CU
        case 256: return launch_big_mode<256, 256, 32, 2, 4, 4, 2, 0>(p, st);     // 128 KiB: 1 block / CU, 3 steps ahead
        case 192: return launch_big_mode<192, 256, 64, 2, 4, 2, 2, 0>(p, st);     // 112 KiB, 2 stages
        case 257: return launch_big_mode<256, 256, 64, 2, 4, 2, 2, 0>(p, st);     // 128 KiB, 2 stages
        case 320: return launch_big_mode<320, 256, 64, 2, 4, 2, 2, 0>(p, st);
        case 160: return launch_big_mode<192, 256, 64, 2, 4, 2, 2, 3>(p, st);     // trimmed: 160 x 256 (wave rows of 96 + 64)
        case 224: return launch_big_mode<256, 256, 64, 2, 4, 2, 2, 3>(p, st);     // trimmed: 224 x 256 (wave rows of 128 + 96)
        case 258: return launch_big_mode<256, 256, 64, 2, 4, 2, 2, 4>(p, st);     // 256x256, DMA issue spread behind the MFMAs
        case 194: return launch_big_mode<192, 256, 64, 2, 4, 2, 2, 4>(p, st);     // 192x256, same
        case 322: return launch_big_mode<320, 256, 64, 2, 4, 2, 2, 4>(p, st);     // 320x256, same
        case 259: return launch_big_mode<256, 256, 64, 2, 4, 2, 2, 2>(p, st);     // same, 4 phases per K tile (half the barriers)     // 144 KiB, 2 stages, 160x64 wave tiles
    }
    return (int)hipErrorInvalidValue;
}
