// whmr_gemm_tn_bf16: C[Mo, No] (fp32) = A^T . B with BOTH operands stored reduction-major: A [K, lda >= Mo], B [K, ldb >= No] (bf16 rows of
// one reduction index).  This is the shape of every weight gradient of the training step (autograd of nn.Linear / Conv2d / ConvTranspose2d:
// vit.py:66-68,93,96,157; whmr.py:419-420,488-498 as run by loss.backward() in core/trainer.py:410-470):
//     dW[n_out, k_in] = sum_m dY[m, n_out] . X[m, k_in]
// with dY and X exactly as the forward / backward kernels leave them (token- or pixel-major) -- the NT kernels (gemm_bf16_big.hip) need both
// operands K-contiguous, i.e. two transposed copies per product (4.4 ms of a 31 ms step went into whmr_transpose_* / whmr_im2col_t).
//
// Tile: BM (64 | 128 | 256) x 256 outputs, 32 reduction rows per step, 8 waves (2 x 4, wave tile BM/2 x 64 on v_mfma_f32_32x32x16_bf16).  A step's
// operand rows go global -> LDS by LDS-DMA (global_load_lds, 16 B per lane, whole 512-B rows) into a 3-slot ring, two steps ahead, counted
// vmcnt, one barrier per step.  The MFMA fragments need, per lane, 8 consecutive REDUCTION indices of one output row / column -- a column of the
// LDS tile: ds_read_b64_tr_b16 delivers 4 of them per read (the 16 lanes of a group read a 4-row x 16-column block and get it transposed).
// Fragment k-slot e of half-wave hi holds tile row R0 + 4 hi + (e & 3) + 8 (e >> 2) for BOTH operands, so the products pair up correctly.
// Bank conflicts: the 4 rows one half-wave reads sit one row pitch (512 B = all 64 banks, twice) apart; the 16-B chunk index of row r is XORed
// with 4 (r & 3) (128-B rows: 4 ((r >> 1) & 1)), which spreads them over four distinct 64-B windows.  The swizzle is applied on the DMA's SOURCE side (LDS destinations of
// a wave instruction are linear).
//
// Split-K (a weight gradient is a few dozen tiles deep in K = all tokens): partial tiles go to the fp32 workspace and tn_reduce_kernel sums
// them in slice order (deterministic).  The grid is one-dimensional over (slice, tile) with the tile fastest, remapped so that every XCD owns
// a contiguous range of it: the tiles of ONE K slice read the same dY / X rows (every output tile needs them; the nine taps of a convolution
// weight gradient read the same pixels one shift apart), and the 32 CUs of an XCD walking ~1-4 slices side by side find them in their own
// L2 -- dispatched round-robin, the tiles of a slice sat on 8 different XCDs and every one of them fetched the rows again (4.5x the unique
// bytes for the 768 x 2304 gradient, 9x for the IUV head's).
#include <algorithm>
#include <cstdlib>
#include "common.h"

#ifndef TN_LAB
#define TN_LAB 0          // tools/lab/tn_lab.hip only (timing ablations, wrong results): 1 no MFMAs, 2 no fragment reads, 4 no LDS-DMA, 8 stamps, 16 L2-hot operands (four-wave body)
#endif
#ifndef TN_LAB_ROW64
#define TN_LAB_ROW64 0    // tools/r6_coresidency_probe.py only: 1 builds the 64-row tile of the gathering kernel back in (see whmr_conv_dw_tn_bf16)
#endif
#ifndef TN4_NS
#define TN4_NS 4          // ring slots of the four-wave body (32 KB each)
#endif
#ifndef TN_NS4
#define TN_NS4 3          // ring slots of the 256-row ping-pong tile; 4 (three steps ahead, 128 KB) measured 0.3 ms SLOWER per training step (22.05 vs 21.73 ms, one box)
#endif

typedef __attribute__((address_space(3))) void tn_lds_void_t;
typedef const __attribute__((address_space(1))) void tn_gbl_void_t;

template <int N> __device__ __forceinline__ void tn_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void tn_wait_lgkmcnt() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }

// 8 reduction rows (R0 + 4 hi + {0..3, 8..11}) of column c0 + (lane & 31) of a swizzled [32][CH * 8] bf16 tile (CH 16-B chunks per row): two
// transposing reads, ISSUED here and hidden from hipcc's waitcnt bookkeeping -- the caller waits once (lgkmcnt(0) + sched_barrier) for all
// fragments of a K step before the MFMAs.
// chunk-index XOR of tile row `row` (CH 16-B chunks per row).  The 4 rows a half-wave's transposing read touches must land in 4 distinct 64-B
// bank windows: rows of >= 256 B all start at bank 0 -> XOR 4 (row & 3); 128-B rows (BM = 64) alternate between the two halves of the banks
// already, rows r and r + 2 collide -> XOR 4 ((row >> 1) & 1).
template <int CH> __device__ __forceinline__ int tn_swz(int row) { return CH >= 16 ? 4 * (row & 3) : 4 * ((row >> 1) & 1); }

struct tn_raw { uint2 x, y; };
template <int CH>
__device__ __forceinline__ void tn_frag_issue(tn_raw& f, uint32_t tile, int R0, int c0, int lane) {
    const int g = lane >> 4, i = lane & 15;
    const int col = c0 + 16 * (g & 1) + 4 * (i & 3);
    const int row = R0 + 4 * (g >> 1) + (i >> 2);
    const uint32_t a0 = tile + row * (CH * 16) + ((((col >> 3) ^ tn_swz<CH>(row))) << 4) + (col & 7) * 2;
    // second read: row + 8 -- same (row & 3) and ((row >> 1) & 1): same swizzle
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3" : "=&v"(f.x), "=&v"(f.y) : "v"(a0), "n"(8 * CH * 16) : "memory");
}
__device__ __forceinline__ bf16x8_t tn_frag_value(const tn_raw& f) {
    union { bf16x8_t v; uint32_t u[4]; } o;
    o.u[0] = f.x.x; o.u[1] = f.x.y; o.u[2] = f.y.x; o.u[3] = f.y.y;
    return o.v;
}

struct tn_params {
    const bf16_t* A; const bf16_t* B; float* C;
    long lda, ldb, ldc;
    int Mo, No, K;
    int k_per_split;          // reduction rows per K slice (multiple of 32)
    int tiles, splits;        // grid = tiles * splits workgroups
    int round_robin;          // A/B only (WHMR_TN_RR=1): dispatch order (slice, tile) without the XCD remap
    float* ws;                // split-K partials [splits][Mo][No] (null: direct store), then the column-sum partials [splits][Mo]
    float* db;                // optional [Mo]: db[m] = sum_k A[k][m] (bias gradient of the same Linear: column sums of dY), or null
    // B gather (convolution weight gradients): reduction index k = (b, oy, ox) over an OH x OW grid, column n = (tap, c) with c < GC
    // (GC % 256 == 0, so a 256-column tile lies inside ONE tap): B[k][n] = img[b, oy * S + ky - P, ox * S + kx - P, c] (NHWC, pixel stride
    // ldb elements), zero outside the IH x IW image.  The rows of a tile then are 512 contiguous bytes of one source pixel -- or of `zeros`.
    int gather, OH, OW, IH, IW, GC, KW, S, P;
    int SX;                   // column stride of the gather (S is the row stride; whmr_conv_dw_tn_bf16 sets SX = S)
    const bf16_t* zeros;      // >= 512 B of zeros
};

// PP (round 4): the two wave rows are two GROUPS half a step apart (the blocked forward kernel's schedule, gemm_blk16_impl.h SCHED 1): per 32-row step a
// group runs MEM (all fragment reads of the step + its share of the DMA two steps ahead + the counted wait) and MFMA (16 MFMAs from registers), with
// ONE barrier per step; group 0 walks a step as [MFMA | MEM], group 1 as [MEM | MFMA], so on every SIMD one wave feeds the matrix pipe while its
// partner waits for LDS.  In the lock-step loop (PP = false) both waves of a SIMD issue their fragment reads at the same time and the pipe idles
// for the LDS round trip, twice per step.
// One (tile, K range) of a product: the body of both kernels below.  `out` (row stride ldo) receives the tile -- the result itself or a split-K partial --
// and `dbo` (when the product carries a bias gradient) the column sums of A over the same K range.
template <int MI, bool GATHER, bool PP>   // wave rows own MI 32-row blocks: BM = 64 MI
__device__ __forceinline__ void tn_body(const tn_params& p, char* smem, int tile, int k_begin, int k_end, float* __restrict__ out, long ldo,
                                        float* __restrict__ dbo) {
    constexpr int BM = 64 * MI, BN = 256, BK = 32;
    constexpr int CHA = BM / 8, CHB = BN / 8;                 // 16-B chunks per tile row
    constexpr int A_BYTES = BK * BM * 2, B_BYTES = BK * BN * 2, SLOT = A_BYTES + B_BYTES;
    constexpr int NS = (PP && MI == 4) ? TN_NS4 : 3;          // ring slots: the ping-pong loop of the 256-row tile may run 3 steps ahead (4 x 32 KB)
    constexpr int UNITS = SLOT / 1024, UPW = (UNITS + 7) / 8; // 1-KiB DMA units per step; per wave 4 (BM 256), 3 (BM 128), 3 or 2 (BM 64: 20 units)
    constexpr int REM = UNITS % 8;                            // waves < REM issue UPW units, the others UPW - 1 (REM == 0: all UPW)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5;
    const int tiles_n = p.No / BN;
    const int tm = tile / tiles_n, tn = tile % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int nkt = (k_end - k_begin) / BK;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(tn_lds_void_t*)smem;
    const bool dma_full = (REM == 0) || (wave < REM);

    // DMA unit u = wave + 8 i: 1 KiB of the slot, linear.  A units first (A_BYTES / 1024), then B.  Inside an operand the unit covers
    // 1024 / (row bytes) rows; lane L -> linear byte L * 16 of the unit -> (row, chunk position) -> source chunk = position ^ 4 (row & 3).
    const bf16_t* usrc[UPW];
    int urow[UPW], uchunk[UPW];
#pragma unroll
    for (int i = 0; i < UPW; ++i) {
        int u = wave + 8 * i;
        if (u >= UNITS) u = UNITS - 1;                                           // never issued (dma_full is false); keeps the address valid
        const bool isA = u < A_BYTES / 1024;
        const int off = (isA ? u : u - A_BYTES / 1024) * 1024 + lane * 16;       // byte offset inside the operand tile
        const int rowb = isA ? BM * 2 : BN * 2;
        const int row = off / rowb, pos = (off % rowb) >> 4;
        const int chunk = pos ^ (isA ? tn_swz<CHA>(row) : tn_swz<CHB>(row));
        urow[i] = row; uchunk[i] = chunk;
        usrc[i] = isA ? p.A + (size_t)(k_begin + row) * p.lda + m0 + chunk * 8
                      : p.B + (size_t)(k_begin + row) * p.ldb + n0 + chunk * 8;
    }
    const size_t stepA = (size_t)BK * p.lda, stepB = (size_t)BK * p.ldb;
    // gather: this tile's tap and channel offset
    const int g_tap = GATHER ? n0 / p.GC : 0, g_c0 = GATHER ? n0 - g_tap * p.GC : 0;
    const int g_ky = GATHER ? g_tap / p.KW : 0, g_kx = GATHER ? g_tap - g_ky * p.KW : 0;
    // gather: grid position (image, oy, ox) of each piece's reduction row, carried from step to step (stage() is called for kt = 0, 1, 2, ...
    // in order; a step advances every row by 32 positions) -- two integer divisions per piece and step cost as much issue time as the
    // step's MFMAs on the 128-row tile
    int gb[UPW], goy[UPW], gox[UPW];
    if constexpr (GATHER) {
#pragma unroll
        for (int i = 0; i < UPW; ++i) {
            const int k = k_begin + urow[i];
            const int ohw = p.OH * p.OW;
            gb[i] = k / ohw;
            const int rem = k - gb[i] * ohw;
            goy[i] = rem / p.OW;
            gox[i] = rem - goy[i] * p.OW;
        }
    }
    auto stage = [&](int kt) {
        const int slot = kt % NS;
#pragma unroll
        for (int i = 0; i < UPW; ++i) {
            if (i == UPW - 1 && !dma_full) continue;
            const int u = wave + 8 * i;
            const bool isA = u < A_BYTES / 1024;
            const bf16_t* src;
            if (GATHER && !isA) {
                const int iy = goy[i] * p.S + g_ky - p.P, ix = gox[i] * p.SX + g_kx - p.P;
                const bool in = (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
                src = in ? p.B + ((size_t)(gb[i] * p.IH + iy) * p.IW + ix) * p.ldb + g_c0 + uchunk[i] * 8 : p.zeros + uchunk[i] * 8;
                gox[i] += BK;
                while (gox[i] >= p.OW) {
                    gox[i] -= p.OW;
                    if (++goy[i] == p.OH) { goy[i] = 0; ++gb[i]; }
                }
            } else {
                src = usrc[i] + (size_t)kt * (isA ? stepA : stepB);
            }
            if (!(TN_LAB & 4)) __builtin_amdgcn_global_load_lds((tn_gbl_void_t*)src, (tn_lds_void_t*)(smem + slot * SLOT + u * 1024), 16, 0, 0);
        }
    };
    f32x16_t acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // bias gradient rides along: the waves of the first column tile that own wave column 0 add up their A fragments (8 reduction rows of one
    // output row per lane and fragment) on the VALU, beside the matrix pipe
    const bool do_db = p.db != nullptr && tn == 0 && wn == 0;
    float dbs[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) dbs[i] = 0.f;

    if (nkt > 0) stage(0);
    if (nkt > 1) stage(1);
    if constexpr (NS == 4) { if (nkt > 2) stage(2); }
    if constexpr (PP) {
        tn_raw qa[2][MI], qb[2][2];                                             // the fragments of one step (both 16-row halves)
        unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};                    // TN_LAB & 8: s_memtime stamps of step 60
        bool rec = false;
#define TN_STAMP(i) do { if ((TN_LAB & 8) && rec) st[i] = __builtin_amdgcn_s_memtime(); } while (0)
        auto MEM = [&](int x) {
            const uint32_t ta = lds0 + (x % NS) * SLOT, tb = ta + A_BYTES;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (TN_LAB & 2) continue;
#pragma unroll
                for (int j = 0; j < 2; ++j) tn_frag_issue<CHB>(qb[ks][j], tb, ks * 16, wn * 64 + j * 32, lane);
#pragma unroll
                for (int i = 0; i < MI; ++i) tn_frag_issue<CHA>(qa[ks][i], ta, ks * 16, wm * (32 * MI) + i * 32, lane);
            }
            TN_STAMP(2);
            // slot (x + NS - 1) % NS held step x - 1: both groups read it in MEM(x - 1), one barrier ago at the latest
            if (x + NS - 1 < nkt) stage(x + NS - 1);
            TN_STAMP(3);
            // own share of step x + 1 has landed; the younger steps (x + 2 .. x + NS - 1, as far as they exist) may fly
            const int young = (nkt - 2 - x) < (NS - 2) ? (nkt - 2 - x) : (NS - 2);
            if (young >= 2) { if (dma_full) tn_wait_vmcnt<2 * UPW>(); else tn_wait_vmcnt<2 * (UPW - 1)>(); }
            else if (young == 1) { if (dma_full) tn_wait_vmcnt<UPW>(); else tn_wait_vmcnt<UPW - 1>(); }
            else tn_wait_vmcnt<0>();
            TN_STAMP(4);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            TN_STAMP(5);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto MFMA = [&]() {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8_t fb[2], fa[MI];
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[j] = tn_frag_value(qb[ks][j]);
#pragma unroll
                for (int i = 0; i < MI; ++i) fa[i] = tn_frag_value(qa[ks][i]);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        if (!(TN_LAB & 1)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
                if (do_db) {
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        const uint32_t w[4] = {qa[ks][i].x.x, qa[ks][i].x.y, qa[ks][i].y.x, qa[ks][i].y.y};
                        float t = 0.f;
#pragma unroll
                        for (int e = 0; e < 4; ++e) t += __uint_as_float(w[e] << 16) + __uint_as_float(w[e] & 0xffff0000u);
                        dbs[i] += t;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        if (nkt > 0) {
            {                                                                   // own share of step 0 (steps 1 .. NS - 2 may fly)
                const int young = (nkt - 1) < (NS - 2) ? (nkt - 1) : (NS - 2);
                if (young >= 2) { if (dma_full) tn_wait_vmcnt<2 * UPW>(); else tn_wait_vmcnt<2 * (UPW - 1)>(); }
                else if (young == 1) { if (dma_full) tn_wait_vmcnt<UPW>(); else tn_wait_vmcnt<UPW - 1>(); }
                else tn_wait_vmcnt<0>();
            }
            __builtin_amdgcn_s_barrier();
            MEM(0);
            if (wm == 1) MFMA();                                                // group 1 is half a step ahead
            for (int k = 0; k < nkt; ++k) {
                __builtin_amdgcn_s_barrier();
                if (TN_LAB & 8) { if (k == 61) { rec = true; TN_STAMP(7); } rec = k == 60; TN_STAMP(0); }
                if (wm == 0) MFMA();                                            // MFMA(k)
                if (wm == 0) TN_STAMP(1);
                if (k + 1 < nkt) {
                    MEM(k + 1);
                    if (wm == 1) MFMA();                                        // MFMA(k + 1)
                    if (wm == 1) TN_STAMP(6);
                }
            }
            if ((TN_LAB & 8) && lane == 0 && blockIdx.x == 8 && p.ws)           // lab: the stamps of one block, per wave, 96 MB into the workspace
                for (int i = 0; i < 8; ++i) ((unsigned long long*)(p.ws + (24u << 20)))[wave * 8 + i] = st[i];
        }
#undef TN_STAMP
    } else
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) { if (dma_full) tn_wait_vmcnt<UPW>(); else tn_wait_vmcnt<UPW - 1>(); }      // own share of step kt has landed (step kt + 1 may fly)
        else tn_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                                          // everyone's share landed; everyone is done with slot (kt - 1) % 3
        if (kt + 2 < nkt) stage(kt + 2);
        const uint32_t ta = lds0 + (kt % 3) * SLOT, tb = ta + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            tn_raw rb[2], ra[MI];
#pragma unroll
            for (int j = 0; j < 2; ++j) tn_frag_issue<CHB>(rb[j], tb, ks * 16, wn * 64 + j * 32, lane);
#pragma unroll
            for (int i = 0; i < MI; ++i) tn_frag_issue<CHA>(ra[i], ta, ks * 16, wm * (32 * MI) + i * 32, lane);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            bf16x8_t fb[2], fa[MI];
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = tn_frag_value(rb[j]);
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[i] = tn_frag_value(ra[i]);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            if (do_db) {
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const uint32_t w[4] = {ra[i].x.x, ra[i].x.y, ra[i].y.x, ra[i].y.y};
                    float t = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) t += __uint_as_float(w[e] << 16) + __uint_as_float(w[e] & 0xffff0000u);
                    dbs[i] += t;
                }
            }
        }
    }
    if (do_db) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const float v = dbs[i] + __shfl_xor(dbs[i], 32, 64);
            if (hi == 0) dbo[m0 + wm * (32 * MI) + i * 32 + l31] = v;
        }
    }
    // D[row = (r & 3) + 8 (r >> 2) + 4 hi][col = l31]: for a fixed r the 32 lanes of a half-wave write 128 contiguous bytes
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * (32 * MI) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                out[(size_t)m * ldo + n] = acc[i][j][r];
            }
        }
}

// ---- W4 (round 4): the 256 x 256 tile on FOUR waves, one per SIMD (512 registers each: 256 accumulators + two fragment sets) -------------------
// Stamps of the 8-wave ping-pong loop (tools/lab/tn_lab.hip): per 32-row step a wave spends 560 cycles in its 16 MFMAs and 1300 in MEM -- 24 transposing
// reads issue at ~19 cycles each per wave (460-580), the 4 LDS-DMA pieces at 42 cycles each beside the partner's reads but 130 beside the partner's
// MFMAs -- so a step lasts 560 + 1300 = 1860 cycles for 1024 cycles of matrix work per SIMD: the two waves of a SIMD do not hide each other's memory
// phase, they queue behind each other's issue.  One wave per SIMD has nobody to queue behind: its MFMA stream (32 per step, 32 cycles each) runs
// back to back and the step's 32 reads + 8 DMA pieces sit in the gaps between MFMAs (1.3 per gap; ~5 fit).  Wave (wm, wn) of 2 x 2 owns 128 x 128
// outputs: 8 fragments per 16-row half step (A 4, B 4) for 16 MFMAs -- 0.5 fragment per MFMA instead of 0.75.  Ring of 4 slots (128 KB), the
// DMA three steps ahead, ONE barrier per step (between the two half steps: the next step's first fragments are read under the second half's MFMAs).
// Tiles of 64 MI x 256 (MI = 4, 2, 1: wave (wm, wn) of 2 x 2 owns 32 MI x 128 outputs) and the gathering B operand of the convolution weight gradients
// run the same body; MI < 4 has fewer MFMAs per fragment (the B fragments are shared by fewer row blocks) and is bound by its reads / its LDS-DMA.
template <int OFF> __device__ __forceinline__ void tn4_read_at(uint2& d, uint32_t addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
// read q = 2 ks + r (half step ks, rows + 8 r) of a fragment in a tile of ROWB-byte rows: byte offset 8 q ROWB (a constant after unrolling)
template <int ROWB> __device__ __forceinline__ void tn4_read(uint2& d, uint32_t addr, const int q) {
    switch (q) {
        case 0: tn4_read_at<0>(d, addr); break;
        case 1: tn4_read_at<8 * ROWB>(d, addr); break;
        case 2: tn4_read_at<16 * ROWB>(d, addr); break;
        default: tn4_read_at<24 * ROWB>(d, addr); break;
    }
}
__device__ __forceinline__ void tn_wait_lgkmcnt_n(const int n) {                        // n: a constant after unrolling
    switch (n) {
        case 0: tn_wait_lgkmcnt<0>(); break;   case 1: tn_wait_lgkmcnt<1>(); break;   case 2: tn_wait_lgkmcnt<2>(); break;   case 3: tn_wait_lgkmcnt<3>(); break;
        case 4: tn_wait_lgkmcnt<4>(); break;   case 5: tn_wait_lgkmcnt<5>(); break;   case 6: tn_wait_lgkmcnt<6>(); break;   case 7: tn_wait_lgkmcnt<7>(); break;
        case 8: tn_wait_lgkmcnt<8>(); break;   case 9: tn_wait_lgkmcnt<9>(); break;   case 10: tn_wait_lgkmcnt<10>(); break; case 11: tn_wait_lgkmcnt<11>(); break;
        case 12: tn_wait_lgkmcnt<12>(); break; case 13: tn_wait_lgkmcnt<13>(); break; case 14: tn_wait_lgkmcnt<14>(); break; default: tn_wait_lgkmcnt<15>(); break;
    }
}

template <int NS, int MI, bool GATHER>
__device__ __forceinline__ void tn4_body(const tn_params& p, char* smem, int tile, int k_begin, int k_end, float* __restrict__ out, long ldo,
                                         float* __restrict__ dbo) {
    constexpr int BM = 64 * MI, BN = 256, BK = 32, CHA = BM / 8, ROWA = BM * 2;
    constexpr int A_BYTES = BK * BM * 2, B_BYTES = BK * BN * 2, SLOT = A_BYTES + B_BYTES;
    constexpr int NA = A_BYTES / 1024, UPW = (NA + 16) / 4;                   // 1-KiB DMA units per step: NA = 4 MI of A, 16 of B; per wave MI + 4
    constexpr int PH0 = (UPW + 1) / 2, PH1 = UPW - PH0;                        // pieces issued under the first / second half step
    constexpr int NM = 4 * MI, NR = 2 * (MI + 4);                              // MFMAs and reads per half step
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int tiles_n = p.No / BN;
    const int tm = tile / tiles_n, tn = tile % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int nkt = (k_end - k_begin) / BK;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(tn_lds_void_t*)smem;

    // DMA unit u = wave + 4 i (i < UPW): i < MI -> A units (1 KiB = 1024 / ROWA rows), else B units (two 512-B rows).  Lane L -> byte L * 16 of the
    // unit -> (row, chunk position); source chunk = position ^ swizzle(row) (the swizzle sits on the source side).  Per-lane 32-bit byte offsets
    // from uniform bases; the gathering B operand carries (image, oy, ox) of its row from step to step instead (pieces are issued in step order).
    uint32_t voff[UPW];
    int gb[UPW], goy[UPW], gox[UPW];
#pragma unroll
    for (int i = 0; i < UPW; ++i) {
        const bool isA = i < MI;
        const int u = wave + 4 * i - (isA ? 0 : NA);
        const int off = u * 1024 + lane * 16;
        const int rowb = isA ? ROWA : 512;
        const int row = off / rowb, pos = (off % rowb) >> 4;
        const int chunk = pos ^ (isA ? tn_swz<CHA>(row) : tn_swz<32>(row));
        if (GATHER && !isA) {
            voff[i] = (uint32_t)(chunk << 4);
            const int k = k_begin + row, ohw = p.OH * p.OW;
            gb[i] = k / ohw;
            const int rem = k - gb[i] * ohw;
            goy[i] = rem / p.OW;
            gox[i] = rem - goy[i] * p.OW;
        } else {
            voff[i] = (uint32_t)(row * (isA ? p.lda : p.ldb) * 2 + (chunk << 4));
            gb[i] = goy[i] = gox[i] = 0;
        }
    }
    const char* baseA = (const char*)(p.A + (size_t)k_begin * p.lda + m0);
    const char* baseB = GATHER ? (const char*)p.B : (const char*)(p.B + (size_t)k_begin * p.ldb + n0);
    const size_t stepA = (size_t)BK * p.lda * 2, stepB = (size_t)BK * p.ldb * 2;
    const int g_tap = GATHER ? n0 / p.GC : 0, g_c0 = GATHER ? n0 - g_tap * p.GC : 0;
    const int g_ky = GATHER ? g_tap / p.KW : 0, g_kx = GATHER ? g_tap - g_ky * p.KW : 0;
    auto dma = [&](int kt, int i) {                                            // piece i of step kt; past the end: data nobody reads (the last step again / zeros)
        const int ks_ = (TN_LAB & 16) ? (kt & 3) : kt < nkt ? kt : nkt - 1;      // lab 16: every step re-reads the first four steps' rows (L2-hot operands)
        const char* src;
        if (GATHER && i >= MI) {
            const int iy = goy[i] * p.S + g_ky - p.P, ix = gox[i] * p.SX + g_kx - p.P;
            const bool in = kt < nkt && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
            src = in ? baseB + (((size_t)(gb[i] * p.IH + iy) * p.IW + ix) * p.ldb + g_c0) * 2 + voff[i] : (const char*)p.zeros + voff[i];
            gox[i] += BK;
            while (gox[i] >= p.OW) {
                gox[i] -= p.OW;
                if (++goy[i] == p.OH) { goy[i] = 0; ++gb[i]; }
            }
        } else {
            src = (i < MI ? baseA + (size_t)ks_ * stepA : baseB + (size_t)ks_ * stepB) + voff[i];
        }
        if (!(TN_LAB & 4)) __builtin_amdgcn_global_load_lds((tn_gbl_void_t*)src, (tn_lds_void_t*)(smem + (kt % NS) * SLOT + (wave + 4 * i) * 1024), 16, 0, 0);
    };
    // fragment read addresses inside slot 0: lane -> (row, column) of tn_frag_issue with R0 = 0; a half step adds 16 rows, the second read 8 rows
    uint32_t fa_off[MI], fb_off[4];
    {
        const int g = lane >> 4, i16 = lane & 15;
        const int row = 4 * (g >> 1) + (i16 >> 2);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int ca = wm * (32 * MI) + i * 32 + 16 * (g & 1) + 4 * (i16 & 3);
            fa_off[i] = lds0 + row * ROWA + ((((ca >> 3) ^ tn_swz<CHA>(row))) << 4) + (ca & 7) * 2;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cb = wn * 128 + j * 32 + 16 * (g & 1) + 4 * (i16 & 3);
            fb_off[j] = lds0 + A_BYTES + row * 512 + ((((cb >> 3) ^ tn_swz<32>(row))) << 4) + (cb & 7) * 2;
        }
    }
    f32x16_t acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // bias gradient: column sums of A.  The tiles_n column tiles of a row tile and both wave columns read the same A fragments: fragment block
    // b = MI wm + i (0 .. 2 MI - 1) of the row tile is summed by the one (column tile, wave column) with b % (2 tiles_n) == 2 tn + wn -- at most one
    // block per wave at tiles_n >= MI / 2 -- as four v_dot2c against bf16 ones, behind a scalar branch (inline asm: the compiler would turn the
    // branch into sixteen unconditional dot products and a select per half step on every wave: +27 % on the bare MFMA stream)
    bool db_on[MI];
    float dbs[MI];
    bool want_db = false;
#pragma unroll
    for (int i = 0; i < MI; ++i) { dbs[i] = 0.f; db_on[i] = p.db != nullptr && ((wm * MI + i) % (2 * tiles_n)) == 2 * tn + wn; want_db |= db_on[i]; }
    auto db_add = [&](float& s, const tn_raw& f) {
        asm volatile("v_dot2c_f32_bf16 %0, %1, %5\n\tv_dot2c_f32_bf16 %0, %2, %5\n\tv_dot2c_f32_bf16 %0, %3, %5\n\tv_dot2c_f32_bf16 %0, %4, %5"
                     : "+v"(s) : "v"(f.x.x), "v"(f.x.y), "v"(f.y.x), "v"(f.y.y), "v"(0x3f803f80u));
    };

    tn_raw qa[2][MI], qb[2][4];                                                // two fragment sets: the half step in the MFMAs, the next one in flight
    // The NR = 2 (MI + 4) reads of a half step are issued in the order its MFMAs (i-major) need them -- B0 A0 B1 B2 B3 A1 .. -- i.e. read position
    // q -> fragment q >> 1 of that order, read q & 1 -- and waited for by COUNT in front of the MFMA that first uses them: what may still be in
    // flight there = the later reads of that half step + the reads the new half step has issued so far (MI = 4: lgkmcnt 12 / 11 / 10 / 9 / 8 for
    // MFMAs 0-4, 10 for MFMA 8, 12 for MFMA 12), so no half step ends on an LDS round trip.  Nothing else in the loop touches lgkmcnt (no scalar
    // loads: checked in the ISA).
#define TN4_READ(dst, sb, ks, q)                                                                                             \
    do {                                                                                                                     \
        if (!(TN_LAB & 2)) {                                                                                                 \
            const int fo_ = (q) >> 1, r_ = (q) & 1;                      /* order: B0 A0 B1 B2 B3 A1 A2 A3 */                \
            if (fo_ == 1) tn4_read<ROWA>(r_ ? qa[dst][0].y : qa[dst][0].x, fa_off[0] + (sb), 2 * (ks) + r_);                 \
            else if (fo_ >= 5) tn4_read<ROWA>(r_ ? qa[dst][(fo_ - 4) % MI].y : qa[dst][(fo_ - 4) % MI].x, fa_off[(fo_ - 4) % MI] + (sb), 2 * (ks) + r_); \
            else { const int j_ = fo_ == 0 ? 0 : fo_ - 1;                                                                    \
                   tn4_read<512>(r_ ? qb[dst][j_].y : qb[dst][j_].x, fb_off[j_] + (sb), 2 * (ks) + r_); }                    \
        }                                                                                                                    \
    } while (0)
    // one half step: the NM = 4 MI MFMAs of fragment set `cur`, and in their gaps the NR reads of the next half step (set cur ^ 1; slot base nsb, half
    // nks) + `np` of the DMA pieces of step dkt (pieces dp0 ..) + the bias sums of set `cur`.  Reads and pieces are unconditional: past the last
    // step they fetch data nobody uses -- no branch in the stream, one vmcnt count for every step.
#define TN4_HALF(cur, nsb, nks, dkt, dp0, np)                                                                                \
    do {                                                                                                                     \
        _Pragma("unroll") for (int i = 0; i < MI; ++i) {                                                                     \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                  \
                const int g_ = i * 4 + j;                                                                                    \
                /* last read position this MFMA needs (fragment order above), if it is the first user of that fragment */   \
                const int need_ = i == 0 ? (j == 0 ? 3 : 2 * (j + 1) + 1) : (j == 0 ? 2 * (4 + i) + 1 : -1);                  \
                if (need_ >= 0) tn_wait_lgkmcnt_n((NR - 1 - need_) + (g_ * NR) / NM);                                        \
                __builtin_amdgcn_sched_barrier(0);                                                                           \
                if (!(TN_LAB & 1)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tn_frag_value(qa[cur][i]), tn_frag_value(qb[cur][j]), acc[i][j], 0, 0, 0); \
                __builtin_amdgcn_sched_barrier(0);                                                                           \
                _Pragma("unroll") for (int q_ = (g_ * NR) / NM; q_ < ((g_ + 1) * NR) / NM; ++q_) TN4_READ((cur) ^ 1, nsb, nks, q_);            \
                _Pragma("unroll") for (int t_ = 0; t_ < (np); ++t_)                                                          \
                    if ((t_ * NM + NM / 2) / (np) == g_) dma(dkt, (dp0) + t_);                                               \
                if (j == 3 && db_on[i]) db_add(dbs[i], qa[cur][i]);                                                          \
                __builtin_amdgcn_sched_barrier(0);                                                                           \
            }                                                                                                                \
        }                                                                                                                    \
    } while (0)

    // prologue: steps 0 .. NS - 2 in flight, step 0 landed, its first half step in registers
    if (nkt > 0) {
#pragma unroll
        for (int s0 = 0; s0 < NS - 1; ++s0)
#pragma unroll
            for (int i = 0; i < UPW; ++i) dma(s0, i);
        tn_wait_vmcnt<(NS - 2) * UPW>();
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int q = 0; q < NR; ++q) TN4_READ(0, 0u, 0, q);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        for (int x = 0; x < nkt; ++x) {
            const uint32_t sb = (uint32_t)((x % NS) * SLOT), sb1 = (uint32_t)(((x + 1) % NS) * SLOT);
            // first half step; reads of the second half (same slot); the first PH0 pieces of step x + NS - 1 (its slot held step x - 1: everybody passed
            // the barrier of step x - 1 after their last read of it)
            TN4_HALF(0, sb, 1, x + NS - 1, 0, PH0);
            // own pieces of step x + 1 have landed: younger = steps x + 2 .. x + NS - 2 (UPW each) + the PH0 pieces just issued
            tn_wait_vmcnt<(NS - 3) * UPW + PH0>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // second half step; reads of step x + 1's first half; the other pieces
            TN4_HALF(1, sb1, 0, x + NS - 1, PH0, PH1);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        tn_wait_vmcnt<0>();                                                      // the surplus pieces land in this workgroup's LDS: not after it has gone
    }
#undef TN4_HALF
#undef TN4_READ
    if (want_db) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const float v = dbs[i] + __shfl_xor(dbs[i], 32, 64);
            if (db_on[i] && hi == 0) dbo[m0 + wm * (32 * MI) + i * 32 + l31] = v;
        }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 128 + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * (32 * MI) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                out[(size_t)m * ldo + n] = acc[i][j][r];
            }
        }
}

template <int MI, bool GATHER, bool PP>
__global__ __launch_bounds__(512, 2) void gemm_tn_kernel(const tn_params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lid = p.round_robin ? (int)blockIdx.x : xcd_remap(blockIdx.x, p.tiles * p.splits);
    const int slice = lid / p.tiles, tile = lid - slice * p.tiles;
    const int k_begin = slice * p.k_per_split;
    const int k_end = min(p.K, k_begin + p.k_per_split);
    tn_body<MI, GATHER, PP>(p, smem, tile, k_begin, k_end, p.ws ? p.ws + (size_t)slice * p.Mo * p.No : p.C, p.ws ? (long)p.No : p.ldc,
                            p.ws ? p.ws + (size_t)p.splits * p.Mo * p.No + (size_t)slice * p.Mo : p.db);
}

// Grouped launch (round 4): the weight gradients of ONE transformer layer (qkv, proj, fc1, fc2: 27 + 9 + 36 + 36 tiles of 256 x 256 at D = 768, all over
// the same K = tokens) in one grid.  Launched one by one each product slices K 7-28 ways to own the chip (~250 blocks), i.e. every launch writes and
// re-reads ~65 MB of fp32 partial tiles (a third of its time: 12 us of stores behind a 47 us main loop + a 13 us reduce); together 108 tiles fill the
// chip with TWO slices -- a quarter of the partial traffic, one prologue / epilogue per 192 steps instead of four per 162.  Block order = (product,
// slice, tile) with the XCD remap of the single launch: the tiles of one slice of one product sit on one XCD (qkv: 27 = 216 / 8) and share its L2.
#define TN_GROUP_MAX 4
struct whmr_tn_item {             // include/whmr_hip.h
    const void* A; long lda;
    const void* B; long ldb;
    float* C; long ldc;
    float* db;
    int Mo, No;
};
struct tn_group {
    tn_params it[TN_GROUP_MAX];
    int first[TN_GROUP_MAX];      // first logical block of every product
    int n, total;
};

template <typename F> __device__ __forceinline__ auto tn_pick(const tn_group& g, int i, F f) {
    auto v = f(g.it[0]);
#pragma unroll
    for (int j = 1; j < TN_GROUP_MAX; ++j)
        if (i == j) v = f(g.it[j]);
    return v;
}

template <bool PP>
__global__ __launch_bounds__(512, 2) void gemm_tn_group_kernel(const tn_group g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lid = xcd_remap(blockIdx.x, g.total);
    int i = 0;
#pragma unroll
    for (int j = 1; j < TN_GROUP_MAX; ++j)
        if (j < g.n && lid >= g.first[j]) i = j;
    tn_params p{};                                                              // uniform: every field lives in SGPRs
    p.A = tn_pick(g, i, [](const tn_params& t) { return t.A; });
    p.B = tn_pick(g, i, [](const tn_params& t) { return t.B; });
    p.C = tn_pick(g, i, [](const tn_params& t) { return t.C; });
    p.ws = tn_pick(g, i, [](const tn_params& t) { return t.ws; });
    p.db = tn_pick(g, i, [](const tn_params& t) { return t.db; });
    p.lda = tn_pick(g, i, [](const tn_params& t) { return t.lda; });
    p.ldb = tn_pick(g, i, [](const tn_params& t) { return t.ldb; });
    p.ldc = tn_pick(g, i, [](const tn_params& t) { return t.ldc; });
    p.Mo = tn_pick(g, i, [](const tn_params& t) { return t.Mo; });
    p.No = tn_pick(g, i, [](const tn_params& t) { return t.No; });
    p.tiles = tn_pick(g, i, [](const tn_params& t) { return t.tiles; });
    p.K = g.it[0].K; p.k_per_split = g.it[0].k_per_split; p.splits = g.it[0].splits;
    int l = lid;
#pragma unroll
    for (int j = 1; j < TN_GROUP_MAX; ++j)
        if (i == j) l = lid - g.first[j];
    const int slice = l / p.tiles, tile = l - slice * p.tiles;
    const int k_begin = slice * p.k_per_split;
    const int k_end = min(p.K, k_begin + p.k_per_split);
    tn_body<4, false, PP>(p, smem, tile, k_begin, k_end, p.ws ? p.ws + (size_t)slice * p.Mo * p.No : p.C, p.ws ? (long)p.No : p.ldc,
                          p.ws ? p.ws + (size_t)p.splits * p.Mo * p.No + (size_t)slice * p.Mo : p.db);
}

__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ ws, int splits, int Mo, int No, float* __restrict__ C, long ldc,
                                                        float* __restrict__ db) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int n4 = No >> 2;
    if (db && idx < Mo) {                                   // column-sum partials sit behind the tile partials
        const float* wd = ws + (size_t)splits * Mo * No + idx;
        float a = wd[0];
        for (int s = 1; s < splits; ++s) a += wd[(size_t)s * Mo];
        db[idx] = a;
    }
    if (idx >= (long)Mo * n4) return;
    const int m = (int)(idx / n4), n = (int)(idx - (long)m * n4) * 4;
    const size_t stride = (size_t)Mo * No;
    const float* w = ws + (size_t)m * No + n;
    // slices are added in slice order (same bits whatever the batching); 8 loads are in flight at a time -- one dependent load per slice made
    // this pass latency-bound (13.5 us for the 9 slices of a 768 x 2304 gradient)
    float4 a = *(const float4*)w;
    for (int s0 = 1; s0 < splits; s0 += 8) {
        float4 b[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) b[j] = s0 + j < splits ? *(const float4*)(w + (size_t)(s0 + j) * stride) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (s0 + j < splits) { a.x += b[j].x; a.y += b[j].y; a.z += b[j].z; a.w += b[j].w; }
    }
    *(float4*)(C + (size_t)m * ldc + n) = a;
}

// single product on the four-wave body
template <int MI, bool GATHER>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_tn4_kernel(const tn_params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lid = p.round_robin ? (int)blockIdx.x : xcd_remap(blockIdx.x, p.tiles * p.splits);
    const int slice = lid / p.tiles, tile = lid - slice * p.tiles;
    const int k_begin = slice * p.k_per_split;
    const int k_end = min(p.K, k_begin + p.k_per_split);
    tn4_body<TN4_NS, MI, GATHER>(p, smem, tile, k_begin, k_end, p.ws ? p.ws + (size_t)slice * p.Mo * p.No : p.C, p.ws ? (long)p.No : p.ldc,
                                 p.ws ? p.ws + (size_t)p.splits * p.Mo * p.No + (size_t)slice * p.Mo : p.db);
}

// Which body runs which product is fixed (round 5; the A/B switches WHMR_TN_W4 / WHMR_TN_PP / WHMR_TN_RR of round 4 are gone with the variants that
// lost -- profiles/r04_tn_group_ab.txt, r04_tn_lab.txt keep their numbers): plain products on the four-wave body, gathering (convolution) products on
// the eight-wave ping-pong body (rocprof, training step: deconv dW 247 vs 222 us, Tz conv dW 390 vs 333 on four waves: the per-lane pixel
// bookkeeping of the gather sits in the MFMA stream), XCD-aware (slice, tile) order everywhere.
template <int MI, bool GATHER>
static int launch_tn4(tn_params p, int tiles, int splits, hipStream_t st) {
    constexpr int LDS = TN4_NS * (32 * 64 * MI * 2 + 32 * 256 * 2);
    auto kern = gemm_tn4_kernel<MI, GATHER>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    p.tiles = tiles; p.splits = splits; p.round_robin = 0;
    hipLaunchKernelGGL(kern, dim3(tiles * splits), dim3(256), LDS, st, p);
    WHMR_CHECK_LAUNCH();
    return 0;
}

template <int MI, bool GATHER, bool PP>
static int launch_tn_pp(tn_params p, int tiles, int splits, hipStream_t st) {
    constexpr int LDS = ((PP && MI == 4) ? TN_NS4 : 3) * (32 * 64 * MI * 2 + 32 * 256 * 2);
    auto kern = gemm_tn_kernel<MI, GATHER, PP>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    p.tiles = tiles; p.splits = splits; p.round_robin = 0;
    hipLaunchKernelGGL(kern, dim3(tiles * splits), dim3(512), LDS, st, p);
    WHMR_CHECK_LAUNCH();
    return 0;
}

template <int MI, bool GATHER>
static int launch_tn(tn_params p, int tiles, int splits, hipStream_t st) {
    return launch_tn_pp<MI, GATHER, true>(p, tiles, splits, st);
}

static int tn_run(tn_params p, int splits, void* workspace, long workspace_bytes, hipStream_t st) {
    const int MI = (p.Mo % 256 == 0) ? 4 : (p.Mo % 128 == 0) ? 2 : 1;
    const int tiles = (p.Mo / (64 * MI)) * (p.No / 256);
    const int steps = p.K / 32;
    if (splits <= 0) {
        splits = tiles >= 192 ? 1 : (256 + tiles / 2) / tiles;          // ~one tile per CU
        if (splits > steps / 8) splits = steps / 8 > 0 ? steps / 8 : 1;  // at least 8 steps per slice
    }
    if (splits > steps) splits = steps;
    while (splits > 1 && (!workspace || (long)splits * p.Mo * (p.No + 1) * 4 > workspace_bytes)) --splits;
    const int sps = (steps + splits - 1) / splits;                       // steps per slice
    splits = (steps + sps - 1) / sps;
    p.k_per_split = sps * 32;
    p.ws = splits > 1 ? (float*)workspace : nullptr;
    int rc;
    // (the gathering kernel has no 64-row instantiation: whmr_conv_dw_tn_bf16 takes Mo % 128 == 0 only, see there)
#if TN_LAB_ROW64
    if (p.gather && MI == 1) rc = launch_tn<1, true>(p, tiles, splits, st); else
#endif
    if (p.gather) rc = MI == 4 ? launch_tn<4, true>(p, tiles, splits, st) : MI == 2 ? launch_tn<2, true>(p, tiles, splits, st) : (int)hipErrorInvalidValue;
    else rc = MI == 4 ? launch_tn4<4, false>(p, tiles, splits, st) : MI == 2 ? launch_tn4<2, false>(p, tiles, splits, st) : launch_tn4<1, false>(p, tiles, splits, st);
    if (rc) return rc;
    if (splits > 1) {
        hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)(((long)p.Mo * (p.No >> 2) + 255) / 256)), dim3(256), 0, st, (const float*)workspace, splits,
                           p.Mo, p.No, p.C, p.ldc, p.db);
        WHMR_CHECK_LAUNCH();
    }
    return 0;
}

// A [K, lda], B [K, ldb] bf16 (16-B aligned rows: lda, ldb multiples of 8), C [Mo, ldc] fp32.  Mo % 64 == 0, No % 256 == 0, K % 32 == 0.
// db (nullable) [Mo] fp32 = column sums of A = the bias gradient when A is dY.
// workspace (fp32, workspace_bytes) holds the split-K partials; splits = 0 picks the slice count (about one tile per CU), splits = 1 needs
// no workspace.  Returns hipErrorInvalidValue for shapes outside the envelope (the caller keeps the transposed-copy path for those).
extern "C" int whmr_gemm_tn_bf16(const void* A, long lda, const void* B, long ldb, float* C, long ldc, float* db, int Mo, int No, int K, int splits,
                                 void* workspace, long workspace_bytes, void* stream) {
    if (Mo <= 0 || No <= 0 || K <= 0 || (Mo % 64) || (No % 256) || (K % 32) || (lda % 8) || (ldb % 8) || (ldc % 4) ||
        ((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((uintptr_t)C & 15) || lda < Mo || ldb < No || ldc < No)
        return (int)hipErrorInvalidValue;
    tn_params p{};
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.Mo = Mo; p.No = No; p.K = K; p.db = db;
    return tn_run(p, splits, workspace, workspace_bytes, (hipStream_t)stream);
}

// the grouped launch on the four-wave body
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_tn4_group_kernel(const tn_group g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lid = xcd_remap(blockIdx.x, g.total);
    int i = 0;
#pragma unroll
    for (int j = 1; j < TN_GROUP_MAX; ++j)
        if (j < g.n && lid >= g.first[j]) i = j;
    tn_params p{};
    p.A = tn_pick(g, i, [](const tn_params& t) { return t.A; });
    p.B = tn_pick(g, i, [](const tn_params& t) { return t.B; });
    p.C = tn_pick(g, i, [](const tn_params& t) { return t.C; });
    p.ws = tn_pick(g, i, [](const tn_params& t) { return t.ws; });
    p.db = tn_pick(g, i, [](const tn_params& t) { return t.db; });
    p.lda = tn_pick(g, i, [](const tn_params& t) { return t.lda; });
    p.ldb = tn_pick(g, i, [](const tn_params& t) { return t.ldb; });
    p.ldc = tn_pick(g, i, [](const tn_params& t) { return t.ldc; });
    p.Mo = tn_pick(g, i, [](const tn_params& t) { return t.Mo; });
    p.No = tn_pick(g, i, [](const tn_params& t) { return t.No; });
    p.tiles = tn_pick(g, i, [](const tn_params& t) { return t.tiles; });
    p.K = g.it[0].K; p.k_per_split = g.it[0].k_per_split; p.splits = g.it[0].splits;
    int l = lid;
#pragma unroll
    for (int j = 1; j < TN_GROUP_MAX; ++j)
        if (i == j) l = lid - g.first[j];
    const int slice = l / p.tiles, tile = l - slice * p.tiles;
    const int k_begin = slice * p.k_per_split;
    const int k_end = min(p.K, k_begin + p.k_per_split);
    tn4_body<TN4_NS, 4, false>(p, smem, tile, k_begin, k_end, p.ws ? p.ws + (size_t)slice * p.Mo * p.No : p.C, p.ws ? (long)p.No : p.ldc,
                p.ws ? p.ws + (size_t)p.splits * p.Mo * p.No + (size_t)slice * p.Mo : p.db);
}

// reduce of a grouped launch: blockIdx.y = product
__global__ __launch_bounds__(256) void tn_reduce_group_kernel(const tn_group g) {
    const int i = blockIdx.y;
    const float* ws = tn_pick(g, i, [](const tn_params& t) { return t.ws; });
    float* C = tn_pick(g, i, [](const tn_params& t) { return t.C; });
    float* db = tn_pick(g, i, [](const tn_params& t) { return t.db; });
    const long ldc = tn_pick(g, i, [](const tn_params& t) { return t.ldc; });
    const int Mo = tn_pick(g, i, [](const tn_params& t) { return t.Mo; }), No = tn_pick(g, i, [](const tn_params& t) { return t.No; });
    const int splits = g.it[0].splits;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int n4 = No >> 2;
    if (db && idx < Mo) {
        const float* wd = ws + (size_t)splits * Mo * No + idx;
        float a = wd[0];
        for (int s = 1; s < splits; ++s) a += wd[(size_t)s * Mo];
        db[idx] = a;
    }
    if (idx >= (long)Mo * n4) return;
    const int m = (int)(idx / n4), n = (int)(idx - (long)m * n4) * 4;
    const size_t stride = (size_t)Mo * No;
    const float* w = ws + (size_t)m * No + n;
    float4 a = *(const float4*)w;
    for (int s = 1; s < splits; ++s) {                                          // slice order, like tn_reduce_kernel (a grouped launch has 2-3 slices)
        const float4 b = *(const float4*)(w + (size_t)s * stride);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    *(float4*)(C + (size_t)m * ldc + n) = a;
}

// n_items (<= 4) products over the SAME reduction length K in one launch: the weight gradients of one transformer layer.  Mo % 256 == 0 here
// (the 256-row tile); otherwise the envelope of whmr_gemm_tn_bf16.  Results are deterministic (fixed slice order) but, with another slice count,
// not bit-identical to the single launches.
extern "C" int whmr_gemm_tn_bf16_group(const whmr_tn_item* items, int n_items, int K, void* workspace, long workspace_bytes, void* stream) {
    if (!items || n_items <= 0 || n_items > TN_GROUP_MAX || K <= 0 || (K % 32)) return (int)hipErrorInvalidValue;
    tn_group g{};
    int tiles = 0;
    for (int i = 0; i < n_items; ++i) {
        const whmr_tn_item& q = items[i];
        if (q.Mo <= 0 || q.No <= 0 || (q.Mo % 256) || (q.No % 256) || (q.lda % 8) || (q.ldb % 8) || (q.ldc % 4) || ((uintptr_t)q.A & 15) ||
            ((uintptr_t)q.B & 15) || ((uintptr_t)q.C & 15) || q.lda < q.Mo || q.ldb < q.No || q.ldc < q.No || !q.A || !q.B || !q.C)
            return (int)hipErrorInvalidValue;
        tn_params& p = g.it[i];
        p.A = (const bf16_t*)q.A; p.B = (const bf16_t*)q.B; p.C = q.C; p.lda = q.lda; p.ldb = q.ldb; p.ldc = q.ldc; p.Mo = q.Mo; p.No = q.No; p.K = K;
        p.db = q.db;
        p.tiles = (q.Mo / 256) * (q.No / 256);
        tiles += p.tiles;
    }
    const int steps = K / 32;
    int splits = tiles >= 192 ? 1 : 256 / tiles;                          // one round of blocks: never more blocks than CUs
    if (splits > steps / 8) splits = steps / 8 > 0 ? steps / 8 : 1;
    if (TN_LAB) { const char* e = getenv("TN_LAB_SPLITS"); if (e) splits = atoi(e); }
    long need = 0;
    for (int i = 0; i < n_items; ++i) need += (long)splits * g.it[i].Mo * (g.it[i].No + 1) * 4;
    if (splits > 1 && (!workspace || need > workspace_bytes)) splits = 1;
    const int sps = (steps + splits - 1) / splits;
    splits = (steps + sps - 1) / sps;
    float* w = (float*)workspace;
    int first = 0;
    for (int i = 0; i < n_items; ++i) {
        tn_params& p = g.it[i];
        p.k_per_split = sps * 32; p.splits = splits;
        p.ws = splits > 1 ? w : nullptr;
        w += (size_t)splits * p.Mo * (p.No + 1);
        g.first[i] = first;
        first += p.tiles * splits;
    }
    for (int i = n_items; i < TN_GROUP_MAX; ++i) { g.it[i] = g.it[0]; g.first[i] = first; }
    g.n = n_items; g.total = first;
    hipStream_t st = (hipStream_t)stream;
    constexpr int LDS4 = TN4_NS * (32 * 256 * 2 + 32 * 256 * 2);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_tn4_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS4);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL(gemm_tn4_group_kernel, dim3(g.total), dim3(256), LDS4, st, g);
    WHMR_CHECK_LAUNCH();
    if (splits > 1) {
        long most = 0;
        for (int i = 0; i < n_items; ++i) most = std::max(most, (long)g.it[i].Mo * (g.it[i].No >> 2));
        hipLaunchKernelGGL(tn_reduce_group_kernel, dim3((unsigned)((most + 255) / 256), n_items), dim3(256), 0, st, g);
        WHMR_CHECK_LAUNCH();
    }
    return 0;
}

// Convolution weight gradient without a column matrix: C [Mo, KH*KW*GC] = A^T . col(img), A [K = B*OH*OW, lda] bf16 (dY or X, one row per
// position of the OH x OW grid), img [B, IH, IW, (pixel stride ldp)] bf16 NHWC; column (tap = ky*KW + kx, c) of reduction row (b, oy, ox) is
// img[b, oy*S + ky - P, ox*S + kx - P, c] or 0 outside the image.  Covers the autograd of Conv2d (A = dY over the OUTPUT grid, img = X:
// dW[co, (ky,kx,ci)], whmr.py:419-420, iuv_predictor.py) and of ConvTranspose2d(k4, s2, p1) (A = X over the INPUT grid, img = dZ, S = 2, P = 1:
// dW[ci, (ky,kx,co)], whmr.py:488-498).  GC % 256 == 0, Mo % 128 == 0, K % 32 == 0; zeros: >= 512 B of zeros on the device.
static int conv_dw_tn_run(const void* A, long lda, const void* img, long ldp, float* C, long ldc, int Mo, int K, int nB, int OH, int OW,
                          int IH, int IW, int GC, int KH, int KW, int S, int SX, int P, const void* zeros, int splits, void* workspace,
                          long workspace_bytes, float* db, void* stream) {
    const long No = (long)KH * KW * GC;
    // Mo % 128, not 64: the 64-row instantiation of the gathering kernel (gemm_tn_kernel<1, true, true>) computed its own product correctly but, while it ran,
    // `v_pk_fma_f32 ... op_sel:[0,1,0]` returned wrong LOW lanes in kernels on OTHER streams (a canary of that one instruction: ~456 000 wrong lanes of 35 M
    // beside it, none beside any other kernel incl. the 128- / 256-row tiles and bare MFMA streams; smpl_skin_bwd_kernel, which had three such instructions,
    // differed in 88-112 of 96-120 launches; only while the tile issues MFMAs) -- round 6, tools/r6_coresidency_probe.py, profiles/r06_coresidency_probe.txt.
    // The tile is not built; callers widen dY to 128 columns (heads_autograd.TN_ROW_PAD).  (The VALU files are also built without packed fp32: build.py.)
    if (Mo <= 0 || K <= 0 || (Mo % (TN_LAB_ROW64 ? 64 : 128)) || (GC % 256) || GC <= 0 || (K % 32) || (lda % 8) || (ldp % 8) || (ldc % 4) || ((uintptr_t)A & 15) ||
        ((uintptr_t)img & 15) || ((uintptr_t)C & 15) || ((uintptr_t)zeros & 15) || !zeros || lda < Mo || ldp < GC || ldc < No ||
        (long)nB * OH * OW != K || No > (1L << 30))
        return (int)hipErrorInvalidValue;
    tn_params p{};
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)img; p.C = C; p.lda = lda; p.ldb = ldp; p.ldc = ldc; p.Mo = Mo; p.No = (int)No; p.K = K;
    p.db = db; p.gather = 1; p.OH = OH; p.OW = OW; p.IH = IH; p.IW = IW; p.GC = GC; p.KW = KW; p.S = S; p.SX = SX; p.P = P; p.zeros = (const bf16_t*)zeros;
    return tn_run(p, splits, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int whmr_conv_dw_tn_bf16(const void* A, long lda, const void* img, long ldp, float* C, long ldc, int Mo, int K, int nB, int OH, int OW,
                                    int IH, int IW, int GC, int KH, int KW, int S, int P, const void* zeros, int splits, void* workspace,
                                    long workspace_bytes, float* db, void* stream) {
    return conv_dw_tn_run(A, lda, img, ldp, C, ldc, Mo, K, nB, OH, OW, IH, IW, GC, KH, KW, S, S, P, zeros, splits, workspace, workspace_bytes, db, stream);
}

// the same with a row stride SY and a column stride SX of their own (the composed Tz convolution reads the map [B, H, W / 6, 6 C] with a 6 x 1 window at
// stride 6 x 1: heads_autograd.TzComposedFn)
extern "C" int whmr_conv_dw_tn2_bf16(const void* A, long lda, const void* img, long ldp, float* C, long ldc, int Mo, int K, int nB, int OH, int OW,
                                     int IH, int IW, int GC, int KH, int KW, int SY, int SX, int P, const void* zeros, int splits, void* workspace,
                                     long workspace_bytes, float* db, void* stream) {
    if (SY <= 0 || SX <= 0) return (int)hipErrorInvalidValue;
    return conv_dw_tn_run(A, lda, img, ldp, C, ldc, Mo, K, nB, OH, OW, IH, IW, GC, KH, KW, SY, SX, P, zeros, splits, workspace, workspace_bytes, db, stream);
}
