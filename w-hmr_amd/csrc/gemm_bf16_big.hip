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
#define WHMR_BIG_BLOCK_ID blockIdx.x
#include "gemm_bf16_big_body.inc"
#undef WHMR_BIG_BLOCK_ID
}

// Up to 9 GEMMs of one tile shape as ONE launch: the residue-class data gradients of a strided convolution (heads_autograd.ConvNHWCFn: S*S stride-1
// implicit GEMMs of K = 4 .. 9 taps x 64 that scatter into interleaved pixels) are each under two rounds of tiles and half prologue / epilogue; as one
// grid their tiles fill the rounds together and the eight launch gaps go.  Block ranges start at multiples of 8 (whole XCD rounds: xcd_remap keeps
// its meaning inside a range); the padding blocks leave at once.
struct whmr_gemm_group {
    whmr_gemm p[9];
    int first[10];
    int n;
};

template <int BM, int BN, int BK, int WM, int WN, int NS, int MINW, int PP, int OUT_BF16, int ACT, bool GATHER>
__global__ __launch_bounds__(64 * WM * WN, MINW) void gemm_bf16_big_group_kernel(const whmr_gemm_group g) {
    int gi = 0;
    while (gi + 1 < g.n && (int)blockIdx.x >= g.first[gi + 1]) ++gi;
    const whmr_gemm p = g.p[gi];
    const int group_block = (int)blockIdx.x - g.first[gi];
    if (group_block >= ((p.M + (BM - (PP == 3 ? 32 : 0)) - 1) / (BM - (PP == 3 ? 32 : 0))) * ((p.N + BN - 1) / BN)) return;
#define WHMR_BIG_BLOCK_ID group_block
#include "gemm_bf16_big_body.inc"
#undef WHMR_BIG_BLOCK_ID
}

template <int BM, int BN, int BK, int WM, int WN, int NS, int MINW, int PP, int OUT_BF16, int ACT, bool GATHER>
static int launch_big_group(const whmr_gemm* ps, int n, hipStream_t st) {
    using cfg = big_cfg<BM, BN, BK, WM, WN, NS>;
    auto kern = gemm_bf16_big_group_kernel<BM, BN, BK, WM, WN, NS, MINW, PP, OUT_BF16, ACT, GATHER>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, OUT_BF16 ? cfg::LDS16 + BN * 4 : cfg::LDS);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    constexpr int BME = BM - (PP == 3 ? 32 : 0);
    whmr_gemm_group g;
    g.n = n;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        g.p[i] = ps[i];
        g.first[i] = total;
        total += (((ps[i].M + BME - 1) / BME) * ((ps[i].N + BN - 1) / BN) + 7) / 8 * 8;
    }
    for (int i = n; i < 10; ++i) g.first[i] = total;
    for (int i = n; i < 9; ++i) g.p[i] = ps[0];
    hipLaunchKernelGGL(kern, dim3(total), dim3(cfg::THREADS), OUT_BF16 ? cfg::LDS16 + BN * 4 : cfg::LDS, st, g);
    WHMR_CHECK_LAUNCH();
    return 0;
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

// n <= 9 gathering (a_mode 1) GEMMs, bf16 output, no activation, no split-K, no sub-pixel phases, as one launch of 192 x 256 x 64 tiles (tile 192) or
// 128 x 64 x 64 (tile 65, N <= 64); every descriptor is checked like a single launch (whmr_gemm_bf16_big).
extern "C" int whmr_gemm_bf16_group(const whmr_gemm* ps, int n, int tile, void* stream) {
    if (!ps || n < 1 || n > 9) return (int)hipErrorInvalidValue;
    for (int i = 0; i < n; ++i) {
        const whmr_gemm& p = ps[i];
        if (p.M <= 0 || p.N <= 0 || p.K <= 0 || (p.K % 64) || p.a_mode != 1 || (p.Cin % 64) || !p.zeros || !p.out_bf16 || p.act != 0 || p.split_k ||
            p.n_phase > 1 || p.C2 || (p.epi_flags & (128 | 256)) || p.row_scale)
            return (int)hipErrorInvalidValue;
        if ((p.epi_flags & 4) && !(p.residual && (p.epi_flags & 1) && p.c_mode == 1 && !(p.N & 7) && !(p.ldr & 7) && p.res_row_mod == 0)) return (int)hipErrorInvalidValue;
    }
    hipStream_t st = (hipStream_t)stream;
    switch (tile) {
        case 192: return launch_big_group<192, 256, 64, 2, 4, 2, 2, 0, 1, 0, true>(ps, n, st);
        case 65: return launch_big_group<128, 64, 64, 2, 1, 2, 2, 0, 1, 0, true>(ps, n, st);
    }
    return (int)hipErrorInvalidValue;
}
