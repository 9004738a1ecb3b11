// whmr_gemm_blk_desc: bf16 MFMA GEMM on BLOCKED operand layouts, two wave groups in ping-pong.  The bf16 inference path of the ViT
// (vit.py:61-140: qkv / proj / fc1 + GELU / fc2, vit.py:157 patch embed) runs on this kernel; the row-major kernel of
// gemm_bf16_big.hip stays for the convolutions, the training graph and odd shapes.
//
// Layout ("blocked"): a [R, C] matrix is stored as [R/32][C/E][32][E] with E = 8 (bf16) or 4 (fp32), i.e. 512-byte units of
// 32 rows x 16 bytes.  That unit is at once
//   * what an MFMA operand fetch reads (lane = row, 8 consecutive k: 16 rows x 4 units for 16x16x32, 32 rows x 2 units for 32x32x16),
//   * what the (operand-swapped) MFMA result owns (lane = row, 4 fp32 / 8 bf16 consecutive columns),
//   * a contiguous 512 B of global memory AND of LDS.
// The bf16 kernel (gemm_blk16_impl.h) computes on v_mfma_f32_16x16x32_bf16, the split-bf16 kernel (gemm_blk_impl.h) on 32x32x16.
// Consequences: global_load_lds copies whole 1-KiB runs (no swizzle: the ds_read_b128 fragment reads of a linear image are
// conflict-free), and the epilogue stores straight from the accumulators in 1-KiB wave stores -- no LDS transpose, no barrier.
//
// Schedule (measured in tools/lab/gemm_lab.hip): 8 waves = 2 wave rows x 4 wave columns, wave tile (32 MI) x 64.  The two
// wave rows are two GROUPS (one wave of each per SIMD) that run one s_barrier apart.  Work unit = half a K tile (32 deep):
//   MEM(h)  : ds_read this wave's fragments of half tile h (2 MI + 4 reads), issue its share of the LDS-DMA of half tile h + 3
//             (ring of 4 half-tile slots, counted vmcnt: two younger groups stay in flight), wait
//   MFMA(h) : 8 MI MFMAs (16x16x32; 4 MI of 32x32x16 in the split-bf16 kernel) straight from registers
// with ONE s_barrier per half tile and the two groups walking each slot in opposite order (group 0: MFMA then MEM, group 1: MEM then
// MFMA): on every SIMD one wave feeds the matrix pipe while the other talks to LDS and the texture addresser.  In a lock-step loop all
// 8 waves issue their DMA at the same time and the matrix pipes idle for the ~1000 clk the L1 needs to take 64 KiB (qkv main loop
// 48 -> 36-38 us in the lab).
#include "gemm_blk16_impl.h"
// (the 32x32x16 kernel of gemm_blk_impl.h now only exists for split-bf16 operands: gemm_blk_x3.hip.  Its bf16 instantiation -- the A/B partner of
// round 3, WHMR_BLK_MFMA=32 -- the two-barrier schedule and the W-direct main loop are no longer compiled: DESIGN 6 keeps their numbers.)
__attribute__((visibility("hidden"))) int blk_x3_launch_tile(const whmr_gemm_blk_desc* pp, int tile, void* stream, int sched);

#ifdef WHMR_BLK_STAMPS
// lab build only (tools/gemm_stamps.py): every bf16 blocked launch takes the next [tiles][2][8] u64 record of a caller-provided device buffer
static unsigned long long* g_stamp_buf = nullptr;
static long g_stamp_cap = 0, g_stamp_used = 0;
static int g_stamp_launches = 0;
unsigned long long* blk_stamp_next(int tiles) {
    if (!g_stamp_buf || g_stamp_used + 1 + (long)tiles * 16 > g_stamp_cap) return nullptr;
    unsigned long long* q = g_stamp_buf + g_stamp_used;
    g_stamp_used += (long)tiles * 16;
    ++g_stamp_launches;
    return q;
}
extern "C" int whmr_debug_blk_stamps(void* buf, long capacity_u64) { g_stamp_buf = (unsigned long long*)buf; g_stamp_cap = capacity_u64; g_stamp_used = 0; g_stamp_launches = 0; return 0; }
extern "C" long whmr_debug_blk_stamps_used(void) { return g_stamp_used; }
#endif


// The chooser minimises (rounds over 256 CUs) x (tile rows).
static const int kBlkTiles[][2] = {{4, 4}, {5, 5}, {4, 3}, {3, 3}, {3, 2}, {2, 2}, {5, 4}, {2, 1}};

extern "C" int whmr_gemm_blk_tile(const whmr_gemm_blk_desc* pp, int tile, void* stream) {
    const whmr_gemm_blk_desc& p = *pp;
    if (p.M <= 0 || p.N <= 0 || (p.N % 256) || p.K < 32 || (p.K % 32) || p.epi < 0 || p.epi > 3) return (int)hipErrorInvalidValue;
    if ((p.epi >= 2) && !p.res) return (int)hipErrorInvalidValue;
    if (p.xhat && (p.epi < 2 || !p.stats_out)) return (int)hipErrorInvalidValue;
    if (p.stats_in && (p.epi >= 2 || !p.colsum || (p.K % 256) || p.K > 1024)) return (int)hipErrorInvalidValue;
    if (p.xhat && p.N > 1024) return (int)hipErrorInvalidValue;
    if ((p.shift || p.shift_stats || p.shift_out) && (!p.xhat || p.shift_stats == p.stats_out)) return (int)hipErrorInvalidValue;
    if (p.epi == 3 && p.res_rows <= 0) return (int)hipErrorInvalidValue;
    if (p.A_lo || p.W_lo || p.C_lo) {
        // split-bf16 operands: both lo halves, a lo output for the bf16 epilogues, a lo half of the folded-LayerNorm operand copy
        if (!p.A_lo || !p.W_lo || (p.epi < 2 && !p.C_lo) || (p.xhat && !p.xhat_lo)) return (int)hipErrorInvalidValue;
        return blk_x3_launch_tile(pp, tile, stream, 1);
    }
    return blk16_launch_tile(p, tile, (hipStream_t)stream);
}

static int g_chain_lab = 0;                                 // whmr_gemm_blk_set_tile(4, v): lab switch of the chain pilot (1 = workgroup-scope fences: WRONG across XCDs, timing only)
static int g_blk_force[4] = {0, 0, 0, 0};                   // whmr_set_option keys 110..113: tile for N = 2304 / (768, K <= 1024) / 3072 / (768, K > 1024)
extern "C" int whmr_gemm_blk_set_tile(int slot, int tile) {
    if (slot == 4) { g_chain_lab = tile; return 0; }
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

// fc1 (epi 1: bf16 + GELU, optionally with the folded LayerNorm) -> fc2 (epi 2: fp32 + residual, optionally emitting the next LayerNorm's operand copy and
// statistics) as ONE persistent launch (gemm_blk16_chain_kernel, gemm_blk16_impl.h) -- the cross-launch PILOT of round 6.  Accepts exactly the pair the
// ViT-B inference path issues on the tiles its chooser picks (fc1 320 x 256, fc2 160 x 256); anything else returns hipErrorInvalidValue and the caller
// issues the two launches.  counters: >= ceil(M / 320) uint32 (zeroed here); err: one int32 the kernel raises when a wait ran into its 20 ms limit.
extern "C" int whmr_gemm_blk_chain(const whmr_gemm_blk_desc* fc1, const whmr_gemm_blk_desc* fc2, void* counters, void* err, void* stream) {
#ifdef WHMR_BLK_STAMPS
    return (int)hipErrorInvalidValue;
#else
    const whmr_gemm_blk_desc &a = *fc1, &b = *fc2;
    if (!counters || !err || a.epi != 1 || b.epi != 2 || a.M != b.M || a.M <= 0 || a.C != b.A || a.N != b.K || (a.N % 256) || (b.N % 256) || (a.K % 32) ||
        (b.K % 32) || !b.res || a.A_lo || a.W_lo || a.C_lo || b.A_lo || b.W_lo || b.C_lo || a.xhat || (b.xhat && (!b.stats_out || b.N > 1024)) ||
        (a.stats_in && (!a.colsum || (a.K % 256) || a.K > 1024)) || b.stats_in)
        return (int)hipErrorInvalidValue;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) { cus = 0; return (int)hipErrorInvalidValue; }
    }
    return launch_blk16_chain<5, 5, 3, 2>(a, b, (unsigned*)counters, (int*)err, cus, g_chain_lab, (hipStream_t)stream);
#endif
}
