// whmr_gemm_bf16: tile-shape chooser in front of the bf16 MFMA GEMM kernel (gemm_bf16_big.hip).
//
// Replaces the reference's nn.Linear / Conv2d / ConvTranspose2d call sites on the hot path:
//   vit.py:93,96,101,112 (qkv / proj), vit.py:66-75 (fc1 + GELU, fc2), vit.py:157-164 (patch embed, after im2col),
//   whmr.py:488-498 (deconv k4 s2 p1 as 4 sub-pixel 2x2 convs, BN folded, ReLU), whmr.py:419 (7x7 s3 conv).
//
// All candidate tiles run the same kernel template; what differs is how well the tile grid fills the 256 CUs.
// Cost model (matches the measured ordering on the ViT-B shapes, profiles/): a launch takes
//     rounds x (blocks co-resident per CU) x BM x BN x eff,   rounds = ceil(tiles / (256 CUs x blocks per CU))
// because co-resident blocks share the CU's matrix pipes.  Examples at M = 12544: N = 768 -> 192x256 (198 tiles, one
// round of 3/4-size tiles); N = 2304 -> 256x256 (441 tiles, 2 rounds); N = 3072 -> 128x128 (2352 tiles, 4.6 -> 5 rounds).
#include "common.h"
#include "gemm_params.h"

extern "C" int whmr_gemm_bf16_big(const whmr_gemm* pp, int tile, void* stream);

struct tile_cfg { int id, bm, bn, per_cu, eff_pct; };   // eff_pct: measured main-loop cost per tile area, relative to 256x256

extern "C" int whmr_gemm_bf16(const whmr_gemm* pp, int flags, void* stream) {
    const whmr_gemm& p = *pp;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || (p.K % 64)) return (int)hipErrorInvalidValue;
    if (p.n_phase > 1 && (p.n_phase != 4 || p.a_mode != 1 || p.c_mode != 1)) return (int)hipErrorInvalidValue;
    if (flags > 1) return whmr_gemm_bf16_big(pp, flags, stream);          // explicit tile id (A/B tests)
    static const tile_cfg cands[] = {{320, 320, 256, 1, 100}, {257, 256, 256, 1, 100}, {192, 192, 256, 1, 100},
                                     {128, 128, 256, 2, 120}, {64, 128, 128, 2, 125}, {65, 128, 64, 3, 150}};
    long best_cost = -1;
    int best = 64;
    for (const tile_cfg& c : cands) {
        const long tiles = (long)((p.M + c.bm - 1) / c.bm) * ((p.N + c.bn - 1) / c.bn) * (p.n_phase > 1 ? p.n_phase : 1);
        const long slots = 256L * c.per_cu;
        const long rounds = (tiles + slots - 1) / slots;
        const long cost = rounds * c.per_cu * c.bm * c.bn * c.eff_pct;
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = c.id; }
    }
    return whmr_gemm_bf16_big(pp, best, stream);
}
