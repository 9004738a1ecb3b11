// whmr_gemm_bf16: tile-shape chooser in front of the bf16 MFMA GEMM kernel (gemm_bf16_big.hip).
//
// Replaces the reference's nn.Linear / Conv2d / ConvTranspose2d call sites on the hot path:
//   vit.py:93,96,101,112 (qkv / proj), vit.py:66-75 (fc1 + GELU, fc2), vit.py:157-164 (patch embed, after im2col),
//   whmr.py:488-498 (deconv k4 s2 p1 as 4 sub-pixel 2x2 convs, BN folded, ReLU), whmr.py:419 (7x7 s3 conv).
//
// All candidate tiles run the same kernel template; what differs is how well the tile grid fills the 256 CUs.
// Cost model (matches the measured ordering on the ViT-B shapes, profiles/): a launch takes
//     (full rounds x blocks co-resident per CU + blocks per CU of the last partial round) x BM x BN x eff,
// a round = 256 CUs x blocks per CU tiles, because co-resident blocks share the CU's matrix pipes.  Examples at M = 12544: N = 768 -> 192x256 (198 tiles, one
// round of 3/4-size tiles); N = 2304 -> 256x256 (441 tiles, 2 rounds); N = 3072 -> 128x128 (2352 tiles, 4.6 -> 5 rounds).
#include "common.h"
#include "gemm_params.h"

extern "C" int whmr_gemm_bf16_big(const whmr_gemm* pp, int tile, void* stream);

// Deterministic split-K epilogue: sum the slices in a fixed order, then bias / skip / activation / convert (4 columns per thread).
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const whmr_gemm p, int splits) {
    const int n4 = p.N >> 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)p.M * n4) return;
    const int m = (int)(idx / n4), n = (int)(idx - (long)m * n4) * 4;
    const float* ws = (const float*)p.workspace + (size_t)m * p.N + n;
    float4 a = *(const float4*)ws;
    for (int s = 1; s < splits; ++s) {
        const float4 b = *(const float4*)(ws + (size_t)s * p.M * p.N);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    float v[4] = {a.x, a.y, a.z, a.w};
    if (p.bias) {
        const float4 b = *(const float4*)(p.bias + n);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
    float r[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.residual) {
        const size_t roff = (size_t)(p.res_row_mod > 0 ? m % p.res_row_mod : m) * p.ldr + n;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = (p.epi_flags & 1) ? bf16_to_f32(((const bf16_t*)p.residual)[roff + e]) : p.residual[roff + e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (p.epi_flags & 2) v[e] += r[e];
        if (p.act == 1) v[e] = gelu_fast(v[e]);
        else if (p.act == 2) v[e] = fmaxf(v[e], 0.f);
        if (p.row_scale) v[e] *= p.row_scale[m];
        if (!(p.epi_flags & 2)) v[e] += r[e];
    }
    const size_t off = (size_t)m * p.ldc + n;
    if (p.out_bf16) *(uint2*)((bf16_t*)p.C + off) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
    else *(float4*)((float*)p.C + off) = make_float4(v[0], v[1], v[2], v[3]);
    if (!p.out_bf16 && (p.epi_flags & 256)) {                       // split-bf16 copy [hi | lo | hi] of the fp32 row (gemm_params.h)
        uint32_t h0, l0, h1, l1;
        split_bf16x2(v[0], v[1], h0, l0);
        split_bf16x2(v[2], v[3], h1, l1);
        const int parts = (p.epi_flags & 512) ? 2 : 3;
        bf16_t* d = (bf16_t*)p.C2 + parts * (size_t)m * p.ldc + n;
        *(uint2*)d = make_uint2(h0, h1);
        *(uint2*)(d + p.N) = make_uint2(l0, l1);
        if (parts == 3) *(uint2*)(d + 2 * p.N) = make_uint2(h0, h1);
    }
}

struct tile_cfg { int id, bm, bn, per_cu, eff_pct; };

// Tuning switches for in-process A/B runs (tools/vit_timing.py): option 1 = use the ping-pong 256x256 main loop (tile 259)
// instead of the lock-step one (tile 257).  Not part of the data path contract; defaults are the shipped configuration.
// Ping-pong wins the isolated GEMM benchmark (qkv 59.5 -> 56 us, 4096-deep K +10 %) but loses inside the ViT forward
// (3.64 -> 3.69 ms per batch-64 step, interleaved A/B on one box), so it ships off.
static int g_opt_pingpong = 0;
static int g_opt_tile[4] = {0, 0, 0, 0};   // options 100..103: force a tile id for N = 2304 / (768, K <= 1024) / 3072 / (768, K > 1024); 0 = chooser
extern "C" int whmr_set_option(int key, int value) {
    if (key == 1) { g_opt_pingpong = value; return 0; }
    if (key >= 100 && key < 104) { g_opt_tile[key - 100] = value; return 0; }
    return (int)hipErrorInvalidValue;
}   // eff_pct: measured main-loop cost per tile area, relative to 256x256

// Explicit tile + split-K count (A/B tests; the chooser below calls it too).  Returns hipErrorInvalidValue when the shape cannot
// be split (no workspace, scatter / phase modes, N or leading dimensions not multiples of 4); `splits` is clamped to the workspace.
extern "C" int whmr_gemm_bf16_split(const whmr_gemm* pp, int tile, int splits_in, void* stream) {
    const whmr_gemm& p = *pp;
    if (!p.workspace || p.n_phase > 1 || p.c_mode != 0 || (p.N & 3) || (p.ldc & 3) || (p.ldr & 3) || (p.K % 64)) return (int)hipErrorInvalidValue;
    if ((p.C2 && !(p.epi_flags & 256)) || (p.epi_flags & (4 | 128))) return (int)hipErrorInvalidValue;      // second output / gelu' product / C-addressed skip: packed epilogue of the unsplit kernel only
    if ((p.epi_flags & 256) && (p.out_bf16 || p.ldc != p.N)) return (int)hipErrorInvalidValue;
    long splits = splits_in;
    while (splits > 1 && splits * p.M * p.N * 4 > p.workspace_bytes) --splits;
    if (splits <= 1) return (int)hipErrorInvalidValue;
    const long kps = ((p.K / 64 + splits - 1) / splits) * 64;
    splits = (p.K + kps - 1) / kps;
    whmr_gemm q = p;
    q.C = p.workspace; q.out_bf16 = 0; q.act = 0; q.bias = nullptr; q.residual = nullptr; q.row_scale = nullptr; q.epi_flags = p.epi_flags & 8; q.ldc = p.N;    // bit 3 (K order of the gather) belongs to the main loop
    q.C2 = nullptr;
    q.split_k = kps;
    const int rc = whmr_gemm_bf16_big(&q, tile, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)(((long)p.M * (p.N >> 2) + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p,
                       (int)splits);
    WHMR_CHECK_LAUNCH();
    return 0;
}

// Split-K WITHOUT the finishing pass: the raw fp32 partial sums of `splits` equal K slices go to C as [splits][M][N] planes (ldc = N) for a consumer that
// adds them itself (whmr_tz_fold: the composed Tz-head convolution, M = 22528, N = 128, K = 9216 -- 176 tiles of 128 x 128 leave a third of the CUs idle
// and each walks 144 K steps alone; two slices fill the chip).  No bias / activation / residual / second output; K must divide into `splits` slices of
// whole 64-element steps.
extern "C" int whmr_gemm_bf16_split_raw(const whmr_gemm* pp, int tile, int splits, void* stream) {
    const whmr_gemm& p = *pp;
    if (splits < 2 || p.K % (64 * splits) || p.n_phase > 1 || p.c_mode != 0 || p.out_bf16 || p.bias || p.residual || p.row_scale || p.C2 || p.act ||
        (p.epi_flags & ~8) || (p.N & 3))
        return (int)hipErrorInvalidValue;
    whmr_gemm q = p;
    q.ldc = p.N;
    q.split_k = p.K / splits;
    return whmr_gemm_bf16_big(&q, tile, stream);
}

extern "C" int whmr_gemm_bf16(const whmr_gemm* pp, int flags, void* stream) {
    const whmr_gemm& p = *pp;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || (p.K % 64)) return (int)hipErrorInvalidValue;
    if (p.n_phase > 1 && (p.n_phase != 4 || p.a_mode != 1 || p.c_mode != 1)) return (int)hipErrorInvalidValue;
    if (flags > 1) return whmr_gemm_bf16_big(pp, flags, stream);          // explicit tile id (A/B tests)
    {
        const int slot = p.N == 2304 ? 0 : p.N == 3072 ? 2 : p.N == 768 ? (p.K <= 1024 ? 1 : 3) : -1;
        if (slot >= 0 && g_opt_tile[slot] && p.a_mode == 0) return whmr_gemm_bf16_big(pp, g_opt_tile[slot], stream);
    }
    // 160: the 192-row kernel with the last 32-row block trimmed (gemm_bf16_big.hip, PP == 3): 12544 x 768 becomes ONE round of
    // 237 tiles instead of 198 (23 % of the CUs idle).  Interleaved in-model A/B (tools/trim_ab.py, ViT-B batch 64): fc2
    // (K = 3072) -35 us per forward, proj (K = 768) +-0; the 224-row form of the 256 kernel (qkv: 504 tiles in two rounds) wins
    // the isolated benchmark by 4 us per launch and LOSES 50 us per forward in the model (the chip is power-limited: filling
    // the idle CUs lowers everyone's clock), so it stays an explicit tile id only.  Deep-K shapes only.
    static const tile_cfg cands[] = {{320, 320, 256, 1, 100}, {259, 256, 256, 1, 94}, {192, 192, 256, 1, 100},
                                     {128, 128, 256, 2, 120}, {64, 128, 128, 2, 125}, {65, 128, 64, 3, 150},
                                     {160, 160, 256, 1, 103}};
    // Few tiles and a deep K (ResNet layer3/4 3x3 convs on a frame or two, fc2 of the ViT at batch 1): every block walks K
    // alone, one exposed L2/HBM round trip per K step on a mostly idle chip.  Slice K over blockIdx.z so that ~2 blocks per
    // CU are resident; partial sums go to the fp32 workspace and splitk_epilogue_kernel finishes (fixed order: deterministic).
    // Thresholds from tools/resnet_shapes.py: pays from K >= 2048 when the 128x128 grid covers less than half of the CUs.
    const long tiles64 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    // Weight-gradient products of the backward pass (dW = dY^T . X: a few dozen 256x256 tiles, K = all tokens): 256x256 tiles
    // sliced over K so that every CU owns exactly one block (tools/dw_probe.py: 3072x768x12544 124 -> 81 us).
    const long tiles256 = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
    if (p.workspace && p.n_phase <= 1 && p.a_mode == 0 && p.K >= 4096 && tiles256 >= 16 && tiles256 <= 128) {
        const long splits = 256 / tiles256;
        if (splits > 1 && whmr_gemm_bf16_split(pp, 257, (int)splits, stream) == 0) return 0;
    }
    if (p.workspace && p.n_phase <= 1 && p.K >= 2048 && tiles64 <= 128) {
        long splits = 512 / tiles64;
        if (splits > p.K / 512) splits = p.K / 512;
        if (splits > 1 && whmr_gemm_bf16_split(pp, 64, (int)splits, stream) == 0) return 0;
    }
    long best_cost = -1;
    const tile_cfg* best = &cands[4];
    for (const tile_cfg& c : cands) {
        if (c.id == 160 && (p.K < 2048 || p.a_mode != 0)) continue;
        const long tiles = (long)((p.M + c.bm - 1) / c.bm) * ((p.N + c.bn - 1) / c.bn) * (p.n_phase > 1 ? p.n_phase : 1);
        const long slots = 256L * c.per_cu;
        // full rounds run per_cu co-resident blocks per CU (they share the matrix pipes); the last, partial round only as
        // many per CU as it has blocks for
        const long full = tiles / slots, rem = tiles % slots;
        const long cost = (full * c.per_cu + (rem + 255) / 256) * c.bm * c.bn * c.eff_pct;
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = &c; }
    }
    if (best->id == 259 && !g_opt_pingpong) return whmr_gemm_bf16_big(pp, 257, stream);
    return whmr_gemm_bf16_big(pp, best->id, stream);
}
