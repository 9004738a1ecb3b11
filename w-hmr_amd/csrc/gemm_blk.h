// Descriptor of the blocked-layout bf16 GEMM (gemm_blk.hip).  Mirrors `struct whmr_gemm_blk_desc` in include/whmr_hip.h.
#pragma once
#include <stdint.h>

struct whmr_gemm_blk_desc {
    const void* A;        // bf16 blocked [ceil(M/32)][K/8][32][8]
    const void* W;        // bf16 blocked [N/32][K/8][32][8]      (nn.Linear weight [N, K], re-packed once)
    void* C;              // epi 0/1: bf16 blocked [ceil(M/32)][N/8][32][8];  epi 2/3: fp32 blocked [ceil(M/32)][N/4][32][4]
    const float* bias;    // [N] or null
    const float* res;     // epi 2: fp32 blocked like C (may alias C);  epi 3: row-major [res_rows, N], row = m % res_rows
    int32_t M, N, K;      // N % 256 == 0, K % 32 == 0; buffers hold whole 32-row blocks
    int32_t epi;          // 0: bf16(acc + bias)   1: bf16(gelu(acc + bias))   2 / 3: fp32(acc + bias + res)
    int32_t res_rows;
    int32_t tile;         // 0 = chooser; else (MI0 << 4) | MI1: 0x44 256 rows, 0x55 320, 0x43 224, 0x33 192, 0x32 160, 0x22 128, 0x54 288
    // ---- LayerNorm folding (vit.py:125,133: the LayerNorm in front of qkv / fc1 never runs as its own pass) ---------------------------
    // producer (epi 2 / 3, xhat != null): besides the fp32 stream C it writes xhat = bf16(C) in the blocked operand layout and, per row and
    //   256-column tile, the partial sums (sum x, sum x^2) of the values it stored: stats_out [rows][N/256][2] fp32 (N <= 1024).
    // consumer (epi 0 / 1, stats_in != null): A = xhat of the raw stream, W = bf16(gamma o W) packed, bias = b + W.beta, colsum[n] = sum_k W'[n,k];
    //   out = rstd[m] * (acc - mean[m] * colsum[n]) + bias[n], mean / rstd of row m from its K/256 partial pairs (fixed summation order).
    void* xhat;
    float* stats_out;
    const float* stats_in;
    const float* colsum;
    float ln_eps;
    // ---- split-bf16 operands ("bf16x3", gemm_blk_x3.hip): x = x_hi + x_lo, three MFMAs per product.  A_lo / W_lo: the lo halves of A / W (same
    // layout); C_lo: lo half of a bf16 result (epi 0 / 1; epi 1 then applies the erf GELU to fp32 accuracy, common.h gelu_as).  All null = plain bf16 operands.
    const void* A_lo;
    const void* W_lo;
    void* C_lo;
    // ---- per-row shift of a folding producer (xhat != null): xhat and stats_out are taken of (C - s_m) with
    //   s_m = (shift ? shift[m] : 0) + (shift_stats ? mean of row m from shift_stats [rows][N/256][2] : 0);   shift_out[m] = s_m when non-null.
    // The consumer formula is unchanged (LayerNorm is shift-invariant); shift_stats must not alias stats_out.
    const float* shift;
    const float* shift_stats;
    float* shift_out;
    void* xhat_lo;        // split-bf16 producer (A_lo set, xhat set): lo half of the centred-row operand pair
};
