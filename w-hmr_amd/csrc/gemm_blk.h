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
};
