// GEMM / implicit-GEMM descriptor shared by the bf16 (MFMA 32x32x16) and fp32 (MFMA 32x32x2) kernels.
// Mirrors `struct whmr_gemm` in include/whmr_hip.h (plain C layout; filled by the host through ctypes).
#pragma once
#include <stdint.h>

struct whmr_gemm {
    const void* A;          // activations: plain [M, lda] or NHWC image for a_mode = 1
    const void* W;          // weights [N, K], K contiguous (nn.Linear layout / conv weight re-ordered (co, ky, kx, ci))
    void* C;                // output, fp32 or bf16 (out_bf16)
    const float* bias;      // [N] or null
    const float* residual;  // fp32, row stride ldr, or null; added after the activation
    const void* zeros;      // >= 256 B of zeros (gather padding source); required when a_mode = 1
    int32_t M, N, K;
    int32_t lda, ldc, ldr;
    int32_t res_row_mod;    // > 0: residual row = m % res_row_mod (pos-embed broadcast, vit.py:320)
    int32_t act;            // 0 none, 1 exact-erf GELU, 2 ReLU
    int32_t out_bf16;       // 1: C is bf16, 0: fp32
    int32_t a_mode;         // 0 plain, 1 conv gather: row m = (b, oy, ox), k = (ky, kx, ci)
    int32_t IH, IW, Cin, OH, OW, KW, SH, SW, PH, PW;   // iy = oy*SH + ky - PH, ix = ox*SW + kx - PW
    int32_t c_mode;         // 0 plain rows (m*ldc), 1 spatial scatter: c_off + b*osb + oy*osy + ox*osx
    int64_t c_off, osb, osy, osx;
    void* workspace;        // optional scratch for split-K partial sums (skinny / few-tile shapes); may be null
    int64_t workspace_bytes;
    // Sub-pixel phases of ConvTranspose2d(k4, s2, p1) in ONE launch (bf16 kernel, a_mode = c_mode = 1): n_phase = 4,
    // phase = blockIdx.y = 2*py + px:  W += phase*phase_w_stride;  PH -= py;  PW -= px;  c_off += py*phase_cy + px*phase_cx.
    int32_t n_phase;
    int32_t epi_flags;      /* bit 0: residual is bf16 (else fp32); bit 1: residual is added BEFORE the activation (ResNet blocks);
                             * bit 3 (bf16 gather): K is ordered (ci chunk of 64, ky, kx, ci in chunk) instead of (ky, kx, ci): all taps of one
                             * 64-channel slice are walked before the next slice, so the window overlap of a large-kernel conv on a map that
                             * exceeds the Infinity Cache is re-read from cache instead of HBM (Tz-head 7x7 s3 conv);
                             * bit 8 (fp32 C): C2 also receives the split-bf16 operand form of C, [hi | lo | hi] along the channel axis (3 N per row / pixel):
                             * the K-concatenated activation operand of the next convolution of the bf16x3 numerics, without a whmr_split3_bf16 pass;
                             * bit 9 (with bit 8): only [hi | lo] (2 N per row / pixel) -- the operand of a narrow convolution that takes the W_lo product
                             * as extra OUTPUT columns instead of a third K slice (the Tz head's 7x7 s3 convolution, N = 64: bound by its A bytes) */
    int64_t phase_w_stride, phase_cy, phase_cx;
    int64_t split_k;        /* internal (set by the bf16 launcher, pass 0): K elements per split-K slice, blockIdx.z = slice */
    const float* row_scale; /* optional [M]: act(acc + bias) is multiplied by row_scale[m] BEFORE the (post-activation) residual is added --
                             * stochastic depth of the training ViT (vit.py:132-139: x + drop_path(branch), per-sample mask / keep_prob) */
    void* C2;               /* optional second output of the bf16 kernel (act = GELU, bf16 C, no residual): C2 = bf16(acc + bias), the PRE-activation,
                             * next to C = gelu(that value) -- the training forward of fc1 keeps both (vit.py:66-68) without a separate GELU pass */
};
