/* C ABI of libwhmr_hip.so -- the MI355X (gfx950) kernels behind the W-HMR forward path.
 *
 * The reference (yw0208/W-HMR) is 100 % Python: its "FFI" for this path is the set of torch op call
 * sites inside models/whmr.py, models/maf_extractor.py, utils/geometry.py and the vendored ViT
 * (SURVEY 2.3).  Each entry point below names the call sites it replaces (paths relative to the
 * reference root).  Conventions:
 *   - every pointer is a DEVICE pointer unless stated otherwise; nothing is allocated or freed;
 *   - `stream` is a hipStream_t; calls only enqueue (no synchronisation) and are graph-capturable;
 *   - return value is a hipError_t cast to int (0 = success); nothing throws;
 *   - bf16 buffers hold raw bfloat16 bits (uint16_t); everything else is IEEE fp32 / int32 / int64.
 * The Python binding a reference maintainer would use is w-hmr_amd/_lib.py (ctypes); see INTEGRATION.md.
 */
#ifndef WHMR_HIP_H
#define WHMR_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* GEMM / implicit-GEMM descriptor:  C[M,N] = act(A[M,K] . W[N,K]^T + bias) + residual.
 * a_mode 1 gathers A rows from an NHWC image (row m = (b, oy, ox), k = (ky, kx, ci),
 * iy = oy*SH + ky - PH, ix = ox*SW + kx - PW, zero outside); c_mode 1 scatters row m to
 * c_off + b*osb + oy*osy + ox*osx (+ n): together they express Conv2d and the 4 sub-pixel phases of
 * ConvTranspose2d(k4, s2, p1). */
struct whmr_gemm {
    const void* A; const void* W; void* C;
    const float* bias; const float* residual; const void* zeros;
    int32_t M, N, K;
    int32_t lda, ldc, ldr;
    int32_t res_row_mod;
    int32_t act;            /* 0 none, 1 exact GELU, 2 ReLU */
    int32_t out_bf16;
    int32_t a_mode;
    int32_t IH, IW, Cin, OH, OW, KW, SH, SW, PH, PW;
    int32_t c_mode;
    int64_t c_off, osb, osy, osx;
};

/* bf16 MFMA GEMM (v_mfma_f32_32x32x16_bf16, fp32 accumulate).  Needs N % 128 == 0, K % 64 == 0.
 * Replaces nn.Linear at vit.py:93,96 (qkv, proj), vit.py:66-68 (fc1, fc2), Conv2d at vit.py:157 (after
 * whmr_patch_im2col), ConvTranspose2d+BN+ReLU at whmr.py:488-498 and Conv2d at whmr.py:419.
 * flags bit0: stage through registers instead of global_load_lds. */
int whmr_gemm_bf16(const struct whmr_gemm* p, int flags, void* stream);

/* exact-fp32 MFMA GEMM (v_mfma_f32_32x32x2_f32), any M/N/K.  Parity mode of the calls above, plus always:
 * Regressor fc1/fc2/decpose/decshape/deccam (whmr.py:118-126), Global_Orient_Regressor (whmr.py:295-301),
 * est_Tz linears (whmr.py:425-427), second Tz conv (whmr.py:420), the timm Block linears (whmr.py:423). */
int whmr_gemm_f32(const struct whmr_gemm* p, int flags, void* stream);

/* LayerNorm over the last dim (C % 4 == 0, C <= 2048), fp32 in, fp32 or bf16 out.  vit.py:125,133,212,242. */
int whmr_layernorm(const float* x, const float* gamma, const float* beta, void* y, int rows, int C, float eps,
                   int out_bf16, void* stream);

/* PatchEmbed gather: NCHW fp32 image (element strides sb, sc, sh, sw) -> [B*Hp*Wp, Cin*P*P] patch rows.  vit.py:157,161. */
int whmr_patch_im2col(const float* x, void* cols, int B, int Cin, int H, int W, int P, int pad,
                      long sb, long sc, long sh, long sw, int out_bf16, void* stream);

/* fp32 -> bf16 cast (weight preparation). */
int whmr_cast_f32_bf16(const float* src, void* dst, long n, void* stream);

/* softmax(scale * Q K^T) V per (image, head) on qkv [B, N, 3, H, d] -> out [B, N, H*d].  vit.py:102-111.
 * is_bf16 = 1: MFMA kernel (d == 64, N <= 256); 0: fp32 kernel (N <= 256, any d). */
int whmr_attention(const void* qkv, void* out, int B, int N, int H, int d, float scale, int is_bf16, void* stream);

#ifdef __cplusplus
}
#endif
#endif
