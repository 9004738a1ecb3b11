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
    void* workspace;        /* optional split-K scratch: splits*M*N floats; null = no split */
    int64_t workspace_bytes;
    /* all 4 sub-pixel phases of ConvTranspose2d(k4,s2,p1) in one launch (bf16 kernel, a_mode = c_mode = 1, n_phase = 4):
     * phase = 2*py + px: W += phase*phase_w_stride; PH -= py; PW -= px; c_off += py*phase_cy + px*phase_cx. */
    int32_t n_phase;
    int32_t epi_flags;      /* bit 0: residual is bf16 (else fp32); bit 1: residual is added BEFORE the activation (ResNet blocks);
                             * bit 2 (bf16 kernel, c_mode 1, bf16 residual and output): the residual is addressed like C (c_off / osb / osy / osx) --
                             * with residual == C the scattered result is ACCUMULATED in place (data gradient of a strided convolution added to
                             * the gradient other consumers of the same map already left there);
                             * bits 4 / 5 (whmr_gemm_f32, at most 1024 output rows, a_mode = c_mode = 0): A / W is given REDUCTION-MAJOR -- A as [K, lda >= M],
                             * W as [K, N] dense -- the backward products of nn.Linear (dX = dY . W, dW = dY^T . X) without transposed copies;
                             * bit 7 (bf16 kernel, bf16 residual and output, row-major C): the residual is the pre-activation Z of a GELU and the output is
                             * bf16(acc + bias) * gelu'(Z) instead of a sum -- fc2's data gradient and the GELU backward in one pass (autograd of vit.py:66-68);
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

/* Up to 9 gathering (a_mode 1) bf16-output GEMMs of one tile shape as ONE launch (tile 192: 192 x 256 x 64; 65: 128 x 64 x 64 for N <= 64): no
 * activation, no split-K, no sub-pixel phases, scattered / accumulating epilogues allowed.  The S*S residue-class data gradients of a strided
 * convolution (autograd of whmr.py:419 `nn.Conv2d(256, 64, 7, 3)`; w-hmr_amd/train/heads_autograd.py) are nine such GEMMs of under two rounds of
 * tiles each.  Descriptors are copied into the kernel arguments: nothing of `ps` is read after the call returns. */
int whmr_gemm_bf16_group(const struct whmr_gemm* ps, int n, int tile, void* stream);

/* bf16 MFMA GEMM (v_mfma_f32_32x32x16_bf16, fp32 accumulate).  Needs K % 64 == 0 (and Cin % 64 == 0 for a_mode 1).
 * Replaces nn.Linear at vit.py:93,96 (qkv, proj), vit.py:66-68 (fc1, fc2), Conv2d at vit.py:157 (after
 * whmr_patch_im2col), ConvTranspose2d+BN+ReLU at whmr.py:488-498 and Conv2d at whmr.py:419.
 * flags 0/1: automatic tile choice (see gemm_bf16.hip); > 1: explicit tile id. */
int whmr_gemm_bf16(const struct whmr_gemm* p, int flags, void* stream);
/* Same contract with an explicit tile id: 64 = 128x128x64 (4 waves, 2 blocks/CU), 128 = 128x256x32 (4 waves, 3-stage),
 * 192 = 192x256x64, 257 = 256x256x64 (8 waves, 2-stage), 256 = 256x256x32 (8 waves, 4-stage). */
int whmr_gemm_bf16_big(const struct whmr_gemm* p, int tile, void* stream);
/* Tuning switch for in-process A/B measurements (key 1: ping-pong 256x256 main loop on/off, default off;
 * keys 100..103: force a tile id for the qkv / proj / fc1 / fc2 shapes of the ViT, 0 = chooser). */
int whmr_set_option(int key, int value);
/* Same, K sliced over `splits` blocks per tile (fp32 partial sums in p->workspace, deterministic epilogue pass). */
int whmr_gemm_bf16_split(const struct whmr_gemm* p, int tile, int splits, void* stream);
/* Split-K without the finishing pass: raw fp32 partial sums of `splits` equal K slices as [splits][M][N] planes at p->C (no bias / activation /
 * residual; K % (64 * splits) == 0) -- for a consumer that adds the planes itself (whmr_tz_fold). */
int whmr_gemm_bf16_split_raw(const struct whmr_gemm* p, int tile, int splits, void* stream);

/* ---- forward glue: the O(batch) arithmetic between the kernels of WHMR.forward, one launch each (geometry.hip) ----
 * whmr_cam_head: camera-calibration head post-processing (whmr.py:513-522; utils/cam_utils.py:121-145; pare softargmax1d / batch_euler2matrix):
 *   logits [Bf, ld >= 3 D] = [vfov | pitch | roll] bins (D <= 256) -> soft-argmax -> angles -> cam_rotmat = R([pitch, 0, roll]),
 *   render_rotmat = R([-pitch, 0, roll]), [B, 3, 3] each; Bf == 1 broadcasts one frame to all B crops, else Bf == B.
 * whmr_orient_state: input state of Global_Orient_Regressor (whmr.py:295-297): xc[b, F..F+15) = [rot6d(cam_rotmat[b]) | rotmat[b, 0]].
 * whmr_orient_tail: its tail (whmr.py:301-305,630-640): r [B, 9] -> unbiased Gram-Schmidt -> g_rot, angle-axis;
 *   g_pose [B, 72] = [aa | pose_aa[:, 3:]], g_rotmat [B, 216] = [g_rot | rotmat[:, 9:]]. */
int whmr_cam_head(const float* logits, int ld, int D, float pitch_lo, float pitch_hi, float roll_lo, float roll_hi, int Bf, int B,
                  float* cam_rotmat, float* render_rotmat, void* stream);
int whmr_orient_state(const float* cam_rotmat, const float* rotmat, long ld_rot, float* xc, long ld, int F, int B, void* stream);
int whmr_orient_tail(const float* r, const float* pose_aa, const float* rotmat, float* g_pose, float* g_rotmat, int B, void* stream);

/* ---- blocked-layout bf16 GEMM: the ViT's bf16 inference path (vit.py:61-140 qkv / proj / fc1 + GELU / fc2, vit.py:157 patch embed).
 * A [R, C] matrix is stored as [ceil(R/32)][C/E][32][E], E = 8 (bf16) / 4 (fp32): 512-byte units of 32 rows x 16 bytes -- the unit an
 * MFMA operand fetch reads (lane = row, 8 consecutive k) AND the (operand-swapped) result owns (lane = row, 8 consecutive columns).  LDS-DMA
 * staging copies whole units, the epilogue stores straight from the accumulators; two wave groups run in ping-pong (gemm_blk.hip).  bf16
 * operands compute on v_mfma_f32_16x16x32_bf16, split-bf16 operand pairs (A_lo / W_lo set) on 32x32x16; the packed layouts are the same. */
struct whmr_gemm_blk_desc {
    const void* A;        /* bf16 blocked [ceil(M/32)][K/8][32][8] */
    const void* W;        /* bf16 blocked [N/32][K/8][32][8] (nn.Linear weight [N, K], packed once) */
    void* C;              /* epi 0/1: bf16 blocked [ceil(M/32)][N/8][32][8]; epi 2/3: fp32 blocked [ceil(M/32)][N/4][32][4] */
    const float* bias;    /* [N] or null */
    const float* res;     /* epi 2: fp32 blocked like C (may alias C); epi 3: row-major [res_rows, N], row = m % res_rows (pos embed, vit.py:320) */
    int32_t M, N, K;      /* N % 256 == 0, K % 32 == 0; buffers hold whole 32-row blocks */
    int32_t epi;          /* 0: bf16(acc + bias); 1: bf16(gelu(acc + bias)); 2 / 3: fp32(acc + bias + res) */
    int32_t res_rows;
    int32_t tile;         /* 0 = chooser, else (MI0 << 4) | MI1 row blocks of the two wave rows: 0x44 = 256 x 256, 0x55 = 320, 0x43 = 224, ... */
    /* LayerNorm folding (vit.py:125,133 never run as their own pass): a producer (epi 2 / 3, xhat != null) also writes xhat = bf16(C) in the
     * blocked operand layout and per row and 256-column tile the partial sums (sum x, sum x^2): stats_out [rows][N/256][2] (N <= 1024).  A consumer (epi 0 / 1,
     * stats_in != null) multiplies A = xhat by W = bf16(gamma o W) and applies out = rstd[m] * (acc - mean[m] * colsum[n]) + bias[n], with
     * bias = b + W.beta, colsum[n] = sum_k W'[n,k], mean / rstd of row m from its K/256 partial pairs. */
    void* xhat; float* stats_out; const float* stats_in; const float* colsum; float ln_eps;
    /* split-bf16 operands ("bf16x3" numerics: the parity-grade mode of the same path -- the reference's Linears are fp32, vit.py:61-115): every
     * operand is a pair x = x_hi + x_lo (x_hi = bf16(x), x_lo = bf16(x - x_hi): 16 significand bits) and a product is three MFMAs per fragment
     * pair (hi.hi + lo.hi + hi.lo, fp32 accumulate).  A_lo / W_lo: lo halves in the layout of A / W; C_lo: lo half of a bf16 result (epi 0 / 1,
     * required there; epi 1 then applies the erf GELU of nn.GELU to fp32 accuracy: A&S 7.1.28, 8.7e-7 of float64).  All three null = plain bf16 operands.  The LayerNorm fold works in
     * this numerics too: a producer then also needs xhat_lo, a consumer takes W = the hi / lo pair of gamma o W and colsum of their sum. */
    const void* A_lo; const void* W_lo; void* C_lo;
    /* per-row shift of a folding producer (xhat != null): xhat / stats_out are taken of (C - s_m), s_m = (shift ? shift[m] : 0) + (shift_stats ?
     * the mean of row m from shift_stats [rows][N/256][2] : 0), shift_out[m] = s_m when non-null.  LayerNorm is shift-invariant, so the consumer
     * is unchanged; what is rounded to bf16 is the CENTRED row (error relative to the row's spread, not its offset).  shift_stats != stats_out. */
    const float* shift; const float* shift_stats; float* shift_out;
    void* xhat_lo;        /* split-bf16 producer (A_lo and xhat set): lo half of the centred-row operand pair (the fold in the bf16x3 numerics) */
};
int whmr_gemm_blk(const struct whmr_gemm_blk_desc* p, void* stream);
int whmr_gemm_blk_tile(const struct whmr_gemm_blk_desc* p, int tile, void* stream);
/* A/B switch: force a tile for the ViT-B shapes (slot 0 qkv N = 2304, 1 proj, 2 fc1 N = 3072, 3 fc2); 0 = chooser.
 * slot 4 (lab, tools/r6_chain_ab.py only): tile != 0 runs whmr_gemm_blk_chain with workgroup-scope fences -- NOT correct across XCDs, timing only. */
int whmr_gemm_blk_set_tile(int slot, int tile);
/* PILOT (round 6; off by default): fc1 (epi 1) -> fc2 (epi 2) of one transformer layer (vit.py:61-76 inside vit.py:117-140) as ONE persistent launch: grid = CU
 * count, every workgroup walks a static list [fc1 tiles | fc2 tiles], an fc2 tile waits on the arrive counters of the fc1 row panels it reads (bounded
 * spin: *err is raised after 20 ms, never a hang).  Same tile bodies as the two whmr_gemm_blk launches: bit-identical results.  Takes exactly the pair of
 * the ViT-B inference path (fc2->A == fc1->C, same M, plain bf16 operands; tiles 320 x 256 / 160 x 256); anything else returns hipErrorInvalidValue and
 * nothing is launched.  counters: >= ceil(M / 320) uint32 (zeroed by the call); err: one int32 (caller zeroes it once). */
int whmr_gemm_blk_chain(const struct whmr_gemm_blk_desc* fc1, const struct whmr_gemm_blk_desc* fc2, void* counters, void* err, void* stream);
/* LayerNorm on the blocked fp32 residual stream -> blocked bf16 GEMM operand (out_std 0) or row-major fp32 [rows, C] (out_std 1: last_norm). */
int whmr_layernorm_blk(const float* x, const float* gamma, const float* beta, void* y, int rows, int C, float eps, int out_std, void* stream);
/* The same (out_std 0) + the row means mean_out [ceil(rows/32)*32]: the first LayerNorm of the folded chain (vit.py:125 of block 0) runs explicitly and
 * seeds the per-row shift of the folding producers behind it (whmr_gemm_blk_desc.shift). */
int whmr_layernorm_blk_mean(const float* x, const float* gamma, const float* beta, void* y, float* mean_out, int rows, int C, float eps, void* stream);
/* PatchEmbed gather (vit.py:157,161) into the blocked bf16 operand layout. */
int whmr_patch_im2col_blk(const float* x, void* cols, int B, int Cin, int H, int W, int P, int pad, long sb, long sc, long sh, long sw,
                          void* stream);
/* whmr_attention (bf16, d = 64, 64 < N <= 256) on blocked qkv [ceil(B*N/32)][3*H*8][32][8] -> blocked out [ceil(B*N/32)][H*8][32][8]. */
int whmr_attention_blk(const void* qkv, void* out, int B, int N, int H, float scale, void* stream);
/* bf16x3 numerics for the row-major / convolution GEMMs (whmr.py:419,488-498): fp32 rows [rows, C] -> bf16 rows [rows, 3C] = [hi | lo | hi]; with the
 * weights laid out [W_hi | W_hi | W_lo] along the same (per-tap channel) axis ONE whmr_gemm_bf16 launch with Cin' = 3 Cin accumulates the three
 * split products in fp32. */
int whmr_split3_bf16(const float* src, void* dst, long rows, int C, void* stream);
/* ---- split-bf16 ("bf16x3") forms of the three blocked helpers: results / operands as hi + lo bf16 pairs (16 significand bits) ----
 * LayerNorm (vit.py:125,133) of the blocked fp32 stream -> blocked operand pair;  PatchEmbed gather (vit.py:157,161) -> blocked pixel pair;
 * attention core (vit.py:102-111; d = 64, 64 < N <= 256): three MFMAs per product in Q.K^T and in P.V, fp32 softmax, P split in registers. */
int whmr_layernorm_blk_x3(const float* x, const float* gamma, const float* beta, void* y_hi, void* y_lo, float* mean_out /* nullable: row means */,
                          int rows, int C, float eps, void* stream);
int whmr_patch_im2col_blk_x3(const float* x, void* cols_hi, void* cols_lo, int B, int Cin, int H, int W, int P, int pad, long sb, long sc, long sh,
                             long sw, void* stream);
int whmr_attention_blk_x3(const void* qkv_hi, const void* qkv_lo, void* out_hi, void* out_lo, int B, int N, int H, float scale, void* stream);
/* A/B switch.  bit 0: the round-3 kernel (one workgroup per (image, head), 32-row tiles) for every N instead of the persistent 16-row-tile kernel
 * (csrc/attention_blk16.hip, N <= 208).  bits 4-7: k + 1 forces k half-microseconds between the start of the CU quarters of the persistent kernel
 * (0 = by shape: 1 us at 13 query tiles, none below).  0 = default. */
int whmr_attention_x3_set_variant(int variant);

/* exact-fp32 MFMA GEMM (v_mfma_f32_32x32x2_f32), any M/N/K.  Parity mode of the calls above, plus always:
 * Regressor fc1/fc2/decpose/decshape/deccam (whmr.py:118-126), Global_Orient_Regressor (whmr.py:295-301),
 * est_Tz linears (whmr.py:425-427), second Tz conv (whmr.py:420), the timm Block linears (whmr.py:423). */
int whmr_gemm_f32(const struct whmr_gemm* p, int flags, void* stream);
/* A/B switch of the large-M 128x128 double-buffered kernel behind whmr_gemm_f32 (1 = on, the default; 0 = always the 64x64 kernel): the two
 * produce the same bits (same k order of the MFMA chain). */
int whmr_gemm_f32_set_big(int on);

/* Weight-gradient product of the training step: C [Mo, No] (fp32, row stride ldc) = A^T . B with BOTH operands reduction-major, A [K, lda], B [K, ldb]
 * bf16 -- dW[n_out, k_in] = sum_m dY[m, n_out] . X[m, k_in] straight from the token- / pixel-major tensors the forward and backward kernels
 * leave (autograd of nn.Linear at vit.py:66-68,93,96,157 as run by core/trainer.py:410-470); no transposed operand copies.  Mo % 64 == 0,
 * No % 256 == 0, K % 32 == 0, 16-B aligned rows; splits = 0 lets the launcher slice K (deterministic fixed-order reduction through `workspace`,
 * fp32, >= splits * Mo * (No + 1) * 4 bytes), splits = 1 needs no workspace.  db (nullable) [Mo] receives the column sums of A in the same pass:
 * the bias gradient when A is dY.  hipErrorInvalidValue outside that envelope. */
int whmr_gemm_tn_bf16(const void* A, long lda, const void* B, long ldb, float* C, long ldc, float* db, int Mo, int No, int K, int splits,
                      void* workspace, long workspace_bytes, void* stream);

/* The same product for up to 4 Linears that share the reduction length K (= tokens) in ONE launch: the weight gradients of one transformer layer
 * (qkv, proj, fc1, fc2 of vit.py:61-165 under loss.backward()).  One by one each of them slices K 7-28 ways to own the chip and moves ~65 MB of
 * fp32 partial tiles through the workspace twice; together their 108 tiles fill it with two slices.  Every item: Mo % 256 == 0, No % 256 == 0,
 * otherwise the envelope above; db nullable.  Deterministic (fixed slice order); with another slice count than the single launch the last bits
 * differ.  workspace (fp32 partials): >= sum over items of splits * Mo * (No + 1) * 4 bytes with splits = 256 / (total 256 x 256 tiles), else
 * the launch runs unsliced.  hipErrorInvalidValue outside the envelope (the caller then issues the single launches). */
struct whmr_tn_item {
    const void* A; long lda;      /* [K, lda >= Mo] bf16: dY */
    const void* B; long ldb;      /* [K, ldb >= No] bf16: X  */
    float* C; long ldc;           /* [Mo, ldc >= No] fp32: dW */
    float* db;                    /* [Mo] fp32 column sums of A, or null */
    int Mo, No;
};
int whmr_gemm_tn_bf16_group(const struct whmr_tn_item* items, int n_items, int K, void* workspace, long workspace_bytes, void* stream);

/* Convolution weight gradient without a column matrix (same kernel, the B operand gathered): C [Mo, KH*KW*GC] (fp32) = A^T . col(img) with
 * A [K = nB*OH*OW, lda] bf16 (one row per position of the OH x OW grid) and img [nB, IH, IW, ldp >= GC] bf16 NHWC; column (tap = ky*KW + kx, c) of
 * reduction row (b, oy, ox) is img[b, oy*S + ky - P, ox*S + kx - P, c], zero outside the image.  Autograd of Conv2d (A = dY over the output
 * grid, img = X -> dW[co, (ky,kx,ci)]: whmr.py:419-420, models/iuv_predictor.py:71-91) and of ConvTranspose2d(k4, s2, p1) (A = X over the input
 * grid, img = dZ, S = 2, P = 1 -> dW[ci, (ky,kx,co)]: whmr.py:488-498) as core/trainer.py:410-470 runs them; replaces whmr_im2col_t + the operand
 * transposes + the NT GEMM.  Mo % 128 == 0 (the 64-row tile of the gathering kernel is not built: gemm_tn.hip), GC % 256 == 0, K % 32 == 0; zeros: >= 512 B of device zeros; db (nullable) [Mo] = column sums of A
 * (the convolution's bias gradient when A is dY). */
int whmr_conv_dw_tn_bf16(const void* A, long lda, const void* img, long ldp, float* C, long ldc, int Mo, int K, int nB, int OH, int OW,
                         int IH, int IW, int GC, int KH, int KW, int S, int P, const void* zeros, int splits, void* workspace,
                         long workspace_bytes, float* db, void* stream);
/* the same with separate row / column strides of the gather (SY, SX) */
int whmr_conv_dw_tn2_bf16(const void* A, long lda, const void* img, long ldp, float* C, long ldc, int Mo, int K, int nB, int OH, int OW,
                          int IH, int IW, int GC, int KH, int KW, int SY, int SX, int P, const void* zeros, int splits, void* workspace,
                          long workspace_bytes, float* db, void* stream);

/* LayerNorm over the last dim (C % 4 == 0, C <= 2048), fp32 in, fp32 or bf16 out.  vit.py:125,133,212,242. */
int whmr_layernorm(const float* x, const float* gamma, const float* beta, void* y, int rows, int C, float eps,
                   int out_bf16, void* stream);

/* PatchEmbed gather: NCHW fp32 image (element strides sb, sc, sh, sw) -> [B*Hp*Wp, Cin*P*P] patch rows.  vit.py:157,161. */
int whmr_patch_im2col(const float* x, void* cols, int B, int Cin, int H, int W, int P, int pad,
                      long sb, long sc, long sh, long sw, int out_bf16, void* stream);

/* fp32 -> bf16 cast (weight preparation). */
int whmr_cast_f32_bf16(const float* src, void* dst, long n, void* stream);

/* softmax(scale * Q K^T) V per (image, head) on qkv [B, N, 3, H, d] -> out [B, N, H*d].  vit.py:102-111.
 * is_bf16 = 1: bf16 MFMA kernel (d == 64, N <= 256); 0: fp32 -- exact-f32 MFMA kernel with an online softmax for d == 64, 32 < N <= 256 (the backbone
 * in the parity mode), VALU kernel for any other d / N <= 256 (the 5-token timm Block of the Tz head). */
int whmr_attention(const void* qkv, void* out, int B, int N, int H, int d, float scale, int is_bf16, void* stream);
/* A/B switches: bit 0: 1 = chunked online-softmax bf16 variant (2 workgroups / CU, default), 0 = single pass; bits 1-2: timing ablations of the
 * round-2 blocked kernel (wrong results); bit 3 set: fp32 attention always on the VALU kernel; bit 4 set: whmr_attention_blk on the round-2 kernel
 * (one workgroup per (image, head), 32-row tiles) instead of the persistent 16-row-tile kernel. */
int whmr_attention_set_variant(int chunked);

/* ---- rotation / projection helpers: utils/geometry.py ------------------------------------------------------------ */
/* mode 0: rot6d_to_rotmat (geometry.py:243-257, in [n,6]); 1: unbiased_gram_schmidt (:260-272, in [n,9]);
 * 2: batch_rodrigues (:14-27, in [n,3]).  out [n,9] row-major. */
int whmr_rot_to_mat(const float* in, float* out, int n, int mode, void* stream);
/* rotation_matrix_to_angle_axis (geometry.py:54-83 via :160-240 and :86-136): [n,9] -> [n,3], NaN -> 0. */
int whmr_mat_to_aa(const float* in, float* out, int n, void* stream);
/* backward of the above: d_in [n,9] = (d aa / d R)^T d_out [n,3] -- what torch autograd computes through geometry.py:54-83 in the reference's training
 * graph (whmr.py:174: theta's pose; whmr.py:632-633: global_pose).  Same branch as the forward; a NaN component passes no gradient (the forward's masked
 * write); at sin^2 == 0 exactly the selected branch k = 2 is differentiated (torch yields NaN there). */
int whmr_mat_to_aa_bwd(const float* in, const float* d_out, float* d_in, int n, void* stream);
/* perspective_projection (geometry.py:310-341).  rot may be null (identity) or batch-1 (rot_bstride 0, else 9);
 * focal per image (focal_bstride 1) or one scalar (0); center / post_div may be null.
 * out = K (R p + t) / z  [ / post_div[b] + post_shift ]  (the latter = whmr.py:173). */
int whmr_perspective(const float* pts, const float* rot, int rot_bstride, const float* trans, const float* focal,
                     int focal_bstride, const float* center, const float* post_div, float post_shift, float* out,
                     int B, int P, void* stream);
/* projection (geometry.py:289-307): weak-perspective camera [B,3] -> [-1,1] crop coordinates. */
int whmr_weak_projection(const float* pts, const float* cam, float* out, int B, int P, float focal, float res_w,
                         float res_h, void* stream);

/* ---- SMPL forward: pare.models.SMPL / smplx lbs as called at whmr.py:132-137,227-232,641-644 ---------------------- */
struct whmr_smpl_model {
    const float* v_template;         /* [6890,3] */
    const float* shapedirs;          /* [30,6890]: shapedirs[v][c][l] stored as [(c*10+l)][v] (transposed once by the host) */
    const float* posedirs;           /* [207,20670] (smplx layout) */
    const float* lbs_weights;        /* [24,6890]: transposed once by the host */
    const float* J_template;         /* [24,3]    = J_regressor . v_template  (folded once on the host) */
    const float* J_shapedirs;        /* [24,3,10] = J_regressor . shapedirs */
    const float* J_regressor;        /* [24,6890] (for smpl_joints45; may be null) */
    const float* J_regressor_extra;  /* [9,6890] */
    const int32_t* parents;          /* [24] */
    const int32_t* extra_vertex_ids; /* [21] VertexJointSelector picks */
    const int32_t* joint_map;        /* [49] into the 54-joint superset (core/constants.py:16-92) */
    const int32_t* marker_ids;       /* [n_markers] (data/smpl/smpl_ssm.npy, whmr.py:100,184) */
    int32_t n_markers;
};
/* pose9 [B,24,9] (+ optional unbiased Gram-Schmidt, whmr.py:129-130) -> rotmat [B,24,9], angle-axis [B,72] (whmr.py:174),
 * skinning transforms A [B,24,12], posed joints [B,24,3], pose feature [B,207].  Null outputs are skipped (A required). */
/* pose_stride / beta_stride: row strides in elements (the operands may be column slices of a [.., 216|10|3] state buffer). */
int whmr_smpl_pose_chain(const struct whmr_smpl_model* m, const float* pose9, long pose_stride, const float* betas,
                         long beta_stride, int B, int do_gs, float* rotmat, float* aa, float* A, float* posed_joints,
                         float* pose_feat, void* stream);
/* blend shapes + pose-corrective offsets + linear blend skinning -> verts [B,6890,3].  pose_off (nullable) = [B,20670]
 * pose_feat . posedirs precomputed by whmr_gemm_f32; null = computed in the kernel. */
int whmr_smpl_skin(const struct whmr_smpl_model* m, const float* betas, long beta_stride, const float* pose_feat, const float* A,
                   const float* pose_off, int B, float* verts, void* stream);
/* joints49 [B,49,3] (24 posed + 21 vertex picks + 9 regressed, JOINT_MAP), optional smpl_joints45 [B,45,3]
 * (whmr.py:186-187) and markers [B,n_markers,3] (whmr.py:184). */
/* scratch: >= B*33*3 floats of workspace.  When smpl_joints45 is requested, J_regressor_extra and J_regressor must be one
 * contiguous [33,6890] buffer (extra rows first). */
int whmr_smpl_joints(const struct whmr_smpl_model* m, const float* verts, const float* posed_joints, int B,
                     float* joints49, float* smpl_joints45, float* markers, float* scratch, void* stream);

/* Tail of Regressor.forward (whmr.py:142-174,190) fused: state row = [pose(216) | shape(10) | cam(3)] (row stride
 * state_stride) -> theta [B,85], kp_2d [B,49,2] (geometry.py:289-307), focal = s*h*Tz/2 [B] (whmr.py:147-149), cam_t [B,3]
 * (geometry.py:139-157), kp_2d_w [B,49,2] = perspective(joints, cam_t, focal, centre)/centre - 1 (whmr.py:165-173). */
int whmr_regressor_post(const float* state, long state_stride, const float* aa, const float* joints49, const float* Tz,
                        const float* bbox_h, const float* center, const float* orig_shape, int B, float focal0, float res_w,
                        float res_h, float* theta, float* kp2d, float* kp2d_w, float* cam_t, float* focal, void* stream);

/* ---- MAF sampler: models/maf_extractor.py:75-143 ------------------------------------------------------------------ */
struct whmr_maf_weights {
    const float* w0t; const float* b0;   /* conv0 transposed [256][128], bias [128] */
    const float* w1t; const float* b1;   /* conv1 transposed [384][64]  (inputs: y0 | raw feature) */
    const float* w2t; const float* b2;   /* conv2 transposed [320][32]  (inputs: y1 | raw feature) */
    const void *w0b, *w1b, *w2b;         /* optional bf16 copies in Conv1d layout [out][in]: with a bf16 channels-last map the MLP runs on MFMA */
};
/* (weak-perspective projection of pts3d with cam ->) bilinear grid_sample(align_corners=True, zero padding) of 256
 * channels at P points -> 3-layer point MLP -> out[b*out_stride + c*P + p].  fmap is addressed by element strides
 * (sb, sc, sy, sx), fp32 or bf16.  With neither pts2d nor pts3d, fmap is [B,256,P] pre-sampled features (reduce_dim).
 * point_feat (nullable) receives the raw sampled features [B,256,P]. */
int whmr_maf_sample(const void* fmap, int fmap_bf16, long sb, long sc, long sy, long sx, int H, int W,
                    const float* pts2d, const float* pts3d, const float* cam, long cam_ld, float focal, float res_w, float res_h,
                    const struct whmr_maf_weights* w, int B, int P, float* out, long out_stride, float* point_feat,
                    void* stream);

/* ---- NHWC helpers for the camera-calibration ResNet-50 on the implicit-GEMM kernel (models/cam_model.py:24-81) */
/* NCHW fp32 image (element strides) -> cols [B*OH*OW, Kpad] bf16, k = (ci*KH + ky)*8 + kx (kx padded to 8, KW <= 8),
 * zero padded to Kpad (stem conv 7x7 s2 p3: Kpad = 192). */
int whmr_conv_im2col(const float* x, void* cols, int B, int Cin, int H, int W, int KH, int KW, int S, int pad, int Kpad,
                     long sb, long sc, long sh, long sw, void* stream);
/* MaxPool2d(k, s, pad) on NHWC, bf16 (C % 8 == 0) or fp32. */
int whmr_maxpool_nhwc(const void* x, void* y, int B, int H, int W, int C, int k, int s, int pad, int is_bf16, void* stream);
/* AdaptiveAvgPool2d((1,1)) on NHWC bf16 / fp32 -> [B, C] fp32 (C % 64 == 0). */
int whmr_avgpool_nhwc(const void* x, float* y, int B, int HW, int C, int is_bf16, void* stream);

/* Regressor input assembly (whmr.py:105,119): xc[b, F..F+234) = [bbox_info(5) | pose(216) | shape(10) | cam(3)].  pose / shape /
 * cam rows have element row strides ld_* (0 = one broadcast row). */
int whmr_regressor_state(const float* bbox_info, const float* pose, long ld_p, const float* shape, long ld_s, const float* cam,
                         long ld_c, int B, float* xc, long ld, int F, void* stream);

/* ---- input side: cv2.warpAffine (8-bit INTER_LINEAR, BORDER_CONSTANT 0) + ToTensor + Normalize of
 * datasets/data_utils/img_utils.py:89-101,209-242,318-326 for all B detections of one uint8 HWC frame.  inv_affine: B x 6
 * doubles, the inverse 2x3 maps (patch pixel -> frame pixel).  out [B,3,patch_h,x_end-x_begin] fp32; raw (nullable) uint8 HWC. */
int whmr_crop_normalize(const uint8_t* frame, int H, int W, long row_stride, const double* inv_affine, int B, int patch_w,
                        int patch_h, int x_begin, int x_end, float* out, uint8_t* raw, const float* mean3, const float* std3,
                        void* stream);

/* ---- backward-pass helpers of the ViT backbone (autograd of vit.py:61-140,313-332; driven by core/trainer.py:410-470).
 * The backward matrix products run on whmr_gemm_bf16 / whmr_gemm_f32: dX = dY.W uses W^T as the weight operand,
 * dW = dY^T.X uses the transposed activations -- both produced by whmr_transpose_cast. */
/* src [R,C] (row stride ld_src) -> dst [C,Rpad] (row stride ld_dst >= Rpad), fp32 <-> bf16; columns R..Rpad-1 are zero filled. */
int whmr_transpose_cast(const void* src, int src_bf16, long ld_src, void* dst, int dst_bf16, long ld_dst, int R, int C, int Rpad,
                        void* stream);
/* whmr_transpose_cast (bf16 -> bf16, 8-aligned shapes) fused with whmr_colsum of the source: the transpose of dY for the dW product also
 * yields the bias gradient.  scratch >= ceil(R/64)*C floats. */
int whmr_transpose_colsum(const void* src, long ld_src, void* dst, long ld_dst, int R, int C, int Rpad, float* out, int accumulate,
                          float* scratch, void* stream);
/* out[c] (+)= sum_r x[r,c]; deterministic two-stage sum; scratch >= max(64*C, 2^20) floats. */
int whmr_colsum(const void* x, int is_bf16, long ld, int R, int C, float* out, int accumulate, float* scratch, void* stream);
/* bf16 operand copies of ALL weights of a module in one launch (training: the optimizer rewrites every weight every step).  Item = fp32 matrix
 * src [N, K] (nn.Linear layout: vit.py:66-68,93,96,157), dst [N, K] bf16 (W operand of y = x . W^T) and dst_t [K, N] bf16 (W operand of the data
 * gradient dX = dY . W); either may be null.  items: DEVICE array sorted by tile_begin; item i owns ceil(N/64) * tiles_k 64x64 tiles, tiles_k =
 * ceil(K/64); total_tiles = their sum.  Same bits as whmr_cast_bf16 / whmr_transpose_cast. */
struct whmr_wprep_item { const float* src; void* dst; void* dst_t; int32_t N, K, tile_begin, tiles_k; };
int whmr_weights_prepare(const void* items, int n_items, int total_tiles, void* stream);
/* LayerNorm backward: dx = dLN(x; gamma)(dy) + dres (dres nullable, dx may alias it); dgamma/dbeta (+)=; scratch >= 2048*C floats.
 * cast_out (nullable) [rows, C] bf16 = dx * row_scale[row] (row_scale nullable): the operand of the next branch's backward GEMMs, stochastic-depth
 * factor included (vit.py:132-139), written by the same pass. */
int whmr_layernorm_bwd(const float* x, const float* dy, const float* gamma, const float* dres, float* dx, float* dgamma,
                       float* dbeta, int accumulate, int rows, int C, float eps, float* scratch, void* cast_out, const float* row_scale,
                       void* stream);
/* exact-erf GELU: forward as its own pass (training keeps the pre-activation) and backward d_pre = d_hid * gelu'(pre). */
int whmr_gelu_fwd(const void* pre, void* out, int is_bf16, long n, void* stream);
int whmr_gelu_bwd(const void* pre, int pre_bf16, const void* dhid, int dhid_bf16, void* dpre, int out_bf16, long n, void* stream);

/* Attention core for training (bf16, d = 64, 64 < N <= 224): forward that also writes the per-query log-sum-exp (log2 domain,
 * lse [B,H,N] fp32), and the MFMA backward dqkv [B,N,3,H,64] bf16 from qkv, o = forward output, dout = d(o) fp32, lse. */
int whmr_attention_fwd_train(const void* qkv, void* out, float* lse, int B, int N, int H, int d, float scale, void* stream);
int whmr_attention_bwd(const void* qkv, const void* o, const float* dout, const float* lse, void* dqkv, int B, int N, int H, int d,
                       float scale, void* stream);

/* Second convolution of the Tz head (whmr.py:420 + the reshape at :571): Conv2d(64,5,k7,s2) on the NHWC map x [B,IH,IW,64]
 * (x_bf16 = 1: bf16, 0: fp32; 2: fp32 [B,IH,IW,128] whose channel halves c and 64 + c are added as they are read -- the bf16x3 form of the first
 * convolution leaves its W_lo product in columns 64..127), weights w [5][7*7][64] fp32 -> tokens [B,5,OH*OW] fp32. */
int whmr_tz_conv1(const void* x, int x_bf16, const float* w, float* tok, int B, int IH, int IW, void* stream);

/* Tail of the COMPOSED Tz-head convolutions (whmr.py:418-421 as applied at :567-571; inference views): conv1(conv0(x)) == Conv2d(256,5,k25,s6), evaluated as
 * the implicit GEMM P[(b,Y,m),(jA,jB,o)] = sum_{q,p,ci} x[b,6Y+q,6m+p,ci] * Wc[o,ci,q+6jA,p+6jB] (whmr_gemm_bf16 / whmr_gemm_f32 in conv-gather mode over the
 * [B,IH,IW/6,6C] view of the map: kernel 6x1, stride 6x1; every map byte is read once).  This entry sums the 25 shifted partials:
 * tok[b,o,r*OW+s] = sum_{jA,jB} P[(b,r+jA,s+jB),(jA*5+jB)*5+o].  P: fp32 rows of ldp floats, [B,OHp,OWp] pixel rows; halves = 2 adds column 128+n
 * (bf16x3: the W_lo product as extra output columns); nsplit partial planes split_stride floats apart are added (1 = a finished GEMM output). */
int whmr_tz_fold(const float* P, int ldp, int halves, int nsplit, long split_stride, float* tok, int B, int OHp, int OWp, int OH, int OW,
                 void* stream);

/* The composed Tz convolution in the TRAINING graph (autograd of whmr.py:567-571 through Wc = compose(w0, w1): w-hmr_amd/train/heads_autograd.py::TzComposedFn).
 * whmr_tz_compose: T [(u, v, o) = 245][(ci, a, b) = 49 Ci] fp32 (= w1 as [(u, v, o), c1] times w0 as [c1, (ci, a, b)]) -> the space-to-depth weight matrix
 * g [128][36 Ci] (bf16 when g_bf16, else fp32) of whmr.py:418-421 as ONE Conv2d(256, 5, k25, s6), rows (jA, jB, o), columns (q, p, ci).
 * whmr_tz_compose_bwd: the transpose, dG [128][36 Ci] fp32 -> dT [245][49 Ci].
 * whmr_tz_unfold: the transpose of whmr_tz_fold, dtok [B, 5, OH, OW] fp32 -> dP [B * OHp * OWp][128] bf16. */
int whmr_tz_compose(const float* T, int Ci, void* g, int g_bf16, void* stream);
int whmr_tz_compose_bwd(const float* dG, int Ci, float* dT, void* stream);
int whmr_tz_unfold(const float* dtok, void* dP, int B, int OHp, int OWp, int OH, int OW, void* stream);

/* estimate_translation (utils/geometry.py:344-408; trainer host stall): S [B,J,3], joints_2d [B,J,3] = (x, y, conf);
 * joints j0..j0+nj-1 enter the weighted least squares; out [B,3]. */
int whmr_estimate_translation(const float* S, const float* joints_2d, int B, int J, int j0, int nj, float focal, float img_w,
                              float img_h, float* out, void* stream);

/* Tz-head tail (whmr.py:574-577): tokens [B,T,D] -> mean over T -> Linear(D,Hd) -> Linear(Hd,1) -> BatchNorm1d(1) eval
 * (bn4 = weight, bias, running_mean, running_var) -> sigmoid -> x10. */
int whmr_tz_tail(const float* tok, int B, int T, int D, const float* w0, const float* b0, int Hd, const float* w1,
                 const float* b1, const float* bn4, float bn_eps, float* tz, void* stream);

/* ---- training mode of the deconv pyramid: ConvTranspose2d(k4,s2,p1) -> BatchNorm2d(batch statistics) -> ReLU
 * (models/whmr.py:459-501,560-564; their autograd as driven by core/trainer.py:410-470).  Maps are channels-last [M = B*H*W, C],
 * C % 8 == 0, 256 % (C/8) == 0.  The matrix products run on whmr_gemm_*: forward = the 4 sub-pixel phases without BN folding,
 * dX = Conv2d(k4,s2,p1) gather of dZ (a_mode 1), dW = X^T . whmr_im2col_t(dZ). */
/* stats [4*C] = mean | 1/sqrt(var+eps) | a = gamma*invstd | b = beta - mean*a; running_mean / running_var (nullable) get the
 * nn.BatchNorm2d momentum update (unbiased variance).  scratch >= 2050*C floats. */
int whmr_bn_stats(const void* z, int z_bf16, long M, int C, const float* gamma, const float* beta, float eps, float momentum,
                  float* running_mean, float* running_var, float* stats, float* scratch, void* stream);
/* y = relu(z*a + b). */
int whmr_bn_apply_relu(const void* z, int z_bf16, const float* stats, void* y, int y_bf16, long M, int C, void* stream);
/* backward of relu(bn(z)) for the upstream gradient dy: dz, dgamma, dbeta ((+)= when accumulate).  scratch >= 2050*C floats. */
int whmr_bn_relu_bwd(const void* z, int z_bf16, const void* dy, int dy_bf16, const float* stats, void* dz, int dz_bf16, float* dgamma,
                     float* dbeta, int accumulate, long M, int C, float* scratch, void* stream);
/* ---- SyncBatchNorm (core/trainer.py:83: nn.SyncBatchNorm.convert_sync_batchnorm before DDP; trained layers: 3 x BatchNorm2d(256) at whmr.py:497 and
 * BatchNorm1d(1) at whmr.py:428): whmr_bn_stats / whmr_bn_relu_bwd cut at the point where the ranks' sums meet -- the host all-reduces ONE packed
 * fp64 vector per layer and direction (torch.distributed, RCCL / gloo) between the two halves.
 *   whmr_bn_sums             sums64 [2C + 1] = sum z | sum z^2 | rows of THIS rank (un-shifted, double)
 *   whmr_bn_stats_from_sums  stats [4C] + running statistics (unbiased variance of the GLOBAL batch) from the summed vector
 *   whmr_bn_bwd_sums         sums64 [2C] = sum g | sum g xhat over this rank's rows (g = dy where relu(bn(z)) > 0, xhat from the global stats),
 *                            and the LOCAL dgamma / dbeta ((+)= when accumulate) -- torch's SyncBatchNorm keeps the parameter gradients per rank
 *   whmr_bn_bwd_apply        dz = a (g - sum_g / N - xhat sum_gx / N), sums64 summed over the ranks, N = *count (device: element 2C of the forward vector)
 * scratch >= 2050*C floats.  With one rank the pair of calls equals the unsplit entry up to the last bit of the double -> float roundings. */
int whmr_bn_sums(const void* z, int z_bf16, long M, int C, double* sums64, float* scratch, void* stream);
int whmr_bn_stats_from_sums(const double* sums64, int C, const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                            float* running_var, float* stats, void* stream);
int whmr_bn_bwd_sums(const void* z, int z_bf16, const void* dy, int dy_bf16, const float* stats, float* dgamma, float* dbeta, int accumulate, long M,
                     int C, double* sums64, float* scratch, void* stream);
int whmr_bn_bwd_apply(const void* z, int z_bf16, const void* dy, int dy_bf16, const float* stats, const double* sums64, const double* count, void* dz,
                      int dz_bf16, long M, int C, float* scratch, void* stream);
/* dst[(ky*KW+kx)*C + c][m] = src[b, oy*S+ky-P, ox*S+kx-P, c] (NHWC src, zero outside), m = (b, oy, ox) < B*OH*OW, row length Mpad. */
int whmr_im2col_t(const void* src, void* dst, int is_bf16, int B, int IH, int IW, int C, int OH, int OW, int KH, int KW, int S, int P,
                  long Mpad, void* stream);

/* ---- backward of the SMPL forward (autograd through pare.models.SMPL as called at whmr.py:132-137, pose2rot=False) -------------------
 * whmr_smpl_joints_bwd: gradients of joints49 [B,49,3] / smpl_joints45 [B,45,3] / markers (each nullable) -> d_posed_joints [B,24,3],
 *   d_regd [B,R,3] (R = 33 with d_smpl_joints45, else 9), and the vertex picks added IN PLACE into d_verts [B,6890,3].
 * whmr_smpl_skin_bwd: d_vposed [B,20670] and the per-block partial sums dA_partial [B,54,288] of the skinning-transform gradient.
 * whmr_smpl_chain_bwd: reverse kinematic chain -> d_rotmat [B,24,9], d_betas [B,10]; d_pf_beta [B,217] = d_vposed . [posedirs ; S]^T
 *   (S = shapedirs as [10, 20670]) comes from whmr_gemm_f32. */
int whmr_smpl_joints_bwd(const struct whmr_smpl_model* m, const float* d_joints49, const float* d_smpl_joints45, const float* d_markers,
                         int B, float* d_verts, float* d_posed_joints, float* d_regd, void* stream);
int whmr_smpl_skin_bwd(const struct whmr_smpl_model* m, const float* betas, long beta_stride, const float* A, const float* pose_off,
                       const float* d_verts, const float* d_regd, int R, int B, float* d_vposed, float* dA_partial, void* stream);
int whmr_smpl_chain_bwd(const struct whmr_smpl_model* m, const float* rotmat, const float* betas, long beta_stride, const float* dA_partial,
                        const float* d_posed_joints, const float* d_pf_beta, int B, float* d_rotmat, float* d_betas, void* stream);

/* ---- backward of the MAF sampler (autograd of maf_extractor.py:75-124; the sample points are constants: whmr.py:586-592 detaches
 * them).  d_out [B, 32*P] (row stride dout_stride) -> d_fmap (nullable; element strides gsb/gsc/gsy/gsx; ACCUMULATED into: fp32 maps with fp32
 * atomics, bf16 channels-last maps (gsc == 1) with a 32-bit compare-and-swap per channel pair) and the point-minor operands of the weight-gradient GEMMs: XT [448, ldt] = [y0 (128) ; f (256) ; y1 (64)],
 * DT [224, ldt] = [d_pre0 (128) ; d_pre1 (64) ; d_pre2 (32)], columns = b*P + p (ldt >= B*P).  dW0 = DT[0:128] . XT[128:384]^T,
 * dW1 = DT[128:192] . XT[0:384]^T, dW2 = DT[192:224] . [XT[384:448] ; XT[128:384]]^T, biases = row sums of DT (whmr_gemm_f32).
 * w = the forward's transposed weights; w0 / w1 / w2 = the Conv1d-layout [out][in] fp32 matrices. */
int whmr_maf_sample_bwd(const void* fmap, int fmap_bf16, long sb, long sc, long sy, long sx, int H, int W, const float* pts2d,
                        const float* pts3d, const float* cam, long cam_ld, float focal, float res_w, float res_h,
                        const struct whmr_maf_weights* w, const float* w0, const float* w1, const float* w2, int B, int P,
                        const float* d_out, long dout_stride, void* d_fmap, int d_fmap_bf16, long gsb, long gsc, long gsy, long gsx,
                        float* XT, float* DT, long ldt, void* stream);

/* The sampler's map gradient in two halves, for a map whose other consumers write their gradient first (the last feature map: Tz head and IUV head run
 * on a side stream).  whmr_maf_sample_bwd with d_fmap_bf16 = 2 leaves, instead of scattering, a compact fp32 record d_fmap [B*P][264]: row b*P + p =
 * 256 channel gradients of point p, its 4 texel offsets (y*W + x, int bits; -1 = outside the map) and its 4 bilinear weights.  whmr_maf_scatter adds
 * the records to a gradient map (same strides / dtype rules as d_fmap above; d_fmap_bf16 = 0 | 1) -- 4 texels per point instead of a dense
 * zero-filled map and a full-size add. */
int whmr_maf_scatter(const float* rec, int B, int P, int W, void* d_fmap, int d_fmap_bf16, long gsb, long gsc, long gsy, long gsx, void* stream);

/* col2im (gather form): dx [B,IH,IW,C] = fold of the column-space gradient dcol [(b,oy,ox)][(ky,kx,c)] (row stride ldcol) of a strided,
 * padded Conv2d -- the data gradient of the Tz-head convolutions (whmr.py:419-420) after dcol = dY . W on whmr_gemm_*. */
int whmr_col2im(const void* dcol, int dcol_bf16, long ldcol, void* dx, int dx_bf16, int B, int IH, int IW, int C, int OH, int OW,
                int KH, int KW, int S, int P, void* stream);

/* Sparse matrix (CSR: ptr [n_out+1], col, val) applied to B point sets: out[b, r, :] (+)= sum_k val[k] * in[b, col[k], :].  The mesh
 * down-sampling products of whmr.py:182-183 (the reference multiplies the densified 1723 x 6890 / 431 x 1723 matrices) and, with the CSR of
 * the transpose, their backward. */
int whmr_csr_apply3(const int32_t* ptr, const int32_t* col, const float* val, const float* in, int n_in, float* out, int n_out, int B,
                    int accumulate, void* stream);

/* Training form of the regressor tail (whmr.py:142-173; geometry.py:289-341,139-157): joints [B,J,3], cam [B,3], Tz [B] -> kp2d, kp2d_w
 * [B,J,2], cam_t [B,3] (from cam.detach()), focal [B] = s.detach()*h*Tz/2, and the backward of all four in one launch.  stage = cfg.TRAIN.STAGE:
 * 1 -> kp2d differentiates the joints, otherwise kp2d_w does (whmr.py:142-145,156-163).  Gradient inputs are nullable. */
int whmr_regressor_post_train(const float* joints, const float* cam, const float* Tz, const float* bbox_h, const float* center,
                              const float* orig_shape, int B, int J, float focal0, float res_w, float res_h, float* kp2d, float* kp2d_w,
                              float* cam_t, float* focal, void* stream);
int whmr_regressor_post_train_bwd(const float* joints, const float* cam, const float* Tz, const float* bbox_h, const float* center,
                                  const float* orig_shape, int B, int J, float focal0, float res_w, float res_h, int stage,
                                  const float* d_kp2d, const float* d_kp2d_w, const float* d_cam_t, const float* d_focal, float* d_joints,
                                  float* d_cam, float* d_Tz, void* stream);

/* fp32 backward of the attention core (autograd of vit.py:102-111 in the fp32 parity mode; the timm Block of the Tz head, whmr.py:423,574): qkv
 * [B, N, 3, H, d] + dout [B, N, H*d] -> dqkv [B, N, 3, H, d], N <= 256, any head dim; scratch >= B*H*2*N*N floats; deterministic. */
int whmr_attention_bwd_f32(const float* qkv, const float* dout, float* dqkv, float* scratch, int B, int N, int H, int d, float scale, void* stream);
/* dst[m, :] = scale[m] * src[m, :] (fp32 -> fp32 / bf16): stochastic-depth mask on the gradient entering a branch (autograd of vit.py:132-139). */
int whmr_scale_rows_cast(const float* src, const float* scale, void* dst, int M, int C, int out_bf16, void* stream);

/* Tail of one regressor stage (whmr.py:142-209 after the skinning) in two launches: dense joint regression over the mesh (9 extra rows, or 33 with
 * J_regressor for smpl_joints45; B x R workgroups), then per image the 54 -> 49 joint gather, markers, and -- when `state` is given -- theta / kp_2d / kp_2d_w / cam_t / focal
 * (whmr.py:142-173,190) plus -- when `xc_next` is given -- the next stage's input state [bbox_info | rotmat | shape | cam] (whmr.py:105,119). */
struct whmr_stage_tail {
    const float* verts; const float* posed_joints; const float* regd /* internal */; float* joints49; float* smpl_joints45; float* markers; int32_t R;
    const float* state; int64_t state_stride; const float* aa; const float* Tz; const float* bbox_h; const float* center; const float* orig_shape;
    float focal0, res_w, res_h; float* theta; float* kp2d; float* kp2d_w; float* cam_t; float* focal;
    const float* bbox_info; const float* rotmat; float* xc_next; int64_t ld_next; int32_t F_next;
};
int whmr_smpl_stage_tail(const struct whmr_smpl_model* m, const struct whmr_stage_tail* t, int B, float* scratch /* >= B*33*3 floats */, void* stream);

/* Pose-corrective blend shapes + skinning in ONE launch (verts.py:46-53, lbs.py:67-77): replaces the [B,207] x [207,20670] whmr_gemm_f32 into a [B,20670]
 * buffer + whmr_smpl_skin.  posedirs_tiled: [108][208][192] (posedirs re-tiled per 64-vertex chunk, k-major, zero padded); pose_feat [B,207] and
 * A [B,24,12] from whmr_smpl_pose_chain; verts [B,6890,3].  Same vertices, bit for bit. */
int whmr_smpl_blend_skin(const struct whmr_smpl_model* m, const float* posedirs_tiled, const float* betas, long beta_stride, const float* pose_feat,
                         const float* A, int B, float* verts, void* stream);
/* The same launch with the pose-corrective offsets (verts.py:51-53) on split-bf16 operands: hi.hi + lo.hi + hi.lo on v_mfma_f32_32x32x16_bf16, fp32
 * accumulate (39 MFMAs of 32 cycles instead of 104 exact-f32 MFMAs of 64: the offsets phase was bound by the f32 matrix rate); shape blend and
 * skinning unchanged (exact f32).  posedirs_x3: [108][13][2][2][192][8] bf16 = per 64-vertex chunk, K step of 16, half (8 k), plane (hi, lo), column:
 * 8 consecutive k.  Vertices within ~3e-7 of whmr_smpl_blend_skin (the offsets are cm corrections of metre-scale coordinates); the module uses it in
 * the bf16 / bf16x3 numerics, the exact form in fp32 and in training. */
int whmr_smpl_blend_skin_x3(const struct whmr_smpl_model* m, const void* posedirs_x3, const float* betas, long beta_stride, const float* pose_feat,
                            const float* A, int B, float* verts, void* stream);
/* tools: route four 64-bit 100 MHz phase stamps of workgroup 0 of whmr_smpl_blend_skin into buf[2..9] (16 uint32 device words; null = off). */
int whmr_smpl_blend_skin_stamps(uint32_t* buf);
/* whmr_smpl_stage_tail with the joint regression as a CSR gather (reg_*: CSR of the first t->R rows of [J_regressor_extra ; J_regressor]): ONE launch,
 * one workgroup per image -- the regressors are > 99 % zeros; replaces the dense B x R-workgroup regression + the tail launch. */
int whmr_smpl_stage_tail_csr(const struct whmr_smpl_model* m, const struct whmr_stage_tail* t, const int32_t* reg_ptr, const int32_t* reg_col,
                             const float* reg_val, int B, void* stream);

/* ---- training-step ground truth: IUV rasteriser = pytorch3d MeshRasterizer(faces_per_pixel 1, blur 0) + HardFlatShader over
 * TexturesVertex as utils/renderer.py:296-446 (IUV_Renderer.verts2iuvimg) uses it from core/trainer.py:442-464.  verts [B, Vsrc, 3]; vmap [V]
 * int64 or null (DensePose vertex duplication); faces [F, 3] int32; tex [V, 3] = (I/24, U, V); cam [B, 3] = (s, tx, ty); K of the orig_h x orig_w
 * image (renderer.py:362-380); scratch scr [B, V, 3] fp32 + zbuf [B, H, W] uint64; out [B, 3, H, W] fp32 (background 0); face_out [B, H, W] int32 or null. */
int whmr_iuv_rasterize(const float* verts, int B, int Vsrc, const int64_t* vmap, int V, const int32_t* faces, int F, const float* tex,
                       const float* cam, float fx, float fy, float px, float py, float focal, int orig_h, int orig_w, int H, int W, float* scr,
                       void* zbuf, float* out, int32_t* face_out, void* stream);

/* Dense-correspondence losses of the AUX supervision, fused: core/trainer.py:255-298 (body_uv_losses, has_iuv = None) on the targets
 * utils/iuvmap.py:67-110 (iuv_img2map, uv_rois = None) derives from the rendered IUV image (core/trainer.py:464-482).  y [B*H*W, ld]: the IUV head's
 * channels-last logits (predict_u 25 | predict_v 25 | predict_uv_index 25 | predict_ann_index 15; models/iuv_predictor.py:71-91), bf16 (y_bf16,
 * ld even) or fp32; iuv [B, 3, H, W] fp32 = (I/24, U, V) with element strides sb, sc, sh, sw.  losses[4] = (loss_U, loss_V, loss_IndexUV,
 * loss_segAnn), U / V already scaled by point_weight (LOSS.POINT_REGRESSION_WEIGHTS) / B.  partial: >= ceil(B*H*W/128)*4 floats of scratch.
 * Deterministic (fixed-order sums). */
int whmr_iuv_losses(const void* y, int y_bf16, long ld, const float* iuv, long sb, long sc, long sh, long sw, int B, int H, int W,
                    float point_weight, float* partial, float* losses, void* stream);
/* Backward of the above: dy [B*H*W, ldg] (y's dtype; columns 90 .. ldg-1 are zeroed: the padded operand ConvNHWCFn's weight gradient reads) =
 * d(g[0] loss_U + g[1] loss_V + g[2] loss_IndexUV + g[3] loss_segAnn) / dy, g [4] on the device. */
int whmr_iuv_losses_bwd(const void* y, int y_bf16, long ld, const float* iuv, long sb, long sc, long sh, long sw, int B, int H, int W,
                        float point_weight, const float* g, void* dy, long ldg, void* stream);

/* ---- attainable ceilings of the box the benchmark runs on (csrc/ceilings.hip; SURVEY 8(d) "datasheet numbers AND measure on the box"); replaces no
 * reference code -- bench.py calls them after its timed region and reports them as roofline.attainable / sclk_mhz_observed.
 * whmr_mfma_ceiling: `blocks` workgroups x 4 waves, each `iters` x 8 register-fed v_mfma_f32_16x16x32_bf16 on random operands
 *   (flop = blocks * 4 * iters * 8 * 16384); stats[0] / stats[1] = s_memrealtime (100 MHz) / s_memtime (shader clock) ticks of one wave over its loop.
 * whmr_hbm_copy: streaming copy of `bytes` (multiple of 16) -- moves 2 * bytes through HBM.
 * whmr_clock_probe_begin (side stream) / _end (observed stream): one wave samples both counters at its start and when _end raises the flag;
 *   state = 6 x uint64, zeroed by the caller: [0] flag, [1..2] realtime / shader clock at start, [3..4] at end, [5] 1 = left through limit_seconds. */
int whmr_mfma_ceiling(int blocks, int iters, float* sink, unsigned long long* stats, void* stream);
int whmr_hbm_copy(const void* src, void* dst, long bytes, void* stream);
int whmr_clock_probe_begin(unsigned long long* state, double limit_seconds, void* stream);
int whmr_clock_probe_end(unsigned long long* state, void* stream);

/* Debug aid: `blocks` workgroups of 128 threads fill `lds_bytes` (512 .. 65536) of LDS with a pattern, idle `spins` x 64 clocks and check it.
 * report (4 x uint32, zeroed by the caller): [0] += dwords found changed, [1] = index + 1 of one of them, [2] = value found, [3] = value expected.
 * Detects a co-resident kernel whose LDS-DMA lands after its workgroup has gone (tools/r6_coresidency_probe.py). */
int whmr_debug_lds_canary(int blocks, int lds_bytes, int spins, unsigned* report, void* stream);
/* Debug aid: table[i] = i * 2654435761u (rows x ld uint32, written by the caller); every thread re-reads its column of all rows `reps` times with plain
 * global loads and compares (report as above). */
int whmr_debug_global_canary(const unsigned* table, int rows, int ld, int reps, unsigned* report, void* stream);
/* Debug aid: every lane runs one multiply-add chain as v_pk_fma_f32 and as two v_fma_f32 and compares the bits; report[0] / [1] += lanes whose low /
 * high half differs; report[2] / [3]: the same for the op_sel:[0,1,0] form (4 x uint32, zeroed by the caller). */
int whmr_debug_pkfma_canary(int blocks, int iters, unsigned* report, void* stream);
/* Debug aid: a bare v_mfma_f32_32x32x16_bf16 stream (`blocks` workgroups of 4 waves, 4 MFMAs per wave and iteration) to run beside a canary. */
int whmr_debug_mfma32_stream(int blocks, int iters, float* sink, void* stream);

#ifdef __cplusplus
}
#endif
#endif
