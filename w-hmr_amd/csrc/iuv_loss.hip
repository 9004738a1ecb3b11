// Dense-correspondence (IUV) auxiliary losses of the training step, forward and backward, straight from the IUV head's channels-last logits and
// the rendered ground-truth IUV image (gfx950, wave64).
//
// Reference: core/trainer.py:255-298 (body_uv_losses, has_iuv = None) applied at :466-482 to the targets utils/iuvmap.py:67-110 (iuv_img2map,
// uv_rois = None) builds from the rendered image (:464).  Per pixel with ground truth (I / 24, U, V):
//   part = round(24 * I/24); the 25 indicator maps hold a single 1 at channel `part` (none when part is outside 0..24);
//   loss_IndexUV = mean_pixels CE(index logits [25], argmax(Imap) = part, or 0 when there is no indicator)
//   loss_segAnn  = mean_pixels CE(ann logits [15], group(part))           (INDEX2MASK groups, iuvmap.py:97-110)
//   loss_U / V   = sum over the entries with Imap > 0 -- channel `part` of every pixel that has an indicator, the background channel
//                  included -- of smooth_l1(u_pred[part] - U), / batch size, * LOSS.POINT_REGRESSION_WEIGHTS
// The torch formulation (whmr_amd.train.aux_supervision.body_uv_losses) walks the [B, 25, H, W] maps a dozen times in each direction (~2 ms of
// the batch-64 step); here one pass reads the 90 logits of a pixel once: HBM-bound, 180 B (bf16) read per pixel forward, + 256 B written backward.
//
// Layout: y [P = B*H*W, ld] rows of (u 25 | v 25 | index 25 | ann 15) as ConvNHWCFn leaves them (bf16 rows padded to 128 columns, fp32 rows
// dense).  A workgroup stages 128 rows through LDS in the logits' own dtype (coalesced 16-byte loads for the padded bf16 rows; odd row stride in
// 4-byte words -- 47 for bf16, 91 for fp32 -- so the per-thread row walks are conflict-free), one thread owns one pixel and holds its 90 logits
// in registers; bf16 keeps the staging area at 24 KB = 12 waves per CU.  The block sums leave in a fixed order and a second launch adds the
// blocks in a fixed order: deterministic.
#include "common.h"

namespace {

constexpr int IUV_PX = 128, IUV_C = 90;
__constant__ int kAnnOfPart[25] = {0, 1, 1, 2, 3, 4, 5, 6, 7, 6, 7, 8, 9, 8, 9, 10, 11, 10, 11, 12, 13, 12, 13, 14, 14};

struct iuv_args {
    const void* y; long ld;                       // logits, row stride in elements
    const float* iuv; long sb, sc, sh, sw;        // ground-truth image [B, 3, H, W] with element strides (the vitpose crop is a strided view)
    int H, W; long P;
    float w_over_b, inv_p;                        // POINT_REGRESSION_WEIGHTS / B, 1 / P
    float* partial;                               // forward: [blocks][4]
    const float* g;                               // backward: upstream gradient of (loss_U, loss_V, loss_IndexUV, loss_segAnn)
    void* dy; long ldg;                           // backward: [P, ldg], columns >= 90 zeroed
};

// LDS rows in 4-byte words: bf16 = 45 words of two channels (+2: odd stride), fp32 = 90 words (+1)
template <typename T> struct rowfmt;
template <> struct rowfmt<bf16_t> {
    static constexpr int WORDS = 45, LDW = 47;
    static __device__ __forceinline__ void unpack(const uint32_t* row, float* l) {
#pragma unroll
        for (int i = 0; i < 45; ++i) { const uint32_t v = row[i]; l[2 * i] = __uint_as_float(v << 16); l[2 * i + 1] = __uint_as_float(v & 0xffff0000u); }
    }
    static __device__ __forceinline__ float at(const uint32_t* row, int c) {
        const uint32_t v = row[c >> 1];
        return __uint_as_float((c & 1) ? (v & 0xffff0000u) : (v << 16));
    }
    static __device__ __forceinline__ void pack(uint32_t* row, const float* l) {
#pragma unroll
        for (int i = 0; i < 45; ++i) row[i] = pack_bf16x2(l[2 * i], l[2 * i + 1]);
    }
};
template <> struct rowfmt<float> {
    static constexpr int WORDS = 90, LDW = 91;
    static __device__ __forceinline__ void unpack(const uint32_t* row, float* l) {
#pragma unroll
        for (int i = 0; i < 90; ++i) l[i] = __uint_as_float(row[i]);
    }
    static __device__ __forceinline__ float at(const uint32_t* row, int c) { return __uint_as_float(row[c]); }
    static __device__ __forceinline__ void pack(uint32_t* row, const float* l) {
#pragma unroll
        for (int i = 0; i < 90; ++i) row[i] = __float_as_uint(l[i]);
    }
};

template <typename T>
__device__ __forceinline__ void stage_rows(const iuv_args& a, long p0, int rows, uint32_t (*s)[rowfmt<T>::LDW]) {
    constexpr int WORDS = rowfmt<T>::WORDS;
    if constexpr (sizeof(T) == 2) {
        const bf16_t* y = (const bf16_t*)a.y;
        if (!(a.ld & 7) && a.ld >= 96 && !((uintptr_t)y & 15)) {
            // 16-byte loads: 12 chunks of 8 channels cover the 90 (rows are at least 96 wide: ConvNHWCFn pads to 128); 16 lanes per row
            const int ch = threadIdx.x & 15, r0 = threadIdx.x >> 4;
            uint4 v[16];                                        // all 16 loads of a thread are in flight before the first LDS write
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int r = r0 + 8 * k;
                v[k] = (ch < 12 && r < rows) ? *(const uint4*)(y + (p0 + r) * a.ld + ch * 8) : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int r = r0 + 8 * k;
                const uint32_t w[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (ch * 4 + e < WORDS) s[r][ch * 4 + e] = w[e];
            }
            return;
        }
    }
    // 4-byte loads (bf16: ld even and the base 4-byte aligned, checked by the launcher)
    const uint32_t* y = (const uint32_t*)a.y;
    const long ldw = sizeof(T) == 2 ? a.ld / 2 : a.ld;
    for (int i = threadIdx.x; i < rows * WORDS; i += IUV_PX) {
        const int r = i / WORDS, c = i - r * WORDS;
        s[r][c] = y[(p0 + r) * ldw + c];
    }
}

struct iuv_target { int part, cls, ann; float u, v; bool has; };

__device__ __forceinline__ iuv_target load_target(const iuv_args& a, long p) {
    const long hw = (long)a.H * a.W;
    const long b = p / hw, rem = p - b * hw;
    const int h = (int)(rem / a.W), w = (int)(rem - (long)h * a.W);
    const float* px = a.iuv + b * a.sb + h * a.sh + w * a.sw;
    iuv_target t;
    const float pf = rintf(px[0] * 24.0f);        // torch.round: half to even
    t.has = pf >= 0.0f && pf <= 24.0f;
    t.part = t.has ? (int)pf : 0;
    t.cls = t.part;                                // argmax of an all-zero indicator column is 0
    t.ann = kAnnOfPart[t.part];
    t.u = px[a.sc];
    t.v = px[2 * a.sc];
    return t;
}

// log(sum exp) of l[0..N) and the shifted exponentials e[c] = exp(l[c] - max) (v_exp_f32 / v_log_f32: ~1e-6 relative, below the fp32 sum's own error)
template <int N>
__device__ __forceinline__ float log_sum_exp(const float* l, float* e, float& sum) {
    float mx = l[0];
#pragma unroll
    for (int c = 1; c < N; ++c) mx = fmaxf(mx, l[c]);
    sum = 0.f;
#pragma unroll
    for (int c = 0; c < N; ++c) { e[c] = __expf(l[c] - mx); sum += e[c]; }
    return mx + __logf(sum);
}

__device__ __forceinline__ float smooth_l1(float d) { const float ad = fabsf(d); return ad < 1.0f ? 0.5f * d * d : ad - 0.5f; }

template <typename T>
__global__ __launch_bounds__(IUV_PX) void iuv_loss_fwd_kernel(iuv_args a) {
    using F = rowfmt<T>;
    __shared__ uint32_t s[IUV_PX][F::LDW];
    __shared__ float red[IUV_PX / 64][4];
    const long p0 = (long)blockIdx.x * IUV_PX;
    const int rows = (int)((a.P - p0) < IUV_PX ? (a.P - p0) : IUV_PX);
    const iuv_target t = load_target(a, p0 + ((int)threadIdx.x < rows ? threadIdx.x : 0));      // issued ahead of the row loads
    stage_rows<T>(a, p0, rows, s);
    __syncthreads();
    float lu = 0.f, lv = 0.f, li = 0.f, la = 0.f;
    if ((int)threadIdx.x < rows) {
        const uint32_t* row = s[threadIdx.x];
        float l[IUV_C], e[25], sum;
        F::unpack(row, l);
        li = log_sum_exp<25>(l + 50, e, sum) - F::at(row, 50 + t.cls);
        la = log_sum_exp<15>(l + 75, e, sum) - F::at(row, 75 + t.ann);
        if (t.has) {
            lu = smooth_l1(F::at(row, t.part) - t.u);
            lv = smooth_l1(F::at(row, 25 + t.part) - t.v);
        }
    }
    lu = wave_sum(lu); lv = wave_sum(lv); li = wave_sum(li); la = wave_sum(la);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wv][0] = lu; red[wv][1] = lv; red[wv][2] = li; red[wv][3] = la; }
    __syncthreads();
    if (threadIdx.x < 4) a.partial[(long)blockIdx.x * 4 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x];
}

// losses[k] = scale_k * sum over the blocks, fixed order: thread t adds blocks t, t + 256, ...; then a tree
__global__ __launch_bounds__(256) void iuv_loss_final_kernel(const float* partial, int nblk, float w_over_b, float inv_p, float* losses) {
    __shared__ float s[256][4];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int b = threadIdx.x; b < nblk; b += 256)
        for (int k = 0; k < 4; ++k) acc[k] += partial[(long)b * 4 + k];
    for (int k = 0; k < 4; ++k) s[threadIdx.x][k] = acc[k];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
            for (int k = 0; k < 4; ++k) s[threadIdx.x][k] += s[threadIdx.x + o][k];
        __syncthreads();
    }
    if (threadIdx.x < 4) losses[threadIdx.x] = s[0][threadIdx.x] * (threadIdx.x < 2 ? w_over_b : inv_p);
}

template <typename T>
__global__ __launch_bounds__(IUV_PX) void iuv_loss_bwd_kernel(iuv_args a) {
    using F = rowfmt<T>;
    __shared__ uint32_t s[IUV_PX][F::LDW];
    const long p0 = (long)blockIdx.x * IUV_PX;
    const int rows = (int)((a.P - p0) < IUV_PX ? (a.P - p0) : IUV_PX);
    const iuv_target t = load_target(a, p0 + ((int)threadIdx.x < rows ? threadIdx.x : 0));      // issued ahead of the row loads
    stage_rows<T>(a, p0, rows, s);
    __syncthreads();
    if ((int)threadIdx.x < rows) {
        uint32_t* row = s[threadIdx.x];            // the row is rewritten in place with its gradient
        float l[IUV_C], e[25], sum;
        F::unpack(row, l);
        const float gu = a.g[0] * a.w_over_b, gv = a.g[1] * a.w_over_b, gi = a.g[2] * a.inv_p, ga = a.g[3] * a.inv_p;
        const float du = t.has ? __builtin_amdgcn_fmed3f(F::at(row, t.part) - t.u, -1.0f, 1.0f) * gu : 0.f;
        const float dv = t.has ? __builtin_amdgcn_fmed3f(F::at(row, 25 + t.part) - t.v, -1.0f, 1.0f) * gv : 0.f;
        const int hit = t.has ? t.part : -1;
        log_sum_exp<25>(l + 50, e, sum);
        float r = gi / sum;
#pragma unroll
        for (int c = 0; c < 25; ++c) {
            l[c] = c == hit ? du : 0.f;
            l[25 + c] = c == hit ? dv : 0.f;
            l[50 + c] = e[c] * r - (c == t.cls ? gi : 0.f);
        }
        log_sum_exp<15>(l + 75, e, sum);
        r = ga / sum;
#pragma unroll
        for (int c = 0; c < 15; ++c) l[75 + c] = e[c] * r - (c == t.ann ? ga : 0.f);
        F::pack(row, l);
    }
    __syncthreads();
    if constexpr (sizeof(T) == 2) {
        bf16_t* dy = (bf16_t*)a.dy;
        if (a.ldg == 128 && !((uintptr_t)dy & 15)) {          // the padded operand of the head's convolution: 256-byte rows, 16 lanes x 16 bytes each
            const int ch = threadIdx.x & 15;
            for (int r = threadIdx.x >> 4; r < rows; r += IUV_PX / 16) {
                uint32_t w[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = ch * 4 + e < F::WORDS ? s[r][ch * 4 + e] : 0u;
                *(uint4*)(dy + (p0 + r) * 128 + ch * 8) = make_uint4(w[0], w[1], w[2], w[3]);
            }
            return;
        }
    }
    // 4-byte stores (bf16: ldg even, two channels per word)
    uint32_t* dy = (uint32_t*)a.dy;
    const int ldw = (int)(sizeof(T) == 2 ? a.ldg / 2 : a.ldg);
    for (int r = 0; r < rows; ++r)
        for (int c = threadIdx.x; c < ldw; c += IUV_PX) dy[(p0 + r) * ldw + c] = c < F::WORDS ? s[r][c] : 0u;
}

int check(const void* y, int y_bf16, long ld, const float* iuv, int B, int H, int W) {
    if (!y || !iuv || B <= 0 || H <= 0 || W <= 0 || ld < IUV_C) return (int)hipErrorInvalidValue;
    if (y_bf16 && ((ld & 1) || ((uintptr_t)y & 3))) return (int)hipErrorInvalidValue;
    return 0;
}

}  // namespace

// loss_U, loss_V, loss_IndexUV, loss_segAnn -> losses[4]; partial: scratch of >= ceil(B*H*W / 128) * 4 floats.
extern "C" int whmr_iuv_losses(const void* y, int y_bf16, long ld, const float* iuv, long sb, long sc, long sh, long sw, int B, int H, int W,
                               float point_weight, float* partial, float* losses, void* stream) {
    if (int e = check(y, y_bf16, ld, iuv, B, H, W)) return e;
    if (!partial || !losses) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    const long P = (long)B * H * W;
    const int nblk = (int)((P + IUV_PX - 1) / IUV_PX);
    iuv_args a{y, ld, iuv, sb, sc, sh, sw, H, W, P, point_weight / (float)B, 1.0f / (float)P, partial, nullptr, nullptr, 0};
    if (y_bf16) hipLaunchKernelGGL(iuv_loss_fwd_kernel<bf16_t>, dim3(nblk), dim3(IUV_PX), 0, st, a);
    else hipLaunchKernelGGL(iuv_loss_fwd_kernel<float>, dim3(nblk), dim3(IUV_PX), 0, st, a);
    WHMR_CHECK_LAUNCH();
    hipLaunchKernelGGL(iuv_loss_final_kernel, dim3(1), dim3(256), 0, st, partial, nblk, a.w_over_b, a.inv_p, losses);
    WHMR_CHECK_LAUNCH();
    return 0;
}

// d(sum_k g[k] * loss_k) / d y -> dy [B*H*W, ldg] in y's dtype, columns 90 .. ldg-1 zeroed (the padded operand the convolution's backward wants).
extern "C" int whmr_iuv_losses_bwd(const void* y, int y_bf16, long ld, const float* iuv, long sb, long sc, long sh, long sw, int B, int H, int W,
                                   float point_weight, const float* g, void* dy, long ldg, void* stream) {
    if (int e = check(y, y_bf16, ld, iuv, B, H, W)) return e;
    if (!g || !dy || ldg < IUV_C || (y_bf16 && ((ldg & 1) || ((uintptr_t)dy & 3)))) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    const long P = (long)B * H * W;
    const int nblk = (int)((P + IUV_PX - 1) / IUV_PX);
    iuv_args a{y, ld, iuv, sb, sc, sh, sw, H, W, P, point_weight / (float)B, 1.0f / (float)P, nullptr, g, dy, ldg};
    if (y_bf16) hipLaunchKernelGGL(iuv_loss_bwd_kernel<bf16_t>, dim3(nblk), dim3(IUV_PX), 0, st, a);
    else hipLaunchKernelGGL(iuv_loss_bwd_kernel<float>, dim3(nblk), dim3(IUV_PX), 0, st, a);
    WHMR_CHECK_LAUNCH();
    return 0;
}
