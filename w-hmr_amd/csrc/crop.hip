// Person-crop + normalise: the demo's per-detection cv2.warpAffine + ToTensor + Normalize
// (datasets/data_utils/img_utils.py:53-101,209-242 via demo/tester.py:112-122) for ALL detections of a frame in one launch.
// The frame is read once from HBM (uint8 HWC, as cv2 holds it); every person's patch leaves as normalised fp32 NCHW -- only the
// column range the model consumes (tester.py:151: inp[:, :, :, 32:-32]) is produced when the caller asks for it.
//
// Arithmetic = OpenCV's 8-bit INTER_LINEAR warpAffine (imgwarp.cpp, restated; cv2 is absent from the image, so this is pinned
// against oracle/crop.py only): inverse affine map in fixed point, AB_BITS = 10, coordinates rounded to 1/32 pixel,
// 15-bit bilinear weights (exact multiples of 32 for 1/32 fractions), BORDER_CONSTANT = 0, result (sum + 2^14) >> 15.
// Then torchvision ToTensor (p / 255) and Normalize ((v - mean) / std) in fp32, in that operation order.
#include "common.h"

__global__ __launch_bounds__(256) void crop_normalize_kernel(const uint8_t* __restrict__ frame, int H, int W, long row_stride,
                                                             const double* __restrict__ inv, int patch_w, int patch_h, int x_begin,
                                                             int out_w, float* __restrict__ out, uint8_t* __restrict__ raw,
                                                             float m0, float m1, float m2, float s0, float s1, float s2) {
    const int b = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= out_w * patch_h) return;
    const int y = idx / out_w, x = x_begin + idx % out_w;
    const double* M = inv + b * 6;
    // imgwarp.cpp warpAffine: adelta / bdelta / X0 / Y0 are cvRound'ed (round half to even) products in AB_SCALE units
    const int adelta = (int)__double2ll_rn(M[0] * x * 1024.0), bdelta = (int)__double2ll_rn(M[3] * x * 1024.0);
    // explicit mul / add: an fma contraction of M1*y + M2 could move a tie and break bit-parity with the CPU restatement
    const int X0 = (int)__double2ll_rn(__dadd_rn(__dmul_rn(M[1], (double)y), M[2]) * 1024.0) + 16;
    const int Y0 = (int)__double2ll_rn(__dadd_rn(__dmul_rn(M[4], (double)y), M[5]) * 1024.0) + 16;
    const int X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
    const int sx = X >> 5, sy = Y >> 5, fx = X & 31, fy = Y & 31;
    const int w00 = (32 - fy) * (32 - fx) * 32, w01 = (32 - fy) * fx * 32, w10 = fy * (32 - fx) * 32, w11 = fy * fx * 32;
    int acc[3] = {1 << 14, 1 << 14, 1 << 14};
    auto tap = [&](int yy, int xx, int w) {
        if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W && w) {
            const uint8_t* p = frame + (size_t)yy * row_stride + (size_t)xx * 3;
            acc[0] += w * p[0]; acc[1] += w * p[1]; acc[2] += w * p[2];
        }
    };
    tap(sy, sx, w00); tap(sy, sx + 1, w01); tap(sy + 1, sx, w10); tap(sy + 1, sx + 1, w11);
    const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
    const size_t plane = (size_t)out_w * patch_h;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int v = acc[c] >> 15;
        out[((size_t)b * 3 + c) * plane + idx] = (__fdiv_rn((float)v, 255.0f) - mean[c]) / sd[c];
        if (raw) raw[(((size_t)b * patch_h + y) * out_w + (x - x_begin)) * 3 + c] = (uint8_t)v;
    }
}

// inv_affine: per person the INVERSE 2x3 map (dst pixel -> frame pixel) as 6 doubles, computed on the host like
// cv2.getAffineTransform + invertAffineTransform.  out: [B, 3, patch_h, x_end - x_begin] fp32; raw (nullable): the uint8 patch.
extern "C" int whmr_crop_normalize(const uint8_t* frame, int H, int W, long row_stride, const double* inv_affine, int B, int patch_w,
                                   int patch_h, int x_begin, int x_end, float* out, uint8_t* raw, const float* mean3, const float* std3,
                                   void* stream) {
    const int out_w = x_end - x_begin;
    if (B <= 0 || H <= 0 || W <= 0 || out_w <= 0 || x_begin < 0 || x_end > patch_w || patch_h <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(crop_normalize_kernel, dim3((out_w * patch_h + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, frame, H, W,
                       row_stride, inv_affine, patch_w, patch_h, x_begin, out_w, out, raw, mean3[0], mean3[1], mean3[2], std3[0], std3[1],
                       std3[2]);
    WHMR_CHECK_LAUNCH();
    return 0;
}
