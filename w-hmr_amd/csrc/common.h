// Shared device helpers for the W-HMR gfx950 kernels (wave64, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;   // raw bfloat16 bits

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // MFMA A/B fragment: 8 bf16 (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#define WHMR_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// fp32 -> bf16, round-to-nearest-even with NaN preserved: the native __bf16 conversion lowers to the gfx950 hardware
// instruction (v_cvt_pk_bf16_f32, two values per issue) -- same rounding rule as torch's float -> bfloat16 cast.
typedef __bf16 bf16x2_native_t __attribute__((ext_vector_type(2)));
typedef float f32x2_native_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const bf16x2_native_t v = __builtin_convertvector((f32x2_native_t){lo, hi}, bf16x2_native_t);
    return __builtin_bit_cast(uint32_t, v);
}

__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// exact (erf) GELU, as torch.nn.GELU() default
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// GELU for the bf16 output path: x * sigmoid(2u), u = sqrt(2/pi) (x + 0.044715 x^3)  (the tanh form, written with one
// exp2 and one rcp: 6 plain VALU + 2 transcendental ops instead of ~50 for libm erff).  |gelu_tanh - gelu_erf| <= 3e-4
// absolute, i.e. < 0.1 ulp of the bf16 result it is rounded to; the fp32 parity path keeps the exact erf form.
__device__ __forceinline__ float gelu_fast(float x) {
    const float x2 = x * x;
    const float t = x * fmaf(-0.1029432f, x2, -2.3022082f);       // -2u*log2(e) = x*(-2.3022082 - 0.1029432 x^2)
    const float e = __builtin_amdgcn_exp2f(t);
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

template <typename T> struct io;
template <> struct io<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct io<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// Bijective XCD-aware block remap (8 XCDs, block b is dispatched to XCD b % 8): gives every XCD a
// contiguous range of logical tile ids so neighbouring tiles share the XCD-private L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
