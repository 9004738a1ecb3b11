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

// (a, b) -> packed bf16 hi halves and packed bf16 lo halves: x ~= hi + lo with 16 significand bits (the split-bf16 operand pair of the
// "bf16x3" numerics); the subtraction is exact in fp32
__device__ __forceinline__ void split_bf16x2(float a, float b, uint32_t& hi, uint32_t& lo) {
    hi = pack_bf16x2(a, b);
    lo = pack_bf16x2(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
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

// GELU for the bf16x3 (parity-grade) epilogue: x * Phi(x) with erf from Abramowitz & Stegun 7.1.28, erf(t) = 1 - (1 + a1 t + ... + a6 t^6)^-16 for t >= 0
// (|error| <= 3e-7), evaluated in fp32: max |gelu_as(x) - gelu(x)| = 8.7e-7 over [-8, 8] against float64 -- the fp32 erf-based GELU of the reference's
// own framework measures 1.2e-6 on the same grid, so this is as exact as the arithmetic being reproduced.  6 FMAs + 4 squarings + one reciprocal:
// about half the instructions of erff, which made the fc1 epilogue of the bf16x3 kernel VALU-bound.
__device__ __forceinline__ float gelu_as(float x) {
    const float t = fabsf(x) * 0.70710678118654752440f;
    float p = fmaf(0.0000430638f, t, 0.0002765672f);
    p = fmaf(p, t, 0.0001520143f);
    p = fmaf(p, t, 0.0092705272f);
    p = fmaf(p, t, 0.0422820123f);
    p = fmaf(p, t, 0.0705230784f);
    p = fmaf(p, t, 1.0f);
    p = p * p; p = p * p; p = p * p; p = p * p;
    const float e = 1.0f - __builtin_amdgcn_rcpf(p);                 // erf(|x| / sqrt 2)
    return 0.5f * x * (1.0f + copysignf(e, x));
}

// GELU for the bf16 output path: x * Phi~(x), Phi~(x) = 0.5 + s r(s^2), s = clamp(x, -4, 4), r = degree-6 minimax polynomial
// (LP fit of x*(Phi~ - Phi) on [-4, 4] with Phi~(4) = 1 pinned, so the tails are exact: gelu -> x and -> 0).
// |gelu_fast - gelu_erf| <= 1.9e-4 absolute over all x (the tanh form the reference's accelerators use is 4.7e-4 off),
// far below the bf16 rounding of the value it feeds.  All full-rate VALU and written on float2 so hipcc emits
// v_pk_mul_f32 / v_pk_fma_f32: 5.5 instructions per element instead of 5 + 2 quarter-rate transcendentals (exp2, rcp) --
// the fc1 epilogue was transcendental-bound.  The fp32 parity path keeps the exact erf form.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t gelu_fast2(f32x2_t x) {
    f32x2_t s;
    s.x = __builtin_amdgcn_fmed3f(x.x, -4.0f, 4.0f);
    s.y = __builtin_amdgcn_fmed3f(x.y, -4.0f, 4.0f);
    const f32x2_t t = s * s;
    f32x2_t r = t * 2.258814658e-08f + (-1.588823733e-06f);
    r = r * t + 4.776381398e-05f;
    r = r * t + (-8.121867222e-04f);
    r = r * t + 8.763687250e-03f;
    r = r * t + (-6.455440501e-02f);
    r = r * t + 3.978702657e-01f;
    return x * (s * r + 0.5f);
}
__device__ __forceinline__ float gelu_fast(float x) {
    const float s = __builtin_amdgcn_fmed3f(x, -4.0f, 4.0f), t = s * s;
    float r = fmaf(t, 2.258814658e-08f, -1.588823733e-06f);
    r = fmaf(r, t, 4.776381398e-05f);
    r = fmaf(r, t, -8.121867222e-04f);
    r = fmaf(r, t, 8.763687250e-03f);
    r = fmaf(r, t, -6.455440501e-02f);
    r = fmaf(r, t, 3.978702657e-01f);
    return x * fmaf(s, r, 0.5f);
}

// d/dx gelu(x) = Phi(x) + x phi(x) for the training backward's bf16 path: 0.5 + s u(s^2), s = clamp(x, -4.5, 4.5), u = degree-9 least-squares fit
// on Chebyshev nodes of (gelu'(s) - 0.5) / s (an even function).  |error| <= 2.6e-4 absolute in fp32 Horner form over all x (the tails are
// 1.00007 and -7e-5 instead of 1 and 0), a tenth of the bf16 rounding of the gradient it multiplies; full-rate packed VALU like gelu_fast2
// (the erf + exp form is ~40 instructions per element and made the GELU backward VALU-bound).
__device__ __forceinline__ f32x2_t gelu_grad_fast2(f32x2_t x) {
    f32x2_t s;
    s.x = __builtin_amdgcn_fmed3f(x.x, -4.5f, 4.5f);
    s.y = __builtin_amdgcn_fmed3f(x.y, -4.5f, 4.5f);
    const f32x2_t t = s * s;
    f32x2_t r = t * -2.744070681e-11f + 3.034363388e-09f;
    r = r * t + (-1.475886388e-07f);
    r = r * t + 4.182379123e-06f;
    r = r * t + (-7.728450434e-05f);
    r = r * t + 9.887899513e-04f;
    r = r * t + (-9.041428673e-03f);
    r = r * t + 5.918059743e-02f;
    r = r * t + (-2.655834586e-01f);
    r = r * t + 7.978483487e-01f;
    return s * r + 0.5f;
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
