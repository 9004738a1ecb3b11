// MFMA-only power probe: which bf16 MFMA shape sustains the higher rate on random operands when the whole chip issues nothing else?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/mfma_power.hip -o tools/lab/mfma_power && tools/lab/mfma_power
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// NA independent accumulators, NF operand fragments rotated through; `iters` rounds
template <int SHAPE, int NA>
__global__ __launch_bounds__(512) void probe(const bf16x8_t* __restrict__ src, float* __restrict__ out, int iters) {
    bf16x8_t a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = src[(threadIdx.x + 512 * i) & 4095]; b[i] = src[(threadIdx.x + 512 * (i + 4)) & 4095]; }
    float s = 0.f;
    if constexpr (SHAPE == 32) {
        f32x16_t acc[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NA; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else {
        f32x4_t acc[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NA; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) s += acc[i][r];
    }
    if (s == 1234.5f) out[0] = s;
}

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;       // 0 random normal-ish, 1 zeros, 2 small-magnitude random
    std::vector<uint16_t> h(4096 * 8);
    srand(3);
    for (auto& v : h) {
        float x = 0.f;
        for (int k = 0; k < 6; ++k) x += (float)rand() / RAND_MAX - 0.5f;     // ~N(0, 0.7)
        v = mode == 1 ? 0 : f2bf(mode == 2 ? x * 0.02f : x * 1.4f);
    }
    bf16x8_t* d; float* o;
    CK(hipMalloc(&d, h.size() * 2)); CK(hipMalloc(&o, 4));
    CK(hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](auto kern, const char* name, double flop_per_mfma, int na, int iters) {
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, d, o, iters);
        CK(hipDeviceSynchronize());
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0));
            for (int l = 0; l < 5; ++l) hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, d, o, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms / 5 < best ? ms / 5 : best;
        }
        const double flops = 256.0 * 8 * (double)iters * na * flop_per_mfma;
        printf("%-28s %8.1f us  %7.0f TFLOP/s\n", name, best * 1e3, flops / (best * 1e-3) / 1e12);
    };
    printf("operands: %s\n", mode == 0 ? "random ~N(0,1)" : mode == 1 ? "zeros" : "random ~N(0,0.02)");
    run(probe<32, 8>, "32x32x16 bf16, 8 acc", 32.0 * 32 * 16 * 2, 8, 600);
    run(probe<16, 8>, "16x16x32 bf16, 8 acc", 16.0 * 16 * 32 * 2, 8, 1200);
    run(probe<16, 16>, "16x16x32 bf16, 16 acc", 16.0 * 16 * 32 * 2, 16, 600);
    run(probe<32, 4>, "32x32x16 bf16, 4 acc", 32.0 * 32 * 16 * 2, 4, 1200);
    return 0;
}
