// Clock-governor probe (round 5): does the package clock react to the launch PATTERN?  A stream of long "steady" kernels (every SIMD issues bf16 MFMAs
// back to back, like a GEMM main loop) with a short "phased" kernel between them whose 256 workgroups alternate MFMA bursts and idle gaps IN STEP
// (what a persistent kernel with barrier-separated phases does) -- or the same phased kernel with its CU quarters started apart.  The steady kernels
// read their own shader clock (s_memtime cycles per s_memrealtime 100 MHz tick); the table shows what the phased neighbour does to it.
// RESULT (profiles/r05_clock_gov_lab.txt): nothing -- 1.93-1.98 GHz in every row.  MFMA bursts in step are NOT enough to reproduce the clock drop the
// persistent bf16x3 attention kernel caused: that kernel's phases also carry LDS-DMA, LDS reads and the softmax VALU work of 16 waves.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/clock_gov.hip -o tools/lab/clock_gov && tools/lab/clock_gov
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void burst(f32x4_t (&acc)[8], const bf16x8_t (&a)[4], const bf16x8_t (&b)[4], int iters) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
    }
}

// rec[blockIdx] = {shader cycles, 100 MHz ticks} of this workgroup (wave 0)
__global__ __launch_bounds__(512) void steady(const bf16x8_t* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ rec, int iters) {
    bf16x8_t a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = src[(threadIdx.x + 512 * i) & 4095]; b[i] = src[(threadIdx.x + 512 * (i + 4)) & 4095]; }
    f32x4_t acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    burst(acc, a, b, iters);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { rec[2 * blockIdx.x] = c1 - c0; rec[2 * blockIdx.x + 1] = r1 - r0; }
    if (s == 1234.5f) out[0] = s;
}

// `periods` x [barrier | MFMA burst of burst_iters | barrier | idle_units x s_sleep(16) (~0.5 us each)]; the CU quarters of every XCD start
// stagger_units x 0.5 us apart (0 = all workgroups in step)
__global__ __launch_bounds__(512) void phased(const bf16x8_t* __restrict__ src, float* __restrict__ out, int periods, int burst_iters, int idle_units, int stagger_units) {
    bf16x8_t a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = src[(threadIdx.x + 512 * i) & 4095]; b[i] = src[(threadIdx.x + 512 * (i + 4)) & 4095]; }
    f32x4_t acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int i = ((blockIdx.x >> 3) & 3) * stagger_units; i > 0; --i) __builtin_amdgcn_s_sleep(16);
    for (int p = 0; p < periods; ++p) {
        __syncthreads();
        burst(acc, a, b, burst_iters);
        __syncthreads();
        for (int i = 0; i < idle_units; ++i) __builtin_amdgcn_s_sleep(16);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 1234.5f) out[0] = s;
}

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }

int main(int argc, char** argv) {
    std::vector<uint16_t> h(4096 * 8);
    srand(3);
    for (auto& v : h) {
        float x = 0.f;
        for (int k = 0; k < 6; ++k) x += (float)rand() / RAND_MAX - 0.5f;
        v = f2bf(x * 1.4f);
    }
    bf16x8_t* d; float* o; unsigned long long* rec;
    const int REPS = 1500;
    CK(hipMalloc(&d, h.size() * 2)); CK(hipMalloc(&o, 4)); CK(hipMalloc(&rec, (size_t)REPS * 256 * 2 * 8));
    CK(hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    std::vector<unsigned long long> hr((size_t)REPS * 512);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // steady kernel: 2 waves per SIMD x 8 MFMAs x 16 cycles per iteration = 256 cycles: 4000 iterations ~ 0.5 ms at 2 GHz
    const int steady_iters = argc > 1 ? atoi(argv[1]) : 4000;
    auto run = [&](const char* name, int periods, int burst_iters, int idle_units, int stagger) {
        // ~0.8 s of the pattern, clocks averaged over the second half
        CK(hipEventRecord(e0));
        for (int r = 0; r < REPS; ++r) {
            hipLaunchKernelGGL(steady, dim3(256), dim3(512), 0, 0, d, o, rec + (size_t)r * 512, steady_iters);
            if (periods) hipLaunchKernelGGL(phased, dim3(256), dim3(512), 0, 0, d, o, periods, burst_iters, idle_units, stagger);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(hr.data(), rec, hr.size() * 8, hipMemcpyDeviceToHost));
        double cyc = 0, tick = 0;
        for (int r = REPS / 2; r < REPS; ++r)
            for (int w = 0; w < 256; ++w) { cyc += (double)hr[(size_t)r * 512 + 2 * w]; tick += (double)hr[(size_t)r * 512 + 2 * w + 1]; }
        const double mhz = cyc / tick * 100.0, us = tick / (256.0 * (REPS - REPS / 2)) / 100.0;
        // (us = wave 0 of each workgroup; the two waves of a SIMD do not share the pipe evenly, the launch itself lasts about twice as long)
        printf("%-78s steady kernel: %6.0f MHz, wave 0 busy %6.1f us   | steady + phased launch pair %6.1f us\n", name, mhz, us, ms * 1e3 / REPS);
        fflush(stdout);
    };
    // phased kernel: burst_iters x 256 cycles per burst (2 waves per SIMD): 40 iterations ~ 5 us; idle unit ~ 0.5 us
    run("steady kernels only", 0, 0, 0, 0);
    run("+ phased kernel, 3 x [5 us MFMA | 5 us idle], workgroups in step", 3, 40, 10, 0);
    run("+ the same, CU quarters 1 us apart", 3, 40, 10, 2);
    run("+ the same, CU quarters 2.5 us apart", 3, 40, 10, 5);
    run("+ phased kernel, 3 x [5 us MFMA | no idle] (continuous), in step", 3, 40, 0, 0);
    run("+ phased kernel, 6 x [2.5 us MFMA | 2.5 us idle], in step", 6, 20, 5, 0);
    run("+ the same, CU quarters 1 us apart", 6, 20, 5, 2);
    run("+ phased kernel, 3 x [2 us MFMA | 8 us idle], in step", 3, 16, 16, 0);
    run("+ phased kernel, 15 x [1 us MFMA | 1 us idle], in step", 15, 8, 2, 0);
    run("steady kernels only (again)", 0, 0, 0, 0);
    return 0;
}
