// LDS read-rate probe: how many bytes per clock and CU do the read forms of the GEMM kernels sustain when all 8 waves of a workgroup issue nothing else?
//   mode 0  ds_read_b128, lane * 16 linear (the forward kernels' fragment read)
//   mode 1  ds_read_b64, lane * 8 linear
//   mode 2  ds_read_b64_tr_b16 with the address pattern of gemm_tn.hip's tn_frag_issue (512-B rows, chunk XOR 4 (row & 3))
//   mode 3  ds_read_b64_tr_b16, same rows WITHOUT the swizzle (what the conflicts would cost)
// One workgroup of 512 threads per CU (96 KB of LDS), `iters` rounds of 12 independent reads + one lgkmcnt(0).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/lds_rate.hip -o tools/lab/lds_rate && tools/lab/lds_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef __attribute__((address_space(3))) void lds_void_t;

template <int MODE>
__global__ __launch_bounds__(512) void probe(float* __restrict__ out, int iters, long long* __restrict__ clk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 96 * 1024 / 4; i += 512) ((uint32_t*)smem)[i] = i * 2654435761u;
    __syncthreads();
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_t*)smem;
    uint32_t addr;
    if (MODE == 0) addr = lds0 + wave * 8192 + lane * 16;
    else if (MODE == 1) addr = lds0 + wave * 8192 + lane * 8;
    else {
        const int g = lane >> 4, i = lane & 15;
        const int c0 = (wave & 3) * 64;
        const int col = c0 + 16 * (g & 1) + 4 * (i & 3);
        const int row = 4 * (g >> 1) + (i >> 2);
        const int swz = MODE == 2 ? 4 * (row & 3) : 0;
        addr = lds0 + (wave >> 2) * 16384 + row * 512 + (((col >> 3) ^ swz) << 4) + (col & 7) * 2;
    }
    uint32_t acc = 0;
    const long long t0 = wall_clock64();
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            uint4 r[12];
#pragma unroll
            for (int j = 0; j < 12; ++j)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[j]) : "v"(addr), "n"((j & 7) * 1024) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 12; ++j) acc ^= r[j].x ^ r[j].w;
        } else if (MODE == 1) {
            uint2 r[12];
#pragma unroll
            for (int j = 0; j < 12; ++j)
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[j]) : "v"(addr), "n"((j & 7) * 512) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 12; ++j) acc ^= r[j].x ^ r[j].y;
        } else {
            uint2 r[12];
#pragma unroll
            for (int j = 0; j < 12; ++j)
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r[j]) : "v"(addr), "n"(((j & 1) * 8 + (j >> 1 & 1) * 16) * 512 + (j >> 2) * 64) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 12; ++j) acc ^= r[j].x ^ r[j].y;
        }
    }
    const long long c1 = clock64();
    const long long t1 = wall_clock64();
    if (acc == 0x12345678u) out[0] = (float)acc;
    if (tid == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = t1 - t0; }
}

template <int MODE>
static void run(const char* name, int bytes_per_lane, float* out, long long* clk, int iters) {
    auto k = probe<MODE>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 96 * 1024, 0, out, 100, clk);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 96 * 1024, 0, out, iters, clk);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    long long h[2];
    CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    const double bytes = (double)iters * 12 * 512 * bytes_per_lane;        // per CU
    printf("%-34s %8.1f us  %8.0f GB/s per CU  clock64 %lld -> %.1f B/clk/CU   wall_clock64 %lld ticks (%.1f B/tick)   %.1f clk per wave-instruction\n", name, ms * 1e3,
           bytes / (ms * 1e-3) / 1e9, h[0], bytes / (double)h[0], h[1], bytes / (double)h[1], (double)h[0] / ((double)iters * 12 * 8));
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    float* out; long long* clk;
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&clk, 64));
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("ds_read_b128 linear", 16, out, clk, iters);
        run<1>("ds_read_b64 linear", 8, out, clk, iters);
        run<2>("ds_read_b64_tr_b16 swizzled (TN)", 8, out, clk, iters);
        run<3>("ds_read_b64_tr_b16 unswizzled", 8, out, clk, iters);
    }
    return 0;
}
