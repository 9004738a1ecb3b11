// Probe the lane <-> element mapping of ds_read_b64_tr_b16 on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(uint16_t* out, int stride_elems) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int lane = threadIdx.x;
    // each lane supplies the address of 4 contiguous elements: row (lane&15)/4 ... test pattern A: tight 4x16 blocks per 16-lane group
    const int g = lane >> 4, i = lane & 15;
    uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds + 2 * (g * 4 * stride_elems + (i >> 2) * stride_elems + (i & 3) * 4);
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
    out[lane * 4 + 0] = v.x & 0xffff; out[lane * 4 + 1] = v.x >> 16; out[lane * 4 + 2] = v.y & 0xffff; out[lane * 4 + 3] = v.y >> 16;
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    for (int stride : {16, 64}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, stride);
        uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("row stride %d elements: lane -> 4 values (element index = row*stride + col)\n", stride);
        for (int l = 0; l < 64; ++l) { printf("lane %2d: ", l); for (int j = 0; j < 4; ++j) printf("(%d,%d) ", h[l*4+j] / stride, h[l*4+j] % stride); printf("\n"); }
    }
    return 0;
}
