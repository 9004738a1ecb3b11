// Per-CU load-rate probe: how many bytes per second does ONE CU pull from L2 through each path when its 8 waves issue nothing else?
//   mode 0  global_load_lds_dwordx4 (LDS-DMA, 1 KiB per wave instruction) into a ring of LDS slots
//   mode 1  global_load_dwordx4 into registers (no LDS write)
//   mode 2  global_load_dwordx4 into registers + ds_write_b128 of the previous batch (register-staged copy to LDS)
// Every workgroup streams over the same `span` bytes (L2 / MALL resident after the first pass), 16 B per lane, `depth` loads in flight per wave.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/dma_rate.hip -o tools/lab/dma_rate && tools/lab/dma_rate [blocks]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void probe(const char* __restrict__ src, long span, int iters, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // wave w of block b walks pieces of 1 KiB: piece index p -> byte (p * 8 + w) * 1024 (the 8 waves of a block read 8 KiB rows), offset by the block
    const char* base = src + wave * 1024 + lane * 16;
    long off = (((long)blockIdx.x * 37) % (span / 8192)) * 8192;
    uint32_t acc = 0;
    uint4 r[DEPTH];
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) r[j] = make_uint4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 2) {                                                        // previous batch -> LDS
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) *(uint4*)(smem + ((j * 8 + wave) * 1024 + lane * 16)) = r[j];
        }
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) {
            const char* p = base + off;
            off += 8192;
            if (off + 8192 > span) off = 0;
            if (MODE == 0) __builtin_amdgcn_global_load_lds((gbl_void_t*)p, (lds_void_t*)(smem + (j * 8 + wave) * 1024), 16, 0, 0);
            else r[j] = *(const uint4*)p;
        }
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else {
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) acc ^= r[j].x ^ r[j].w;
        }
    }
    if (MODE == 0) { __syncthreads(); acc = ((uint32_t*)smem)[tid]; }
    if (acc == 0x12345678u) out[0] = (float)acc;
}

// Streaming probe: continuous issue (one piece out, vmcnt(DEPTH - 1)), contiguous 1-KiB pieces, every block its own region (`share` = 1) or `share`
// blocks per region (the column tiles of a GEMM row panel: first toucher misses, the others hit the line in flight), fresh data (span >> caches).
template <int DEPTH, int AUX = 0>
__global__ __launch_bounds__(512) void probe_stream(const char* __restrict__ src, long region, int share, int iters, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const char* base = src + (size_t)(blockIdx.x / share) * region + wave * 1024 + lane * 16;
    long off = 0;
    for (int it = 0; it < iters; ++it) {
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(base + off), (lds_void_t*)(smem + ((it % DEPTH) * 8 + wave) * 1024), 16, 0, AUX);
        off += 8192;
        if (off + 8192 > region) off = 0;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (((uint32_t*)smem)[tid] == 0x12345678u) out[0] = 1.f;
}

template <int DEPTH, int AUX = 0>
static void run_stream(const char* src, long region, int share, int blocks, float* out) {
    auto k = probe_stream<DEPTH, AUX>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    const int iters = (int)(region / 8192);                                  // one pass over the region: every byte fresh
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 128 * 1024, 0, src, region, share, iters, out);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)iters * 8192;
    printf("stream LDS-DMA aux %2d depth %2d (%3d KB in flight per CU) blocks %3d share %2d region %3ld MB: %7.1f us  %6.1f GB/s per CU  %5.2f TB/s unique\n", AUX, DEPTH, DEPTH * 8,
           blocks, share, region >> 20, ms * 1e3, bytes / (ms * 1e-3) / 1e9, bytes * (blocks / share) / (ms * 1e-3) / 1e12);
}

// Scalar-prefetch probe: does pulling the lines of a piece into the XCD's L2 through the SCALAR cache path (8 s_load_dword, one per 128-B line, PF pieces
// ahead) lift the per-CU rate of the LDS-DMA that follows on fresh data?  (the 57 GB/s per-CU ceiling sits in front of L2 on the vector path)
template <int DEPTH, int PF>
__global__ __launch_bounds__(512) void probe_spf(const char* __restrict__ src, long region, int iters, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* wbase = src + (size_t)blockIdx.x * region + wave * 1024;          // wave-uniform
    long off = 0;
    for (int it = 0; it < iters; ++it) {
        if (PF > 0) {
            long po = off + (long)PF * 8192;
            while (po + 8192 > region) po -= (region / 8192) * 8192;
            if (po < 0) po = 0;
            const char* pp = wbase + po;
            // results are never read; they land asynchronously, so they go to fixed high SGPRs nothing else in this kernel uses
            asm volatile("s_load_dword s88, %0, 0x0\n\ts_load_dword s89, %0, 0x80\n\ts_load_dword s90, %0, 0x100\n\ts_load_dword s91, %0, 0x180\n\t"
                         "s_load_dword s92, %0, 0x200\n\ts_load_dword s93, %0, 0x280\n\ts_load_dword s94, %0, 0x300\n\ts_load_dword s95, %0, 0x380"
                         :: "s"(pp) : "memory", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95");
        }
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(wbase + off + lane * 16), (lds_void_t*)(smem + ((it % DEPTH) * 8 + wave) * 1024), 16, 0, 0);
        off += 8192;
        if (off + 8192 > region) off = 0;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (((uint32_t*)smem)[tid] == 0x12345678u) out[0] = 1.f;
}

template <int DEPTH, int PF>
static void run_spf(const char* src, long region, int blocks, float* out) {
    auto k = probe_spf<DEPTH, PF>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    const int iters = (int)(region / 8192);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 128 * 1024, 0, src, region, iters, out);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)iters * 8192;
    printf("fresh stream, LDS-DMA depth %2d, scalar prefetch %2d pieces ahead, blocks %3d region %3ld MB: %7.1f us  %6.1f GB/s per CU  %5.2f TB/s chip\n", DEPTH, PF, blocks,
           region >> 20, ms * 1e3, bytes / (ms * 1e-3) / 1e9, bytes * blocks / (ms * 1e-3) / 1e12);
}

// TN pattern: a block streams K rows of a [K, ld] bf16 matrix, 512 B (256 columns at its tile offset) per row, 32 rows per step, as 1-KiB pieces of two
// rows each (lanes 0-31 row r, lanes 32-63 row r + 1), optionally with gemm_tn.hip's chunk swizzle; `tiles` blocks share the same rows (column tiles).
//   SWZ 0/1; rows_wrap: K rows before wrapping (small = L2 resident, large = streaming)
template <int SWZ, int DEPTH>
__global__ __launch_bounds__(512) void probe_tn(const char* __restrict__ src, int ld_bytes, int rows_wrap, int tiles, int iters, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x % tiles, grp = blockIdx.x / tiles;
    const char* base = src + (size_t)grp * rows_wrap * ld_bytes + tile * 512;
    int k = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) {
            // piece (wave, j): rows k + 2 (wave + 8 j) + {0, 1}
            const int row = k + 2 * (wave + 8 * j) + (lane >> 5);
            const int pos = lane & 31;
            const int chunk = SWZ ? (pos ^ (4 * (row & 3))) : pos;
            const char* p = base + (size_t)row * ld_bytes + chunk * 16;
            __builtin_amdgcn_global_load_lds((gbl_void_t*)p, (lds_void_t*)(smem + (j * 8 + wave) * 1024), 16, 0, 0);
        }
        k += 16 * DEPTH;
        if (k + 16 * DEPTH > rows_wrap) k = 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (((uint32_t*)smem)[tid] == 0x12345678u) out[0] = 1.f;
}

template <int SWZ, int DEPTH>
static void run_tn(const char* name, const char* src, int ld_bytes, int rows_wrap, int tiles, int blocks, float* out) {
    auto k = probe_tn<SWZ, DEPTH>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    const int iters = 4000 / DEPTH;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 96 * 1024, 0, src, ld_bytes, rows_wrap, tiles, 50, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 96 * 1024, 0, src, ld_bytes, rows_wrap, tiles, iters, out);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)iters * DEPTH * 8192;
    printf("%-30s swz %d depth %2d ld %5d B rows %6d tiles %2d blocks %3d: %7.1f us  %6.1f GB/s per CU\n", name, SWZ, DEPTH, ld_bytes, rows_wrap, tiles, blocks, ms * 1e3,
           bytes / (ms * 1e-3) / 1e9);
}


// GEMM-pattern probe (round 5): every workgroup streams TWO operands like a 256 x 256 tile of the blocked GEMM -- per half K tile
// 16 KiB of an "A" panel and 16 KiB of a "W" panel, 4 pieces per wave, three half tiles (96 KiB) in flight, one s_barrier per half tile, every
// workgroup at the same K offset -- and the panels are SHARED the way the GEMM's tile map shares them:
//   MAP 0  same-XCD sharing = the kernel's xcd_remap: block b sits on XCD b % 8 with local index li = b / 8; A panel = (xcd, li / tiles_n) is read by
//          the tiles_n column tiles of a row panel (share 9 for qkv, 12 for fc1), W panel = li % tiles_n by the 32 / tiles_n row panels of the XCD
//          (share 3.5 / 2.7) -- and by the other seven XCDs through their own L2s;
//   MAP 1  the same sharing factors with the sharers on DIFFERENT XCDs (logical tile id = b: dispatch order);
//   MAP 2  no sharing: every workgroup its own A and W panels.
// `klen` = bytes of one panel (256 rows x K x 2 B); the GEMM's own K = 768 gives 393 KB, longer panels show the steady state.
template <int MAP>
__global__ __launch_bounds__(512) void probe_gemm(const char* __restrict__ a_src, const char* __restrict__ w_src, long klen, int tiles_n, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x, xcd = b & 7, li = b >> 3;
    long a_panel, w_panel;
    if (MAP == 0) { a_panel = (long)xcd * 8 + li / tiles_n; w_panel = li % tiles_n; }
    else if (MAP == 1) { a_panel = b / tiles_n; w_panel = b % tiles_n; }
    else { a_panel = b; w_panel = b; }
    const char* ab = a_src + a_panel * klen + wave * 1024 + lane * 16;      // a half tile of a panel = 16 contiguous KiB (the blocked layout)
    const char* wb = w_src + w_panel * klen + wave * 1024 + lane * 16;
    const int H = (int)(klen / 16384);
    auto stage = [&](int h) {
        char* slot = smem + (h & 3) * 32768;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(ab + (long)h * 16384), (lds_void_t*)(slot + wave * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(ab + (long)h * 16384 + 8192), (lds_void_t*)(slot + 8192 + wave * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(wb + (long)h * 16384), (lds_void_t*)(slot + 16384 + wave * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(wb + (long)h * 16384 + 8192), (lds_void_t*)(slot + 24576 + wave * 1024), 16, 0, 0);
    };
    stage(0); stage(1); stage(2);
    for (int h = 0; h < H; ++h) {
        if (h + 3 < H) { stage(h + 3); asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
        else if (h + 2 < H) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (h + 1 < H) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    __syncthreads();
    if (((uint32_t*)smem)[tid] == 0x12345678u) out[0] = 1.f;
}

template <int MAP>
static double run_gemm(const char* a_src, const char* w_src, long klen, int tiles_n, int blocks, float* out, const char* what) {
    auto k = probe_gemm<MAP>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 128 * 1024, 0, a_src, w_src, klen, tiles_n, out);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = 2.0 * klen;                                         // per workgroup
    const char* maps[3] = {"same-XCD sharers (xcd_remap)", "cross-XCD sharers (dispatch order)", "no sharing"};
    printf("GEMM pattern %-36s tiles_n %2d panel %5ld KB blocks %3d %-22s: %7.1f us  %6.1f GB/s per CU  %5.3f us per 32-KB half tile\n", maps[MAP], tiles_n, klen >> 10, blocks, what,
           ms * 1e3, bytes / (ms * 1e-3) / 1e9, ms * 1e3 / (klen / 16384));
    return ms;
}

template <int MODE, int DEPTH>
static void run(const char* name, const char* src, long span, int blocks, float* out) {
    auto k = probe<MODE, DEPTH>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    const int iters = 4000 / DEPTH;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 96 * 1024, 0, src, span, 50, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 96 * 1024, 0, src, span, iters, out);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)iters * DEPTH * 8192;                        // per block
    printf("%-44s depth %2d  blocks %3d  span %4ld MB: %7.1f us  %6.1f GB/s per CU  %6.2f TB/s chip\n", name, DEPTH, blocks, span >> 20, ms * 1e3,
           bytes / (ms * 1e-3) / 1e9, bytes * blocks / (ms * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 256;
    const long span = (argc > 2 ? atol(argv[2]) : 8) << 20;
    char* src; float* out;
    CK(hipMalloc(&src, span + 65536)); CK(hipMalloc(&out, 64));
    CK(hipMemset(src, 1, span + 65536));
    if (argc > 3 && argv[3][0] == 'a') {                         // cache-policy bits of the LDS-DMA (aux operand) on fresh data, one CU group and the whole chip
        for (int rep = 0; rep < 2; ++rep)
            for (int nb : {24, 252}) {
                const long region = ((span / nb) >> 13) << 13;
                const long r = region > (32l << 20) ? (32l << 20) : region;
                run_stream<8, 0>(src, r, 1, nb, out);  CK(hipMemset((void*)src, rep + 2, span));
                run_stream<8, 1>(src, r, 1, nb, out);  CK(hipMemset((void*)src, rep + 3, span));
                run_stream<8, 2>(src, r, 1, nb, out);  CK(hipMemset((void*)src, rep + 4, span));
                run_stream<8, 3>(src, r, 1, nb, out);  CK(hipMemset((void*)src, rep + 5, span));
                run_stream<8, 16>(src, r, 1, nb, out); CK(hipMemset((void*)src, rep + 6, span));
                run_stream<8, 17>(src, r, 1, nb, out); CK(hipMemset((void*)src, rep + 7, span));
                run_stream<8, 18>(src, r, 1, nb, out); CK(hipMemset((void*)src, rep + 8, span));
                run_stream<8, 19>(src, r, 1, nb, out); CK(hipMemset((void*)src, rep + 9, span));
            }
        return 0;
    }
    if (argc > 3 && argv[3][0] == 'x') {                         // GEMM-pattern sharing probe: same-XCD vs cross-XCD vs unshared, short (K = 768) and long panels
        char* flush;
        const long fl = 768l << 20;
        CK(hipMalloc(&flush, fl));
        for (int rep = 0; rep < 2; ++rep)
            for (long klen : {393216l, 6291456l}) {                 // 256 rows x 768 x 2 B (the GEMM's own panels) and 16 x that (steady state)
                if (256 * klen * 2 > span) { printf("span too small for panel %ld\n", klen); continue; }
                const char* a = src; const char* w = src + 256 * klen;
                for (int tn : {9, 12}) {
                    CK(hipMemset(flush, rep + 1, fl));               // push the operands out of L2 and the Infinity Cache: "fresh"
                    run_gemm<0>(a, w, klen, tn, 256, out, "fresh");
                    run_gemm<0>(a, w, klen, tn, 256, out, "again (MALL / L2 warm)");
                    CK(hipMemset(flush, rep + 2, fl));
                    run_gemm<1>(a, w, klen, tn, 256, out, "fresh");
                    run_gemm<1>(a, w, klen, tn, 256, out, "again (MALL / L2 warm)");
                }
                CK(hipMemset(flush, rep + 3, fl));
                run_gemm<2>(a, w, klen, 1, 256, out, "fresh");
                run_gemm<2>(a, w, klen, 1, 256, out, "again (MALL / L2 warm)");
                run_gemm<0>(a, w, klen, 9, 32, out, "32 blocks, warm");   // 4 CUs per XCD: is the shared rate a per-CU or a per-XCD limit?
                run_gemm<0>(a, w, klen, 9, 128, out, "128 blocks, warm");
            }
        return 0;
    }
    if (argc > 3 && argv[3][0] == 'p') {                         // scalar prefetch in front of the LDS-DMA, fresh data
        for (int rep = 0; rep < 2; ++rep)
            for (int nb : {24, 96, 252}) {
                const long region = ((span / nb) >> 13) << 13;
                const long r = region > (32l << 20) ? (32l << 20) : region;
                run_spf<8, 0>(src, r, nb, out);  CK(hipMemset((void*)src, rep + 2, span));
                run_spf<8, 8>(src, r, nb, out);  CK(hipMemset((void*)src, rep + 3, span));
                run_spf<8, 24>(src, r, nb, out); CK(hipMemset((void*)src, rep + 4, span));
                run_spf<8, 64>(src, r, nb, out); CK(hipMemset((void*)src, rep + 5, span));
            }
        return 0;
    }
    if (argc > 3 && argv[3][0] == 's') {                         // streaming sweep: per-CU rate on fresh data against depth, active CUs and sharing
        for (int rep = 0; rep < 2; ++rep)
            for (int share : {1, 3, 12})
                for (int nb : {24, 96, 252}) {
                    const long region = ((span / (nb / share)) >> 13) << 13;
                    const long r = region > (48l << 20) ? (48l << 20) : region;
                    run_stream<4>(src, r, share, nb, out);
                    run_stream<8>(src, r, share, nb, out);
                    run_stream<16>(src, r, share, nb, out);
                    CK(hipMemset((void*)src, rep + 2, span));                // evict: the next pass reads fresh lines again
                }
        return 0;
    }
    if (argc > 3) {                                              // TN pattern sweep
        for (int rep = 0; rep < 2; ++rep)
            for (int ld : {1536, 4608, 6144, 2048}) {
                const int tiles = ld / 512;
                const int groups = blocks / tiles;                   // row groups: each streams its own rows
                for (int rows : {256, 8192}) {
                    if ((long)groups * rows * ld > span) continue;
                    run_tn<1, 4>("LDS-DMA, TN pattern", src, ld, rows, tiles, groups * tiles, out);
                    run_tn<0, 4>("LDS-DMA, TN pattern", src, ld, rows, tiles, groups * tiles, out);
                    run_tn<1, 8>("LDS-DMA, TN pattern", src, ld, rows, tiles, groups * tiles, out);
                }
            }
        return 0;
    }
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 4>("global_load_lds_dwordx4 (LDS-DMA)", src, span, blocks, out);
        run<0, 8>("global_load_lds_dwordx4 (LDS-DMA)", src, span, blocks, out);
        run<1, 4>("global_load_dwordx4 -> registers", src, span, blocks, out);
        run<1, 8>("global_load_dwordx4 -> registers", src, span, blocks, out);
        run<2, 4>("global_load_dwordx4 -> registers -> ds_write", src, span, blocks, out);
        run<2, 8>("global_load_dwordx4 -> registers -> ds_write", src, span, blocks, out);
    }
    return 0;
}
