// Attainable ceilings of THIS box, measured next to the benchmark ("take gfx950 datasheet numbers and also measure on the box").
// bench.py runs them after its timed region and reports them beside the nominal peaks:
//   whmr_mfma_ceiling   register-fed v_mfma_f32_16x16x32_bf16 stream on random operands on every SIMD (no LDS, no memory): the dense bf16 matrix
//                       rate the package sustains at the clock its governor grants under that load (nominal 2.5 PF assumes 2.4 GHz);
//   whmr_hbm_copy       16-byte-per-lane streaming copy (read + write): the HBM rate a memory-bound kernel can reach;
//   whmr_clock_probe_*  one wave that samples s_memtime (shader clock) and s_memrealtime (100 MHz) when it starts and when a flag is raised on
//                       another stream -- the average shader clock of the XCD it sits on while the benchmark's steps run beside it.
// Not on the data path: nothing here computes a result the model uses.
#include "common.h"

// stats[0] = s_memrealtime ticks (100 MHz) of wave 0 of block 0 over its loop, stats[1] = s_memtime (shader clocks) over the same span
__global__ __launch_bounds__(256) void mfma_ceiling_kernel(int iters, unsigned seed, float* sink, unsigned long long* stats) {
    const int lane = threadIdx.x & 63;
    // random bf16 operands in [-2, 2): exponent field 0x3f80..0x3fff, random sign and mantissa (switching activity like real data, no inf / nan)
    unsigned s = seed ^ (blockIdx.x * 2654435761u) ^ (threadIdx.x * 40503u);
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (short)(((s >> 16) & 0x807f) | 0x3f80); };
    bf16x8_t a[2], b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) { a[i][e] = rnd(); b[i][e] = rnd(); }
    f32x4_t acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const bool rec = blockIdx.x == 0 && threadIdx.x == 0;
    unsigned long long r0 = 0, c0 = 0;
    if (rec) { r0 = __builtin_amdgcn_s_memrealtime(); c0 = __builtin_amdgcn_s_memtime(); }
    // inline asm: eight INDEPENDENT accumulators in VGPRs, one MFMA each per iteration, nothing else in the loop (the intrinsic form let hipcc move the
    // accumulators between VGPRs and AGPRs inside the loop: ~30 instead of 16 cycles per MFMA)
#define WHMR_CEIL_MFMA(i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[(i) & 1]), "v"(b[((i) >> 1) & 1]))
    for (int it = 0; it < iters; ++it) {
        WHMR_CEIL_MFMA(0); WHMR_CEIL_MFMA(1); WHMR_CEIL_MFMA(2); WHMR_CEIL_MFMA(3);
        WHMR_CEIL_MFMA(4); WHMR_CEIL_MFMA(5); WHMR_CEIL_MFMA(6); WHMR_CEIL_MFMA(7);
    }
#undef WHMR_CEIL_MFMA
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");       // the last results have left the matrix pipe before the VALU reads them
    if (rec) {
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        stats[0] = r1 - r0; stats[1] = c1 - c0;
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (t == 12345.678f) sink[lane] = t;                  // keeps the accumulators live
}

// One launch: `blocks` workgroups of 4 waves (one per SIMD), 8 independent MFMAs per wave and iteration.  flop = blocks * 4 * iters * 8 * 16384.
extern "C" int whmr_mfma_ceiling(int blocks, int iters, float* sink, unsigned long long* stats, void* stream) {
    if (blocks <= 0 || iters <= 0 || !sink || !stats) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(mfma_ceiling_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, 0x9e3779b9u, sink, stats);
    WHMR_CHECK_LAUNCH();
    return 0;
}

__global__ __launch_bounds__(256) void hbm_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long n16) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}

extern "C" int whmr_hbm_copy(const void* src, void* dst, long bytes, void* stream) {
    if (!src || !dst || bytes <= 0 || (bytes & 15)) return (int)hipErrorInvalidValue;
    const long n16 = bytes >> 4;
    long blocks = (n16 + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(hbm_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, (uint4*)dst, n16);
    WHMR_CHECK_LAUNCH();
    return 0;
}

// state[0] = flag (raised by whmr_clock_probe_end), state[1..2] = realtime / shader clock at the start, state[3..4] = at the end, state[5] = 1 when the
// probe left through its time limit instead of the flag.  The wait is BOUNDED (limit_ticks of the 100 MHz counter): a probe that is never released
// ends by itself.
__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long* state, unsigned long long limit_ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    state[1] = r0; state[2] = c0;
    unsigned long long r = r0;
    bool timed_out = false;
    while (__hip_atomic_load(&state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0ull) {
        __builtin_amdgcn_s_sleep(64);
        r = __builtin_amdgcn_s_memrealtime();
        if (r - r0 > limit_ticks) { timed_out = true; break; }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    state[3] = r1; state[4] = c1; state[5] = timed_out ? 1ull : 0ull;
}

__global__ void clock_probe_raise_kernel(unsigned long long* state) {
    if (threadIdx.x == 0) __hip_atomic_store(&state[0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// state: 6 x uint64 in device memory, zeroed by the caller.  begin goes on a SIDE stream, end on the stream whose work is being observed.
extern "C" int whmr_clock_probe_begin(unsigned long long* state, double limit_seconds, void* stream) {
    if (!state || !(limit_seconds > 0.0) || limit_seconds > 5.0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, (unsigned long long)(limit_seconds * 1e8));
    WHMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int whmr_clock_probe_end(unsigned long long* state, void* stream) {
    if (!state) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(clock_probe_raise_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state);
    WHMR_CHECK_LAUNCH();
    return 0;
}

// LDS canary (round 6: a kernel whose late LDS-DMA lands in the LDS of the workgroup that took its place).  Every workgroup fills `words` dwords of
// dynamic LDS with a pattern of its own, idles `spins` x 64 clocks, and checks it.  report[0] += words found changed, report[1] = LDS dword index of one
// of them + 1, report[2] = the value found there, report[3] = the value expected.
__global__ __launch_bounds__(128) void lds_canary_kernel(int words, int spins, unsigned* __restrict__ report) {
    extern __shared__ unsigned canary[];
    const unsigned tag = 0xA5000000u ^ (blockIdx.x * 2654435761u);
    for (int i = threadIdx.x; i < words; i += 128) canary[i] = tag + (unsigned)i;
    __syncthreads();
    for (int s = 0; s < spins; ++s) __builtin_amdgcn_s_sleep(1);
    __syncthreads();
    for (int i = threadIdx.x; i < words; i += 128) {
        const unsigned got = canary[i];
        if (got != tag + (unsigned)i) {
            atomicAdd(&report[0], 1u);
            report[1] = (unsigned)i + 1u; report[2] = got; report[3] = tag + (unsigned)i;
        }
    }
}

extern "C" int whmr_debug_lds_canary(int blocks, int lds_bytes, int spins, unsigned* report, void* stream) {
    if (blocks <= 0 || lds_bytes < 512 || lds_bytes > 64 * 1024 || spins < 0 || spins > (1 << 20) || !report) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(lds_canary_kernel, dim3(blocks), dim3(128), lds_bytes, (hipStream_t)stream, lds_bytes / 4, spins, report);
    WHMR_CHECK_LAUNCH();
    return 0;
}

// Global-read canary: table[i] = i * 2654435761u (written by the caller).  Every thread re-reads `rows` strided rows of `ld` dwords at its own column
// `reps` times and compares; report as above ([1] = dword index + 1, [2] found, [3] expected).  Plain global loads, like a kernel reading constant tables.
__global__ __launch_bounds__(128) void global_canary_kernel(const unsigned* __restrict__ table, int rows, int ld, int reps, unsigned* __restrict__ report) {
    const int v = blockIdx.x * 128 + threadIdx.x;
    if (v >= ld) return;
    for (int r = 0; r < reps; ++r)
        for (int k = 0; k < rows; ++k) {
            const unsigned i = (unsigned)k * (unsigned)ld + (unsigned)v;
            const unsigned got = table[i];
            if (got != i * 2654435761u) {
                atomicAdd(&report[0], 1u);
                report[1] = i + 1u; report[2] = got; report[3] = i * 2654435761u;
            }
        }
}

extern "C" int whmr_debug_global_canary(const unsigned* table, int rows, int ld, int reps, unsigned* report, void* stream) {
    if (!table || rows <= 0 || ld <= 0 || reps <= 0 || !report) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(global_canary_kernel, dim3((ld + 127) / 128), dim3(128), 0, (hipStream_t)stream, table, rows, ld, reps, report);
    WHMR_CHECK_LAUNCH();
    return 0;
}

// Packed-FMA canary: every lane runs the same chain twice, as v_pk_fma_f32 on a register pair and as two v_fma_f32, and compares the bits.
// report[0] += lanes whose LOW half differs, report[1] += lanes whose HIGH half differs; report[2] / [3]: the same for the op_sel:[0,1,0] form.
__global__ __launch_bounds__(128) void pkfma_canary_kernel(int iters, unsigned* __restrict__ report) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const float t = (float)(threadIdx.x + 128 * (blockIdx.x & 15)) * 1e-3f;
    f2 a = {1.0f + t, 0.5f - t}, acc = {0.1f, 0.2f};
    float sx = 0.1f, sy = 0.2f;
    for (int i = 0; i < iters; ++i) {
        f2 b = {0.999f - (float)(i & 255) * 1e-5f, -0.998f + (float)(i & 127) * 1e-5f};
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(sx) : "v"(a.x), "v"(b.x));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(sy) : "v"(a.y), "v"(b.y));
    }
    if (__float_as_uint(acc.x) != __float_as_uint(sx)) atomicAdd(&report[0], 1u);
    if (__float_as_uint(acc.y) != __float_as_uint(sy)) atomicAdd(&report[1], 1u);
    // the same with op_sel:[0,1,0]: BOTH lanes multiply by the HIGH half of the second source (the form hipcc emits for "a pair times one scalar of a pair":
    // smpl_skin_bwd_kernel's shape sum, rows 1 / 5 / 9 -- the rows that went wrong beside the 64-row TN tile)
    f2 acc2 = {0.1f, 0.2f};
    float tx = 0.1f, ty = 0.2f;
    for (int i = 0; i < iters; ++i) {
        f2 b = {0.999f - (float)(i & 255) * 1e-5f, -0.998f + (float)(i & 127) * 1e-5f};
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc2) : "v"(a), "v"(b));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(tx) : "v"(a.x), "v"(b.y));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(ty) : "v"(a.y), "v"(b.y));
    }
    if (__float_as_uint(acc2.x) != __float_as_uint(tx)) atomicAdd(&report[2], 1u);
    if (__float_as_uint(acc2.y) != __float_as_uint(ty)) atomicAdd(&report[3], 1u);
}

// A bare v_mfma_f32_32x32x16_bf16 stream (4 independent accumulators per wave): the companion of the canaries above.
__global__ __launch_bounds__(256) void mfma32_stream_kernel(int iters, float* sink) {
    unsigned s = 0x9e3779b9u ^ (blockIdx.x * 2654435761u) ^ (threadIdx.x * 40503u);
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (short)(((s >> 16) & 0x807f) | 0x3f80); };
    bf16x8_t a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = rnd(); b[e] = rnd(); }
    f32x16_t acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[1]) : "v"(a), "v"(b));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[2]) : "v"(a), "v"(b));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[3]) : "v"(a), "v"(b));
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[i][r];
    if (t == 12345.678f) sink[0] = t;
}

extern "C" int whmr_debug_mfma32_stream(int blocks, int iters, float* sink, void* stream) {
    if (blocks <= 0 || iters <= 0 || !sink) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(mfma32_stream_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, sink);
    WHMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int whmr_debug_pkfma_canary(int blocks, int iters, unsigned* report, void* stream) {
    if (blocks <= 0 || iters <= 0 || !report) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(pkfma_canary_kernel, dim3(blocks), dim3(128), 0, (hipStream_t)stream, iters, report);
    WHMR_CHECK_LAUNCH();
    return 0;
}
