// Attention core on the BLOCKED token layout, round-5 form:  O = softmax(scale Q K^T) V  per (image, head), bf16 operands, fp32 accumulate
// (vit.py:102-111; the qkv / proj Linears around it are whmr_gemm_blk launches).  Replaces attention_bf16_chunk_kernel<NKT, true> for the
// ViT shapes (d = 64, 64 < N <= 256); same entry point (whmr_attention_blk), same operand layouts.
//
// What bounded the old kernel (profiles/r04_vit224_gemm_pmc.txt: matrix pipes 14.6 % busy, waves parked 32 % of their cycles; 26.8 us at
// N = 196, batch 64 against ~13 us of Q / K / V / O traffic): one workgroup per (image, head) loads ALL its operands through registers, stores them
// to LDS (V transposed with 4-byte stores), and only then computes; the two co-resident workgroups of a CU start together, so their memory and
// compute phases coincide instead of covering each other, 768 workgroups leave a half-empty second round, and 32-row tiles pad N = 196 to 224.
//
// This kernel:
//   * PERSISTENT: min(items, CUs) workgroups, workgroup w walks items w, w + G, ... (768 items = 3 per CU at batch 64 x 12 heads).  While item i
//     is computed, the K / V tiles of item i + 1 stream into the OTHER half of LDS by LDS-DMA (global_load_lds: no registers, no LDS store
//     instructions, no transposing pass), issued by LOADER waves of their own (up to three behind the NQT compute waves), and so does its Q
//     tile (one Q image, refilled as soon as every wave has read its two fragments) -- the memory system always has one item per CU in flight,
//     the compute waves issue no loads at all, and their only exposed waits per item are the two barriers.
//   * 16-row tiles on v_mfma_f32_16x16x32_bf16: NQT = ceil(N / 16) query tiles = waves (13 at N = 196: 6 % padding instead of 31 %), one tile per
//     wave, so the whole score row of a query (NQT x 4 registers) stays in registers: ONE softmax pass, no online rescaling, no LDS for P.
//   * K image in LDS = [d chunk 8][key KP][8 d] (16-B pieces, key-contiguous -- the blocked layout's own order, copied piece by piece):
//     the S^T = K Q^T fragments are conflict-free ds_read_b128.  V image = the same with VP = 32 ceil(NQT / 2) + 8 rows per chunk and is read
//     TRANSPOSED by ds_read_b64_tr_b16 (a 16-lane group reads a [4 keys][16 d] block and each lane receives one d column of it): the A
//     operand of O^T = V^T P^T without a transposed copy.  VP = 8 (mod 16) puts the two d-chunks a 32-lane half touches 128 B apart (mod 256).
//   * S^T = K Q^T ("swapped"): D[key 4g + r][query c] -- a lane owns ONE query and 4 keys per key tile, so max / sum are in-lane plus two
//     cross-group exchanges per item, and the exponentials of key tiles 2j, 2j + 1 ARE the B fragment (k-slot e of lane group g <-> key
//     32 j + 16 (e >> 2) + 4 g + (e & 3)) of the j-th P V step; the V fragment is read in the same key order (two transposing reads, 16 keys apart).
//   * O^T accumulators: lane = query, 4 consecutive d per d tile; pairs of lane groups exchange halves (v_permlane16_swap) and every lane
//     stores one whole 16-B piece of the blocked output.  Rows past N of the last query tile recompute row N - 1 (clamped loads) and store the
//     same bytes to the same place: no exec-masked store, so the count of stores a wave has in flight at the item boundary is a constant.
#include <type_traits>
#include "common.h"

#define LOG2E 1.4426950408889634f

typedef short att_s4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) att_s4_t att_lds_s4_t;
typedef __attribute__((address_space(3))) void att_lds_void_t;
typedef const __attribute__((address_space(1))) void att_gbl_void_t;

template <int NQT>
struct att16_cfg {
    static constexpr int NK2 = (NQT + 1) / 2;            // P V steps of 32 keys
    static constexpr int KP = NQT * 16;                  // key rows per d chunk of the K image
    static constexpr int VP = NK2 * 32 + 8;              // ... of the V image (= 8 mod 16: see the header)
    static constexpr int KBYTES = 8 * KP * 16, VBYTES = 8 * VP * 16, BUF = KBYTES + VBYTES, QBYTES = KBYTES, QOFF = 2 * BUF, LDS = 2 * BUF + QBYTES;
    static constexpr int KINSTR = 8 * KP / 64, VINSTR = 8 * VP / 64, QINSTR = KINSTR;      // LDS-DMA wave instructions (1 KiB each) per item
    static constexpr int NL = (16 - NQT) < 3 ? (16 - NQT) : 3;            // loader waves behind the NQT compute waves (<= 1024 threads)
    static_assert(NL >= 1, "at most 15 query tiles");
    static_assert((8 * KP) % 64 == 0 && (8 * VP) % 64 == 0, "whole wave instructions");
};

// LDS reads and the Q loads are inline asm: hipcc cannot tell the K / V images apart from the LDS-DMA of the NEXT item that is in flight while this item
// is computed, and would drain vmcnt to zero in front of every C++ LDS read (and in front of the back edge for loads that cross it).  The waits
// below carry the registers as in-out operands, so no consumer can be scheduled above them.
template <int OFF> __device__ __forceinline__ bf16x8_t att_read128(uint32_t addr) {
    bf16x8_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int OFF> __device__ __forceinline__ uint2 att_read_tr(uint32_t addr) {
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
// LDS operations return in order: with CNT younger reads in flight, lgkmcnt(CNT) says the tied (older) ones have landed.  Operations the
// compiler issues on its own (ds_bpermute of the softmax exchanges, scalar loads) only add to the count: the wait gets stricter, never wrong.
template <int CNT> __device__ __forceinline__ void att_wait_lds(bf16x8_t (&f)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "n"(CNT));
}
template <int CNT> __device__ __forceinline__ void att_wait_tr(uint2 (&f)[8]) {
    asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "n"(CNT));
}
// compile-time loop: the body receives std::integral_constant<int, I>, so tile indices can be folded into the LDS instructions' immediate offsets
// (ONE address register per image instead of one per tile: as run-time offsets they cost ~30 VGPRs and, in the split-bf16 kernel, spills)
template <int I, int N, class F> __device__ __forceinline__ void att_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        att_static_for<I + 1, N>(f);
    }
}
#define ATT_IC(name, ic) constexpr int name = decltype(ic)::value

// lab instrumentation (tools/attn_stamps.py; never defined in the product build): s_memtime per wave and item of workgroup ATT16_STAMP_WG at the phase
// boundaries -> a device buffer set by whmr_debug_att16_stamps
#ifdef ATT16_STAMPS
__device__ unsigned long long* g_att16_stamps;
extern "C" int whmr_debug_att16_stamps(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_att16_stamps), &buf, sizeof(buf)); }
#define ATT_STAMP(it, slot) do { if (g_att16_stamps && blockIdx.x == 100 && lane == 0 && (it) < 4) g_att16_stamps[((it) * 16 + wave) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ATT_STAMP(it, slot) do { } while (0)
#endif

// byte offset of the 16-B piece (row m, 16-B column unit col8) of a blocked matrix with l8 units per 32-row block
__device__ __forceinline__ size_t att_piece(int m, int col8, int l8) { return (((size_t)(m >> 5) * l8 + col8) * 32 + (m & 31)) * 16; }

template <int NQT>
__global__ __launch_bounds__((NQT + att16_cfg<NQT>::NL) * 64) void attention_blk16_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int items, int N, int H,
                                                                    float scale, int abl) {
    using cfg = att16_cfg<NQT>;
    constexpr int NK2 = cfg::NK2, KP = cfg::KP, VP = cfg::VP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int C = H * 64, ld8 = (3 * C) >> 3, oc8 = C >> 3;
    const char* qkv_b = (const char*)qkv;

    // ---- K / V of one item -> LDS half `buf`: piece p of an image = (chunk p / P, key p % P), LDS address p * 16: every wave instruction writes
    // one contiguous KiB, rows past N re-read row N - 1 (finite data under a zero probability)
    auto issue_dma = [&](int item, int buf) {
        const int b = item / H, h = item - b * H, m_img = b * N;
        char* kb = smem + buf * cfg::BUF;
#pragma unroll 1
        for (int ii = wave - NQT; ii < cfg::KINSTR + cfg::VINSTR; ii += cfg::NL) {
            const bool isv = ii >= cfg::KINSTR;
            const int i0 = isv ? ii - cfg::KINSTR : ii;
            const int p = i0 * 64 + lane;
            const int chunk = isv ? p / VP : p / KP;
            int key = p - chunk * (isv ? VP : KP);
            key = key < N ? key : N - 1;
            const char* src = qkv_b + att_piece(m_img + key, ((isv ? 2 * C : C) >> 3) + h * 8 + chunk, ld8);
            char* dst = kb + (isv ? cfg::KBYTES : 0) + i0 * 1024;                   // wave-uniform; the hardware adds lane * 16
            __builtin_amdgcn_global_load_lds((att_gbl_void_t*)src, (att_lds_void_t*)dst, 16, 0, 0);
        }
    };
    // ---- Q of one item -> the (single) Q image behind the two K / V halves, same [chunk][row KP][8 d] layout as K.  Q is only read at the very top of
    // an item (two fragment reads per wave, then a second barrier), so one image suffices: the loaders refill it right behind that barrier.
    // The compute waves issue NO vector-memory loads at all (as register loads in flight across the item, the Q fragments were either drained
    // by hipcc's own vmcnt(0) right behind the load, or -- as inline asm -- open to register copies before they had landed).
    auto issue_q = [&](int item) {
        const int b = item / H, h = item - b * H, m_img = b * N;
#pragma unroll 1
        for (int ii = wave - NQT; ii < cfg::QINSTR; ii += cfg::NL) {
            const int p = ii * 64 + lane;
            const int chunk = p / KP;
            int row = p - chunk * KP;
            row = row < N ? row : N - 1;
            const char* src = qkv_b + att_piece(m_img + row, h * 8 + chunk, ld8);
            __builtin_amdgcn_global_load_lds((att_gbl_void_t*)src, (att_lds_void_t*)(smem + cfg::QOFF + ii * 1024), 16, 0, 0);
        }
    };

    const float sc = scale * LOG2E;
    auto compute = [&](int item, int buf, const bf16x8_t (&q)[2], int it) {
        const uint32_t kb = (uint32_t)(uintptr_t)(att_lds_void_t*)smem + buf * cfg::BUF;
        const uint32_t vb = kb + cfg::KBYTES;
        // S^T = K Q^T: s[kt][r] = S[key 16 kt + 4 g + r][query c].  Ring of THREE fragment sets (two key tiles each): the reads of the next two
        // pairs are in flight under the MFMAs of this one -- with one set ahead a wave's 4 MFMAs (64 cycles) waited for a whole LDS round trip
        f32x4_t s[NQT];
        const uint32_t ka = kb + (g * KP + c) * 16;
        constexpr int NP = (NQT + 1) / 2;                                // key-tile pairs
        bf16x8_t kf[3][4];                                               // [set][tile 0 half 0 | tile 0 half 1 | tile 1 half 0 | tile 1 half 1]
        auto kread = [&](bf16x8_t (&f)[4], auto kt_) {
            ATT_IC(kt, kt_);
            f[0] = att_read128<kt * 256>(ka);
            f[1] = att_read128<4 * KP * 16 + kt * 256>(ka);
            if constexpr (kt + 1 < NQT) {
                f[2] = att_read128<kt * 256 + 256>(ka);
                f[3] = att_read128<4 * KP * 16 + kt * 256 + 256>(ka);
            } else {
                f[2] = f[0]; f[3] = f[1];
            }
        };
        kread(kf[0], std::integral_constant<int, 0>{});
        if constexpr (NP > 1) kread(kf[1], std::integral_constant<int, 2>{});
        att_static_for<0, NP>([&](auto pr_) {
            ATT_IC(pr, pr_);
            constexpr int kt = 2 * pr, cur = pr % 3;
            if constexpr (pr + 2 < NP) kread(kf[(pr + 2) % 3], std::integral_constant<int, kt + 4>{});
            constexpr int c1 = pr + 1 < NP ? (2 * (pr + 1) + 1 < NQT ? 4 : 2) : 0, c2 = pr + 2 < NP ? (2 * (pr + 2) + 1 < NQT ? 4 : 2) : 0;
            att_wait_lds<c1 + c2>(kf[cur]);                              // reads issued after this pair's
            s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[cur][0], q[0], f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            if constexpr (kt + 1 < NQT) s[kt + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[cur][2], q[0], f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[cur][1], q[1], s[kt], 0, 0, 0);
            if constexpr (kt + 1 < NQT) s[kt + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[cur][3], q[1], s[kt + 1], 0, 0, 0);
        });
        ATT_STAMP(it, 2);
        // softmax over the keys of query c.  Four independent chains (the register index r) for the maximum and the sum instead of one chain of 4 NQT
        // dependent operations; the exchanges across the four lane groups are v_permlane16_swap / v_permlane32_swap of two copies (VALU, no LDS
        // round trip): swapping the odd rows (upper half) of one copy with the even rows (lower half) of the other leaves the two partners' values side by side.
        // Only the last key tile holds padding.
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if ((NQT - 1) * 16 + 4 * g + r >= N) s[NQT - 1][r] = -INFINITY;
        f32x4_t m4 = s[0];
#pragma unroll
        for (int kt = 1; kt < NQT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) m4[r] = fmaxf(m4[r], s[kt][r]);
        float mx = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
        {
            const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
            const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
        }
        const float mxs = mx * sc;
        // The exponentials are taken LAZILY, two key tiles (one P V step) at a time behind the MFMAs of the previous step: an MFMA executes for 16 cycles
        // after it has issued, the wave's VALU work (4 v_exp + scale / sum / pack per tile) issues meanwhile -- as a phase of its own the softmax kept
        // the matrix pipe idle for ~1 900 cycles per wave and item (stamps), and all waves of a SIMD sit in the same phase behind each barrier.
        f32x4_t sum4 = f32x4_t{0.f, 0.f, 0.f, 0.f};
        uint32_t pk[NQT][2];                                             // P as packed bf16: [kt][keys 4 g + (0, 1) | 4 g + (2, 3)]
        auto expo = [&](auto kt_) {
            ATT_IC(kt, kt_);
            if constexpr (kt < NQT) {
                const f32x4_t t = s[kt] * sc - mxs;
                f32x4_t e;
#pragma unroll
                for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(t[r]);
                sum4 += e;
                pk[kt][0] = pack_bf16x2(e[0], e[1]);
                pk[kt][1] = pack_bf16x2(e[2], e[3]);
            }
        };
        expo(std::integral_constant<int, 0>{});
        expo(std::integral_constant<int, 1>{});
        ATT_STAMP(it, 3);
        // O^T = V^T P^T: o[dt][r] = O[query c][d = 16 dt + 4 g + r]
        f32x4_t o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        // transposing read: lane i of group g supplies the 8 bytes (key kbase + (i >> 2), d = 16 dt + 4 (i & 3) ..) and receives d column i of the block
        const uint32_t va = vb + ((((c & 3) >> 1) * VP + 4 * g + (c >> 2)) * 16 + 8 * (c & 1));
        uint2 vr[2][8];                                                   // two sets: [set][2 dt + (keys 32 j + 4 g .. | + 16)].  (A ring of three = 24 reads
        // in flight is past what the 4-bit lgkmcnt field can express -- lgkmcnt(15) with 16 younger reads measured WRONG results, non-deterministically.)
        auto vread = [&](uint2 (&f)[8], auto j_) {
            ATT_IC(j, j_);
            f[0] = att_read_tr<j * 512>(va);                    f[1] = att_read_tr<j * 512 + 256>(va);
            f[2] = att_read_tr<2 * VP * 16 + j * 512>(va);      f[3] = att_read_tr<2 * VP * 16 + j * 512 + 256>(va);
            f[4] = att_read_tr<4 * VP * 16 + j * 512>(va);      f[5] = att_read_tr<4 * VP * 16 + j * 512 + 256>(va);
            f[6] = att_read_tr<6 * VP * 16 + j * 512>(va);      f[7] = att_read_tr<6 * VP * 16 + j * 512 + 256>(va);
        };
        vread(vr[0], std::integral_constant<int, 0>{});
        att_static_for<0, NK2>([&](auto j_) {
            ATT_IC(j, j_);
            constexpr int cur = j & 1;
            union { bf16x8_t v; uint32_t u[4]; } pf;
            pf.u[0] = pk[2 * j][0];
            pf.u[1] = pk[2 * j][1];
            if constexpr (2 * j + 1 < NQT) { pf.u[2] = pk[2 * j + 1][0]; pf.u[3] = pk[2 * j + 1][1]; }
            else { pf.u[2] = pf.u[3] = 0u; }
            if constexpr (j + 1 < NK2) {
                vread(vr[cur ^ 1], std::integral_constant<int, j + 1>{});
                att_wait_tr<8>(vr[cur]);
            } else {
                att_wait_tr<0>(vr[cur]);
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                union { bf16x8_t v; uint2 h[2]; } vf;
                vf.h[0] = vr[cur][2 * dt];
                vf.h[1] = vr[cur][2 * dt + 1];
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf.v, pf.v, o[dt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);                           // the next step's probabilities: VALU work under the MFMAs just issued
            expo(std::integral_constant<int, 2 * j + 2>{});
            expo(std::integral_constant<int, 2 * j + 3>{});
            __builtin_amdgcn_sched_barrier(0);
        });
        float sum = (sum4[0] + sum4[1]) + (sum4[2] + sum4[3]);
        {
            const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(sum), __float_as_uint(sum), false, false);
            sum = __uint_as_float(a[0]) + __uint_as_float(a[1]);
            const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(sum), __float_as_uint(sum), false, false);
            sum = __uint_as_float(b[0]) + __uint_as_float(b[1]);
        }
        const float inv = __builtin_amdgcn_rcpf(sum);                    // 1 ulp; the result is rounded to bf16
        ATT_STAMP(it, 4);
        // store: lane (g, c) holds d = 16 dt + 4 g + (0..3) = half of the 16-B piece 2 dt + (g >> 1).  Even groups complete the piece of dt = 2 t
        // with their odd neighbour's half, odd groups the piece of dt = 2 t + 1 with their even neighbour's (v_permlane16_swap exchanges the odd
        // rows of its first operand with the even rows of its second)
        const int b = item / H, h = item - b * H;
        int qr = 16 * wave + c;
        qr = qr < N ? qr : N - 1;
        char* orow = (char*)out;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x4_t oa = o[2 * t] * inv, ob = o[2 * t + 1] * inv;
            const uint32_t xa0 = pack_bf16x2(oa[0], oa[1]), xa1 = pack_bf16x2(oa[2], oa[3]);
            const uint32_t yb0 = pack_bf16x2(ob[0], ob[1]), yb1 = pack_bf16x2(ob[2], ob[3]);
            const auto r0 = __builtin_amdgcn_permlane16_swap(xa0, yb0, false, false);
            const auto r1 = __builtin_amdgcn_permlane16_swap(xa1, yb1, false, false);
            const int dt = 2 * t + (g & 1);
            *(uint4*)(orow + att_piece(b * N + qr, h * 8 + 2 * dt + (g >> 1), oc8)) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
        }
    };

    int item = blockIdx.x, buf = 0;
    if (item >= items) return;
    // abl (whmr_attention_set_variant bits 1-2, timing probes with WRONG results): 2 = no operand traffic after the first item (every item is
    // computed on the first one's K / V / Q), 4 = no arithmetic (operand traffic, barriers and the stores of the first item only)
    if (wave >= NQT) {
        // ---- LOADER waves (NL of them, every NL-th piece each): all LDS-DMA pieces of an item.  Issuing them blocks for as long as the memory pipeline
        // takes to accept ~80 KiB per CU with every CU bursting at once (stamps: 3 300 - 6 100 cycles when the compute waves issued their own
        // shares behind the barrier -- nothing else ran meanwhile); on waves of their own that costs nothing.  Two barriers per item, like the
        // compute waves: A = the item's operands are in LDS and the other K / V half is free, B = every wave holds its Q fragments, the Q image is free.
        issue_dma(item, 0);
        issue_q(item);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (;;) {
            __builtin_amdgcn_s_barrier();                  // A
            __builtin_amdgcn_s_barrier();                  // B
            const int next = item + gridDim.x;
            if (next >= items) break;
            if (!(abl & 2)) {
                issue_q(next);
                issue_dma(next, buf ^ 1);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            item = next;
            buf ^= 1;
        }
        return;
    }
    const uint32_t qa = (uint32_t)(uintptr_t)(att_lds_void_t*)smem + cfg::QOFF + (g * KP + 16 * wave + c) * 16;
    for (int it = 0;; ++it) {
        __builtin_amdgcn_s_barrier();                      // A
        ATT_STAMP(it, 0);
        bf16x8_t q[2];
        q[0] = att_read128<0>(qa);
        q[1] = att_read128<4 * KP * 16>(qa);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q[0]), "+v"(q[1]));
        __builtin_amdgcn_s_barrier();                      // B
        const int next = item + gridDim.x;
        const bool more = next < items;
        ATT_STAMP(it, 1);
        if (!(abl & 4) || item == (int)blockIdx.x) compute(item, (abl & 2) ? 0 : buf, q, it);
        ATT_STAMP(it, 5);
        if (!more) break;
        item = next;
        buf ^= 1;
    }
}

// =====================================================================================================================================
// The same kernel for the "bf16x3" (split-bf16) numerics: every operand a hi / lo bf16 pair, every product three MFMAs (hi.hi + hi.lo + lo.hi,
// fp32 accumulate) -- replaces attention_x3_blk_kernel (attention_x3.hip: one workgroup per (image, head) and CU, load -> stage -> compute ->
// store in series, 3 rounds of 768 workgroups: 62 us at N = 196, batch 64) for N <= 208.
// LDS holds ONE set of images (K, V, Q as hi + lo pairs = 158 KiB at N = 196), so the prefetch is staged by operand instead of by item:
//   A  K(i), Q(i) in LDS          -> Q fragments to registers, S^T = K Q^T (18 -> 6 MFMAs per key tile)
//   B  everybody done with K, Q   -> the loaders request K(i + 1), Q(i + 1); softmax, P split into hi / lo
//   C  V(i) in LDS                -> O^T = V^T P^T
//   D  everybody done with V      -> the loaders request V(i + 1); store O as a hi / lo pair
// so K / Q of the next item stream in under the softmax and P V, V under S^T and the softmax.  Every loader issues the SAME number of pieces per
// stage (the surplus re-requests the stage's last piece): its counted vmcnt waits are then compile-time constants.
// V image: VP = KP + 8 rows per chunk (= 8 mod 16, see the header); for an odd number of key tiles the last P V step's second half (zero
// probabilities) re-reads the first half's rows instead of rows the image does not have.
template <int NQT>
struct att16x3_cfg {
    static constexpr int NK2 = (NQT + 1) / 2, KP = NQT * 16, VP = KP + 8;
    static constexpr int KB = 8 * KP * 16, VB = 8 * VP * 16;
    static constexpr int K_OFF = 0, V_OFF = 2 * KB, Q_OFF = 2 * KB + 2 * VB, LDS = 4 * KB + 2 * VB;
    static constexpr int KI = 8 * KP / 64, VI = 8 * VP / 64;                 // wave instructions per image
    static constexpr int NL = (16 - NQT) < 3 ? (16 - NQT) : 3;
    static constexpr int TOT_KQ = 4 * KI, TOT_V = 2 * VI;
    static constexpr int CNT_KQ = (TOT_KQ + NL - 1) / NL, CNT_V = (TOT_V + NL - 1) / NL;      // pieces per loader and stage
    static_assert(NL >= 1 && LDS <= 160 * 1024 && CNT_KQ <= 63 && CNT_V <= 63, "");
};

template <int CNT> __device__ __forceinline__ void att_wait_tr4(uint2 (&f)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "n"(CNT));
}

template <int NQT>
__global__ __launch_bounds__((NQT + att16x3_cfg<NQT>::NL) * 64) void attention_blk16_x3_kernel(const bf16_t* __restrict__ qkv_hi, const bf16_t* __restrict__ qkv_lo,
                                                                                            bf16_t* __restrict__ out_hi, bf16_t* __restrict__ out_lo,
                                                                                            int items, int N, int H, float scale, int stagger) {
    using cfg = att16x3_cfg<NQT>;
    constexpr int NK2 = cfg::NK2, KP = cfg::KP, VP = cfg::VP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int C = H * 64, ld8 = (3 * C) >> 3, oc8 = C >> 3;
    int item = blockIdx.x;
    if (item >= items) return;
    // The workgroups of a persistent launch run their items in step: all 256 CUs enter the same phase together.  At 13 query tiles (16 waves)
    // the package then drops its clock for the WHOLE forward (measured; why exactly is a hypothesis) (2126 -> 2046 MHz, profiles/r05_power_clock_attention_stagger.txt),
    // which costs the GEMMs more than this kernel saves.  Starting the four quarters of every XCD's CUs 1 us apart keeps the clock; `stagger`
    // = half-microseconds between quarters (0 where the clock does not react: <= 12 query tiles).
    for (int i = ((blockIdx.x >> 3) & 3) * stagger; i > 0; --i) __builtin_amdgcn_s_sleep(16);

    if (wave >= NQT) {
        // ---- loaders
        const int l = wave - NQT;
        // stage K + Q: instruction index x in [0, TOT_KQ): image x / KI = K hi | K lo | Q hi | Q lo, instruction x % KI of it
        auto issue_kq = [&](int it) {
            const int b = it / H, h = it - b * H, m_img = b * N;
#pragma unroll 1
            for (int i = 0; i < cfg::CNT_KQ; ++i) {
                int x = l + i * cfg::NL;
                x = x < cfg::TOT_KQ ? x : cfg::TOT_KQ - 1;
                const int img = x / cfg::KI, ii = x - img * cfg::KI;
                const int p = ii * 64 + lane;
                const int chunk = p / KP;
                int row = p - chunk * KP;
                row = row < N ? row : N - 1;
                const char* base = (const char*)((img & 1) ? qkv_lo : qkv_hi);
                const char* src = base + att_piece(m_img + row, ((img < 2 ? C : 0) >> 3) + h * 8 + chunk, ld8);
                char* dst = smem + (img < 2 ? cfg::K_OFF + img * cfg::KB : cfg::Q_OFF + (img - 2) * cfg::KB) + ii * 1024;
                __builtin_amdgcn_global_load_lds((att_gbl_void_t*)src, (att_lds_void_t*)dst, 16, 0, 0);
            }
        };
        auto issue_v = [&](int it) {
            const int b = it / H, h = it - b * H, m_img = b * N;
#pragma unroll 1
            for (int i = 0; i < cfg::CNT_V; ++i) {
                int x = l + i * cfg::NL;
                x = x < cfg::TOT_V ? x : cfg::TOT_V - 1;
                const int img = x / cfg::VI, ii = x - img * cfg::VI;
                const int p = ii * 64 + lane;
                const int chunk = p / VP;
                int row = p - chunk * VP;
                row = row < N ? row : N - 1;
                const char* base = (const char*)(img ? qkv_lo : qkv_hi);
                const char* src = base + att_piece(m_img + row, ((2 * C) >> 3) + h * 8 + chunk, ld8);
                char* dst = smem + cfg::V_OFF + img * cfg::VB + ii * 1024;
                __builtin_amdgcn_global_load_lds((att_gbl_void_t*)src, (att_lds_void_t*)dst, 16, 0, 0);
            }
        };
        issue_kq(item);
        issue_v(item);
        for (;;) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(cfg::CNT_V) : "memory");       // K, Q of `item` have landed (its V may still fly)
            __builtin_amdgcn_s_barrier();                                            // A
            __builtin_amdgcn_s_barrier();                                            // B
            const int next = item + gridDim.x;
            const bool more = next < items;
            if (more) {
                issue_kq(next);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(cfg::CNT_KQ) : "memory");  // V of `item` has landed (K, Q of the next item may fly)
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();                                            // C
            __builtin_amdgcn_s_barrier();                                            // D
            if (!more) break;
            issue_v(next);
            item = next;
        }
        return;
    }

    // ---- compute waves: wave = query tile
    const float sc = scale * LOG2E;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(att_lds_void_t*)smem;
    const uint32_t ka = lds0 + cfg::K_OFF + (g * KP + c) * 16;
    const uint32_t qa = lds0 + cfg::Q_OFF + (g * KP + 16 * wave + c) * 16;
    const uint32_t va = lds0 + cfg::V_OFF + ((((c & 3) >> 1) * VP + 4 * g + (c >> 2)) * 16 + 8 * (c & 1));
    for (;;) {
        __builtin_amdgcn_s_barrier();                                                // A
        bf16x8_t q[4];                                                               // hi half 0 | hi half 1 | lo half 0 | lo half 1
        q[0] = att_read128<0>(qa);
        q[1] = att_read128<4 * KP * 16>(qa);
        q[2] = att_read128<cfg::KB>(qa);
        q[3] = att_read128<cfg::KB + 4 * KP * 16>(qa);
        att_wait_lds<0>(q);
        // S^T = K Q^T with split operands: per key tile and d half k_lo q_hi + k_hi q_lo + k_hi q_hi.  Ring of three sets, one key tile each
        // (4 reads: hi / lo x two halves): two tiles' reads in flight under the 6 MFMAs of this one.
        f32x4_t s[NQT];
        bf16x8_t kf[3][4];                                                           // [set][hi half 0 | hi half 1 | lo half 0 | lo half 1]
        auto kread = [&](bf16x8_t (&f)[4], auto kt_) {
            ATT_IC(kt, kt_);
            f[0] = att_read128<kt * 256>(ka);
            f[1] = att_read128<4 * KP * 16 + kt * 256>(ka);
            f[2] = att_read128<cfg::KB + kt * 256>(ka);
            f[3] = att_read128<cfg::KB + 4 * KP * 16 + kt * 256>(ka);
        };
        // ring depth: three sets (two key tiles of reads in flight) where the registers allow it, two at 13 query tiles (128-register budget of 16 waves)
        constexpr int KR = NQT >= 13 ? 2 : 3;
        kread(kf[0], std::integral_constant<int, 0>{});
        if constexpr (KR == 3 && NQT > 1) kread(kf[1], std::integral_constant<int, 1>{});
        att_static_for<0, NQT>([&](auto kt_) {
            ATT_IC(kt, kt_);
            constexpr int cur = kt % KR, ahead = KR - 1;
            if constexpr (kt + ahead < NQT) kread(kf[(kt + ahead) % KR], std::integral_constant<int, kt + ahead>{});
            constexpr int young = (kt + ahead < NQT ? ahead : (NQT - 1 - kt)) * 4;
            att_wait_lds<young>(kf[cur]);
            f32x4_t a = f32x4_t{0.f, 0.f, 0.f, 0.f};
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[cur][2], q[0], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[cur][3], q[1], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[cur][0], q[2], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[cur][1], q[3], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[cur][0], q[0], a, 0, 0, 0);
            s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[cur][1], q[1], a, 0, 0, 0);
        });
        __builtin_amdgcn_s_barrier();                                                // B: K and Q may be overwritten
        // softmax (as in the bf16 kernel), probabilities split into hi / lo
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if ((NQT - 1) * 16 + 4 * g + r >= N) s[NQT - 1][r] = -INFINITY;
        f32x4_t m4 = s[0];
#pragma unroll
        for (int kt = 1; kt < NQT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) m4[r] = fmaxf(m4[r], s[kt][r]);
        float mx = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
        {
            const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
            const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
        }
        const float mxs = mx * sc;
        // lazy exponentials + hi / lo split, one P V step ahead of its MFMAs (see the bf16 kernel)
        f32x4_t sum4 = f32x4_t{0.f, 0.f, 0.f, 0.f};
        uint32_t ph[NQT][2], pl[NQT][2];
        auto expo = [&](auto kt_) {
            ATT_IC(kt, kt_);
            if constexpr (kt < NQT) {
                const f32x4_t t = s[kt] * sc - mxs;
                f32x4_t e;
#pragma unroll
                for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(t[r]);
                sum4 += e;
                split_bf16x2(e[0], e[1], ph[kt][0], pl[kt][0]);
                split_bf16x2(e[2], e[3], ph[kt][1], pl[kt][1]);
            }
        };
        expo(std::integral_constant<int, 0>{});
        expo(std::integral_constant<int, 1>{});
        __builtin_amdgcn_s_barrier();                                                // C: V is in LDS
        // O^T = V^T P^T with split operands: per (key pair j, d tile) v_lo p_hi + v_hi p_lo + v_hi p_hi.  Ring of three sets, one d tile each
        // (4 transposing reads: hi / lo x the two 4-key groups)
        f32x4_t o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        uint2 vr[3][4];                                                              // [set][hi keys .. | hi keys + 16 | lo keys .. | lo keys + 16]
        auto vread = [&](uint2 (&f)[4], auto u_) {                                   // unit u = 4 j + dt
            ATT_IC(u, u_);
            constexpr int j = u >> 2, dt = u & 3;
            constexpr bool wrap = 2 * j + 1 >= NQT;                                  // the second 16 keys of the last step do not exist: zero probabilities
            constexpr int off = j * 512 + dt * (2 * VP * 16);
            f[0] = att_read_tr<off>(va);
            f[2] = att_read_tr<cfg::VB + off>(va);
            f[1] = att_read_tr<off + (wrap ? 0 : 256)>(va);
            f[3] = att_read_tr<cfg::VB + off + (wrap ? 0 : 256)>(va);
        };
        constexpr int NU = 4 * NK2;
        vread(vr[0], std::integral_constant<int, 0>{});
        vread(vr[1], std::integral_constant<int, 1>{});
        att_static_for<0, NU>([&](auto u_) {
            ATT_IC(u, u_);
            constexpr int j = u >> 2, dt = u & 3, cur = u % 3;
            if constexpr (u + 2 < NU) vread(vr[(u + 2) % 3], std::integral_constant<int, u + 2>{});
            att_wait_tr4<(u + 2 < NU) ? 8 : (u + 1 < NU) ? 4 : 0>(vr[cur]);
            union { bf16x8_t v; uint32_t w[4]; } fh, fl;
            fh.w[0] = ph[2 * j][0]; fh.w[1] = ph[2 * j][1];
            fl.w[0] = pl[2 * j][0]; fl.w[1] = pl[2 * j][1];
            if constexpr (2 * j + 1 < NQT) {
                fh.w[2] = ph[2 * j + 1][0]; fh.w[3] = ph[2 * j + 1][1];
                fl.w[2] = pl[2 * j + 1][0]; fl.w[3] = pl[2 * j + 1][1];
            } else {
                fh.w[2] = fh.w[3] = fl.w[2] = fl.w[3] = 0u;
            }
            union { bf16x8_t v; uint2 h[2]; } vh, vl;
            vh.h[0] = vr[cur][0]; vh.h[1] = vr[cur][1];
            vl.h[0] = vr[cur][2]; vl.h[1] = vr[cur][3];
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vl.v, fh.v, o[dt], 0, 0, 0);
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh.v, fl.v, o[dt], 0, 0, 0);
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh.v, fh.v, o[dt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (dt == 1) expo(std::integral_constant<int, 2 * j + 2>{});             // the next step's probabilities under this step's MFMAs
            if constexpr (dt == 3) expo(std::integral_constant<int, 2 * j + 3>{});
            __builtin_amdgcn_sched_barrier(0);
        });
        float sum = (sum4[0] + sum4[1]) + (sum4[2] + sum4[3]);
        {
            const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(sum), __float_as_uint(sum), false, false);
            sum = __uint_as_float(a[0]) + __uint_as_float(a[1]);
            const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(sum), __float_as_uint(sum), false, false);
            sum = __uint_as_float(b[0]) + __uint_as_float(b[1]);
        }
        const float inv = 1.0f / sum;
        __builtin_amdgcn_s_barrier();                                                // D: V may be overwritten
        // store O as a hi / lo pair (the bf16 kernel's piece assembly, once per part)
        const int b = item / H, h = item - b * H;
        int qr = 16 * wave + c;
        qr = qr < N ? qr : N - 1;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x4_t oa = o[2 * t] * inv, ob = o[2 * t + 1] * inv;
            uint32_t xa0, xa1, yb0, yb1, la0, la1, lb0, lb1;
            split_bf16x2(oa[0], oa[1], xa0, la0);
            split_bf16x2(oa[2], oa[3], xa1, la1);
            split_bf16x2(ob[0], ob[1], yb0, lb0);
            split_bf16x2(ob[2], ob[3], yb1, lb1);
            const int dt = 2 * t + (g & 1);
            const size_t off = att_piece(b * N + qr, h * 8 + 2 * dt + (g >> 1), oc8);
            const auto r0 = __builtin_amdgcn_permlane16_swap(xa0, yb0, false, false);
            const auto r1 = __builtin_amdgcn_permlane16_swap(xa1, yb1, false, false);
            *(uint4*)((char*)out_hi + off) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
            const auto s0 = __builtin_amdgcn_permlane16_swap(la0, lb0, false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(la1, lb1, false, false);
            *(uint4*)((char*)out_lo + off) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        }
        const int next = item + gridDim.x;
        if (next >= items) break;
        item = next;
    }
}

static int att16_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus = n;
    }
    return cus;
}

template <int NQT>
static int att16_launch(const void* qkv, void* out, int B, int N, int H, float scale, hipStream_t st, int abl) {
    using cfg = att16_cfg<NQT>;
    auto kern = attention_blk16_kernel<NQT>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, cfg::LDS);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const int items = B * H;
    const int cus = att16_cus();
    const int grid = items < cus ? items : cus;
    hipLaunchKernelGGL(kern, dim3(grid), dim3((NQT + cfg::NL) * 64), cfg::LDS, st, (const bf16_t*)qkv, (bf16_t*)out, items, N, H, scale, abl);
    WHMR_CHECK_LAUNCH();
    return 0;
}

// called by whmr_attention_blk (attention.hip) for 64 < N <= 256
int whmr_attention_blk16_launch(const void* qkv, void* out, int B, int N, int H, float scale, hipStream_t st, int abl) {
    switch ((N + 15) / 16) {
        case 5: return att16_launch<5>(qkv, out, B, N, H, scale, st, abl);
        case 6: return att16_launch<6>(qkv, out, B, N, H, scale, st, abl);
        case 7: return att16_launch<7>(qkv, out, B, N, H, scale, st, abl);
        case 8: return att16_launch<8>(qkv, out, B, N, H, scale, st, abl);
        case 9: return att16_launch<9>(qkv, out, B, N, H, scale, st, abl);
        case 10: return att16_launch<10>(qkv, out, B, N, H, scale, st, abl);
        case 11: return att16_launch<11>(qkv, out, B, N, H, scale, st, abl);
        case 12: return att16_launch<12>(qkv, out, B, N, H, scale, st, abl);
        case 13: return att16_launch<13>(qkv, out, B, N, H, scale, st, abl);
        case 14: return att16_launch<14>(qkv, out, B, N, H, scale, st, abl);
        case 15: return att16_launch<15>(qkv, out, B, N, H, scale, st, abl);
    }
    return -1;                  // N > 240: 16 query tiles + the loader wave exceed 1024 threads -- the caller keeps the round-2 kernel
}

template <int NQT>
static int att16x3_launch(const void* qh, const void* ql, void* oh, void* ol, int B, int N, int H, float scale, hipStream_t st, int lab) {
    using cfg = att16x3_cfg<NQT>;

    auto kern = attention_blk16_x3_kernel<NQT>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, cfg::LDS);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const int items = B * H;
    const int cus = att16_cus();
    const int grid = items < cus ? items : cus;
    // lab = k + 1 forces k (A/B); by default only a launch that fills the chip is staggered (a few workgroups neither move the clock nor should wait)
    const int stagger = lab ? lab - 1 : ((NQT >= 13 && items >= cus) ? 2 : 0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3((NQT + cfg::NL) * 64), cfg::LDS, st, (const bf16_t*)qh, (const bf16_t*)ql, (bf16_t*)oh, (bf16_t*)ol, items, N, H, scale, stagger);
    WHMR_CHECK_LAUNCH();
    return 0;
}

// called by whmr_attention_blk_x3 (attention_x3.hip) for 64 < N <= 208; -1 = not covered (the caller keeps its own kernel)
int whmr_attention_blk16_x3_launch(const void* qh, const void* ql, void* oh, void* ol, int B, int N, int H, float scale, hipStream_t st, int lab) {
    switch ((N + 15) / 16) {
        case 5: return att16x3_launch<5>(qh, ql, oh, ol, B, N, H, scale, st, lab);
        case 6: return att16x3_launch<6>(qh, ql, oh, ol, B, N, H, scale, st, lab);
        case 7: return att16x3_launch<7>(qh, ql, oh, ol, B, N, H, scale, st, lab);
        case 8: return att16x3_launch<8>(qh, ql, oh, ol, B, N, H, scale, st, lab);
        case 9: return att16x3_launch<9>(qh, ql, oh, ol, B, N, H, scale, st, lab);
        case 10: return att16x3_launch<10>(qh, ql, oh, ol, B, N, H, scale, st, lab);
        case 11: return att16x3_launch<11>(qh, ql, oh, ol, B, N, H, scale, st, lab);
        case 12: return att16x3_launch<12>(qh, ql, oh, ol, B, N, H, scale, st, lab);
        case 13: return att16x3_launch<13>(qh, ql, oh, ol, B, N, H, scale, st, lab);
    }
    return -1;
}
