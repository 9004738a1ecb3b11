// SMPL call of a regressor stage as THREE launches (models/whmr.py:128-209; math: models/smpl_webuser/lbs.py:27-80): whmr_smpl_pose_chain (smpl_lbs.hip) ->
// whmr_smpl_blend_skin -> whmr_smpl_stage_tail_csr.  The two kernels of this file:
//
//   blend + skin  work item = 64 vertices x 32 images on a 6-wave workgroup: the pose-corrective offsets pose_feature . posedirs on
//            v_mfma_f32_32x32x2_f32 (exact f32, a sequential fma chain over k = the arithmetic of the fp32 GEMM it replaced, so the same bits): wave w
//            owns 32 of the item's 192 (vertex, coordinate) columns, A operand = the 32 images' pose features from LDS, B operand = posedirs rows
//            straight from global memory (re-tiled per 64-vertex chunk, all 104 k-steps in flight); the 32 x 192 offsets go through LDS to a
//            (vertex x image subset) pass that adds the shape blend -> v_posed, in place; then the skinning T = sum_j w_j A_j ALSO runs on the f32 MFMA
//            (round 5: [32 images, 24 joints] . [24 joints, 32 vertices] per transform entry, wave = (vertex tile, output coordinate)) and
//            v = T [v_posed; 1] is applied in the accumulator layout.  posedirs (17 MB) is read once per 32 images.
//   CSR tail      one workgroup per image: the 9 (+24) joint-regressor rows as a CSR gather over the skinned mesh (products through LDS, one thread per
//            (row, coordinate) adds its segment in index order: deterministic), then smpl_stage_tail_image (joint map, markers, theta, kp_2d,
//            kp_2d_w, cam_t, focal, the next stage's input state).
//
// History: both were written as phases 2 / 3 of a ONE-launch SMPL call (persistent grid, two spin grid barriers, coherent sc1 traffic between the
// phases; round 3).  That kernel was bit-identical but slower than the three launches (56 vs 42 us at batch 64: one workgroup per CU pays every
// phase's memory round trips alone) and its barrier relied on co-residency that nothing enforced (two such launches on two streams could
// spin forever) -- removed in round 4 (last version: git show 7a2c1ef:w-hmr_amd/csrc/smpl_fused.hip).  The COH template parameter of
// the shared device code (smpl_dev.h) is what is left of it: false everywhere.
// Floating-point contraction per EXPRESSION (the language rule), not across statements after inlining (hipcc's default "fast"): whether a product
// is fused into an fma then depends on the source expression only, not on the kernel it was inlined into -- the per-phase kernels and the
// one-launch kernel share smpl_dev.h / geometry_dev.h and must produce the same bits.
#pragma clang fp contract(on)
#include "smpl_dev.h"

struct whmr_smpl_call {
    const float* pose9; int64_t pose_stride;        // [B, 216] rows (row stride in floats)
    const float* betas; int64_t beta_stride;        // [B, 10]
    int32_t B, do_gs;
    float* rotmat; float* aa;                       // optional outputs of phase 1
    float* A; float* posed_joints; float* pose_feat;   // [B,24,12], [B,24,3], [B,207]: phase-1 results the later phases read (required)
    float* verts;                                   // [B, 6890, 3]
    const int32_t* reg_ptr; const int32_t* reg_col; const float* reg_val;   // CSR of the [R, 6890] regressor rows (extra rows first, then J_regressor)
    uint32_t* barrier;                              // tools only: 16 uint32 whose words [2..13] receive phase stamps of workgroup 0 (may be null)
    const float* posedirs_tiled;                    // [108][208][192]: posedirs re-tiled per 64-vertex chunk, k-major inside a chunk, zero padded
    const bf16_t* posedirs_x3;                      // [108][13][2][2][192][8] bf16: the same tile as split-bf16 hi / lo planes in MFMA operand order (X3 kernels)
    whmr_stage_tail tail;                           // (unused by the launches of this file)
};

#define FUSED_VT 64          // vertices per phase-2 item (192 posedirs columns = 6 MFMA column tiles, one per wave)
#define FUSED_IG 32          // images per phase-2 item (the M of the 32x32x2 MFMA; smaller batches are zero-padded)
#define FUSED_NT 384         // threads per workgroup: 6 waves
#define FUSED_NNZ 1536       // CSR products staged per pass (phase 3): 4 per thread
#define FUSED_KP 208         // pose-feature depth padded to the MFMA's k step (row 207 is zero)


// LDS of phase 2: sPF [208][32] | sBeta [10][32] | sA [32][289] | sPO [32][192].  An image's 24 x 12 skinning transforms sit 289 floats apart
// (288 + 1): the skinning MFMAs read entry (j, e) of 32 IMAGES at once -- with a stride of 288 = 9 x 32 all of them on one LDS bank.
#define P2_AS (NJ * 12 + 1)
#define P2_PF 0
#define P2_BETA (FUSED_KP * FUSED_IG)
#define P2_A (P2_BETA + 10 * FUSED_IG)
#define P2_PO (P2_A + FUSED_IG * P2_AS)
#define P2_FLOATS (P2_PO + FUSED_IG * 3 * FUSED_VT)

// COH: pose_feat / A were written by an earlier phase of the SAME kernel and the vertices are read by a later one (coherent sc1 accesses);
// false: a separate launch (whmr_smpl_blend_skin) -- plain cached accesses.
// X3 (round 6; inference numerics bf16 / bf16x3): the pose-corrective offsets on v_mfma_f32_32x32x16_bf16 with split-bf16 operands (hi.hi + lo.hi + hi.lo,
// fp32 accumulate) instead of 104 exact-f32 MFMAs of 64 cycles each -- the offsets phase was bound by the f32 matrix rate (6.5 us of the launch's
// 18 us: two SIMDs carry two waves).  The offsets are centimetre corrections of metre-scale coordinates: their 1e-5 relative error is ~3e-7 of a vertex.
// posedirs comes pre-split (hi / lo planes, 8 consecutive k per 16-byte piece), the pose features are split as they are staged.  X3 = false keeps the
// exact-f32 chain (fp32 numerics, training, the bit-identity tests against the five-launch form).
template <bool COH, bool X3 = false>
__device__ __forceinline__ void fused_phase2_item(const whmr_smpl_model& m, const whmr_smpl_call& p, int vb, int b0, float* smem) {
    float* sPF = smem + P2_PF;
    float* sBeta = smem + P2_BETA;
    float* sA = smem + P2_A;
    float* sPO = smem + P2_PO;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, B = p.B;
    const int l31 = lane & 31, hi = lane >> 5;
    const int v0 = vb * FUSED_VT;
    // this item's posedirs tile: 208 x 192 floats, CONTIGUOUS (the [207, 20670] original puts a chunk's k rows 82 KB apart: 208 DRAM pages and TLB
    // entries per wave for 128-B pieces -- measured 26 us per item, 0.65 TB/s); zero padded, so no clamps.  All 104 k-steps of this lane's column
    // are requested FIRST (104 registers), ahead of the staging loads below: the two latencies overlap and are paid once.
    // (separate-launch form only: four 100 MHz stamps of workgroup 0 behind p.barrier, when given -- tools/smpl_timing.py)
    uint64_t* stamps = (!COH && p.barrier && blockIdx.x == 0 && tid == 0) ? (uint64_t*)(p.barrier + 2) : nullptr;
    if (stamps) stamps[0] = wall_clock64();
    constexpr int KS = FUSED_KP / 16;                                             // K steps of the bf16 MFMA (13)
    float pv[X3 ? 1 : FUSED_KP / 2];
    bf16x8_t bx[X3 ? KS : 1][2];                                                  // X3: [k step][hi / lo plane], 8 consecutive k of this lane's column
    if constexpr (X3) {
        const bf16x8_t* pt = (const bf16x8_t*)p.posedirs_x3 + ((size_t)vb * KS * 2 + hi) * 2 * (3 * FUSED_VT) + 32 * wave + l31;
#pragma unroll
        for (int st = 0; st < KS; ++st)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) bx[st][pl] = pt[((size_t)st * 2 * 2 + pl) * (3 * FUSED_VT)];
    } else {
        const float* pt = p.posedirs_tiled + (size_t)vb * (FUSED_KP * 3 * FUSED_VT) + hi * (3 * FUSED_VT) + 32 * wave + l31;
#pragma unroll
        for (int u = 0; u < FUSED_KP / 2; ++u) pv[u] = pt[(size_t)u * (2 * 3 * FUSED_VT)];
    }
    // staging: every coherent (sc1) load of a thread is ISSUED before the first one is used -- a load-use pair per loop iteration would pay the
    // memory round trip once per element (18 + 24 of them)
    {
        constexpr int NPFL = (FUSED_KP * FUSED_IG + FUSED_NT - 1) / FUSED_NT, NAL = (FUSED_IG * NJ * 12 + FUSED_NT - 1) / FUSED_NT;
        float tpf[NPFL], ta[NAL];
#pragma unroll
        for (int i = 0; i < NPFL; ++i) {
            const int e = tid + i * FUSED_NT, k = e / FUSED_IG, bb = e % FUSED_IG;
            tpf[i] = (e < FUSED_KP * FUSED_IG && b0 + bb < B && k < NPF) ? ld_f<COH>(p.pose_feat + (size_t)(b0 + bb) * NPF + k) : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NAL; ++i) {
            const int e = tid + i * FUSED_NT, bb = e / (NJ * 12);
            ta[i] = (e < FUSED_IG * NJ * 12 && b0 + bb < B) ? ld_f<COH>(p.A + (size_t)(b0 + bb) * NJ * 12 + (e % (NJ * 12))) : 0.f;
        }
        for (int e = tid; e < 10 * FUSED_IG; e += FUSED_NT) {
            const int k = e / FUSED_IG, bb = e % FUSED_IG;
            sBeta[e] = (b0 + bb < B) ? p.betas[(size_t)(b0 + bb) * p.beta_stride + k] : 0.f;
        }
        if constexpr (X3) {
            // split-bf16 planes in MFMA operand order: element (k = 16 s + 8 h + j, image) -> sPFx[((s * 2 + h) * 2 + plane) * 32 + image][j]
            bf16_t* sx = (bf16_t*)sPF;
#pragma unroll
            for (int i = 0; i < NPFL; ++i) {
                const int e = tid + i * FUSED_NT;
                if (e < FUSED_KP * FUSED_IG) {
                    const int k = e / FUSED_IG, bb = e % FUSED_IG;
                    uint32_t h2, l2;
                    split_bf16x2(tpf[i], 0.f, h2, l2);
                    const int base = ((((k >> 4) * 2 + ((k >> 3) & 1)) * 2) * FUSED_IG + bb) * 8 + (k & 7);
                    sx[base] = (bf16_t)(h2 & 0xffffu);
                    sx[base + FUSED_IG * 8] = (bf16_t)(l2 & 0xffffu);
                }
            }
        } else {
#pragma unroll
        for (int i = 0; i < NPFL; ++i) { const int e = tid + i * FUSED_NT; if (e < FUSED_KP * FUSED_IG) sPF[e] = tpf[i]; }
        }
#pragma unroll
        for (int i = 0; i < NAL; ++i) { const int e = tid + i * FUSED_NT; if (e < FUSED_IG * NJ * 12) sA[(e / (NJ * 12)) * P2_AS + e % (NJ * 12)] = ta[i]; }
    }
    __syncthreads();
    if (stamps) stamps[1] = wall_clock64();
    // ---- pose-corrective offsets (verts.py:51-53) = pose_feature . posedirs on v_mfma_f32_32x32x2_f32 -- exact f32, a sequential fma chain over k:
    // the instruction the per-phase path's GEMM uses, hence the same bits.  Wave w owns columns 32 w .. 32 w + 31 of the item's 192 for all 32
    // images: A operand = pose features [image = lane & 31][k = k0 + (lane >> 5)] from LDS, B operand = posedirs [k][column] straight from global
    // memory (two 128-B row pieces per step, 13 steps in flight).
    {
        f32x16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if constexpr (X3) {
            // 13 K steps x (hi.hi + lo.hi + hi.lo): 39 bf16 MFMAs of 32 cycles instead of 104 f32 MFMAs of 64; all 26 operand reads first
            const bf16x8_t* sx = (const bf16x8_t*)sPF + hi * 2 * FUSED_IG + l31;
            bf16x8_t ax[KS][2];
#pragma unroll
            for (int st = 0; st < KS; ++st)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) ax[st][pl] = sx[(st * 2 * 2 + pl) * FUSED_IG];
#pragma unroll
            for (int st = 0; st < KS; ++st) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ax[st][0], bx[st][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ax[st][1], bx[st][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ax[st][0], bx[st][1], acc, 0, 0, 0);
            }
        } else {
        constexpr int S = FUSED_KP / 2;
        // the pose-feature operands are read ahead in two halves (52 registers each): fetched one step at a time each ds_read's latency sat on
        // the dependent MFMA chain (130 cycles per step measured, 64 for the instruction itself)
        float av[S / 2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int u = 0; u < S / 2; ++u) av[u] = sPF[(2 * (half * (S / 2) + u) + hi) * FUSED_IG + l31];
#pragma unroll
            for (int u = 0; u < S / 2; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], pv[half * (S / 2) + u], acc, 0, 0, 0);
        }
        }
        // C layout: register r of lane (l31, hi) = image (r & 3) + 8 (r >> 2) + 4 hi, column l31
#pragma unroll
        for (int r = 0; r < 16; ++r) sPO[((r & 3) + 8 * (r >> 2) + 4 * hi) * (3 * FUSED_VT) + 32 * wave + l31] = acc[r];
    }
    __syncthreads();
    if (stamps) stamps[2] = wall_clock64();
    // ---- few images in this group (B = 1 .. 6: the demo, one image per thread subset): the round-3 form -- thread = (vertex, image), shape blend + offset +
    // skinning as a VALU fma chain straight to memory.  The matrix form below always computes a 32-image tile (4.2 us whatever B; this: ~2 us at B = 1).
    // Both are the arithmetic of smpl_skin_vertex: the same bits (tests: three-launch vs five-launch form at B = 1 .. 130).
    if (B - b0 <= FUSED_NT / 64) {
        const int lv = tid & (FUSED_VT - 1), bb = tid >> 6;
        const int v = v0 + lv;
        if (v < NV && b0 + bb < B) {
            const float t0 = m.v_template[3 * v], t1 = m.v_template[3 * v + 1], t2 = m.v_template[3 * v + 2];
            float sd[30], w[NJ];
#pragma unroll
            for (int k = 0; k < 30; ++k) sd[k] = m.shapedirs[(size_t)k * NV + v];
#pragma unroll
            for (int j = 0; j < NJ; ++j) w[j] = m.lbs_weights[(size_t)j * NV + v];
            float acc[3];
            smpl_shape_vertex(t0, t1, t2, sd, sBeta + bb, FUSED_IG, acc);
            const float* po = sPO + bb * (3 * FUSED_VT) + 3 * lv;
            acc[0] += po[0]; acc[1] += po[1]; acc[2] += po[2];
            float A1[NJ * 12];                                                         // (sA rows are not 16-B aligned any more: gather the image's transforms by value)
#pragma unroll
            for (int e = 0; e < NJ * 12; ++e) A1[e] = sA[bb * P2_AS + e];
            smpl_skin_vertex_regs<COH>(w, A1, acc[0], acc[1], acc[2], p.verts + ((size_t)(b0 + bb) * NV + v) * 3);
        }
        __syncthreads();
        if (stamps) stamps[3] = wall_clock64();
        return;
    }
    // ---- shape blend + offset -> v_posed, IN PLACE in sPO: thread = (vertex lv = tid & 63, image subset sub = tid >> 6): images bb = sub, sub + 6, ...
    // The skinning weights of the wave's 32 vertices (the B operand of the skinning MFMAs below) are requested first: they land under this pass.
    const int vt = wave & 1, cc = wave >> 1;                                      // skinning: this wave's vertex tile (32 vertices) and output coordinate
    const int vsk = v0 + vt * 32 + l31;
    float wq[NJ / 2];
    {
        const int vc = vsk < NV ? vsk : NV - 1;
#pragma unroll
        for (int sidx = 0; sidx < NJ / 2; ++sidx) wq[sidx] = m.lbs_weights[(size_t)(2 * sidx + hi) * NV + vc];
    }
    {
        const int lv = tid & (FUSED_VT - 1), sub = tid >> 6;
        const int v = v0 + lv;
        if (v < NV && b0 + sub < B) {
            const float t0 = m.v_template[3 * v], t1 = m.v_template[3 * v + 1], t2 = m.v_template[3 * v + 2];
            float sd[30];
#pragma unroll
            for (int k = 0; k < 30; ++k) sd[k] = m.shapedirs[(size_t)k * NV + v];
            for (int bb = sub; bb < FUSED_IG; bb += FUSED_NT / 64) {
                if (b0 + bb >= B) break;
                float acc[3];
                smpl_shape_vertex(t0, t1, t2, sd, sBeta + bb, FUSED_IG, acc);
                float* po = sPO + bb * (3 * FUSED_VT) + 3 * lv;
                acc[0] += po[0]; acc[1] += po[1]; acc[2] += po[2];
                po[0] = acc[0]; po[1] = acc[1]; po[2] = acc[2];
            }
        }
    }
    __syncthreads();
    // ---- skinning on the matrix pipes (lbs.py:67-77): T = sum_j w_j A_j as [32 images, 24 joints] . [24 joints, 32 vertices] per transform entry,
    // v_mfma_f32_32x32x2_f32 -- exact f32, a sequential fma chain over j starting from 0: the arithmetic of smpl_skin_vertex (smpl_dev.h), so the same
    // bits.  Wave (vt, cc) owns output coordinate cc of 32 vertices x 32 images: the four entries 4 cc .. 4 cc + 3 of T (four accumulator tiles, 12
    // MFMAs each), then v_cc = T[4cc+2] z + T[4cc+1] y + T[4cc] x + T[4cc+3] in the accumulator layout (lane = vertex, register = image).
    // As a VALU loop (thread = vertex x image subset, 288 FMAs per pair fed by 72 broadcast ds_read_b128) this was 10 of the launch's 22 us.
    {
        f32x16_t T4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) T4[q][r] = 0.f;
        const float* ap = sA + l31 * P2_AS + hi * 12 + 4 * cc;                        // A operand: image l31, joint 2 s + hi, entry 4 cc + q
#pragma unroll
        for (int sidx = 0; sidx < NJ / 2; ++sidx) {
            float a4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) a4[q] = ap[sidx * 24 + q];
#pragma unroll
            for (int q = 0; q < 4; ++q) T4[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q], wq[sidx], T4[q], 0, 0, 0);
        }
        // C layout: register r of lane (l31, hi) = image (r & 3) + 8 (r >> 2) + 4 hi, vertex l31 of the tile
        if (vsk < NV) {
            const float* vp = sPO + 3 * (vt * 32 + l31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int bb = (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (b0 + bb < B) {
                    const float x = vp[bb * (3 * FUSED_VT)], y = vp[bb * (3 * FUSED_VT) + 1], z = vp[bb * (3 * FUSED_VT) + 2];
                    st_f<COH>(p.verts + ((size_t)(b0 + bb) * NV + vsk) * 3 + cc, fmaf(T4[2][r], z, fmaf(T4[1][r], y, T4[0][r] * x)) + T4[3][r]);
                }
            }
        }
    }
    __syncthreads();                                                              // LDS is reused by the next item
    if (stamps) stamps[3] = wall_clock64();
}

// Joint regression as a CSR gather over the skinned mesh + the stage tail, one workgroup (NT threads) per image b = first, first + step, ...:
// products through LDS (all column indices first, then all vertex reads), one thread per (row, coordinate) adds its segment in index order
// (deterministic).  COH: the vertices were written by an earlier phase of the SAME kernel (coherent loads).
template <bool COH, int NT>
__device__ __forceinline__ void fused_phase3(const whmr_smpl_model& m, const whmr_stage_tail& tail, const float* __restrict__ verts,
                                             const int32_t* __restrict__ reg_ptr, const int32_t* __restrict__ reg_col, const float* __restrict__ reg_val,
                                             int B, int first, int step, char* smem) {
    const int tid = threadIdx.x;
    float (*sReg)[3] = (float (*)[3])smem;                                    // [36][3]
    float (*sJ)[3] = (float (*)[3])(smem + 36 * 3 * 4);                       // [49][3]
    float* sProd = (float*)(smem + (36 + 49) * 3 * 4 + 12);                   // [FUSED_NNZ][3]
    int32_t* sTab = (int32_t*)(sProd + FUSED_NNZ * 3);                        // joint_map [49] | extra ids [21] | marker ids [<= 186]
    const int R = tail.R;
    const int nnz = reg_ptr[R];
    const int nmk = m.n_markers < 186 ? m.n_markers : 0;                      // (more markers than the table holds: read them from global memory)
    for (int e = tid; e < 70 + nmk; e += NT) sTab[e] = e < 49 ? m.joint_map[e] : e < 70 ? m.extra_vertex_ids[e - 49] : m.marker_ids[e - 70];
    for (int b = first; b < B; b += step) {
        const float* vb = verts + (size_t)b * NV * 3;
        if (tid < R * 3) sReg[tid / 3][tid % 3] = 0.f;
        for (int e0 = 0; e0 < nnz; e0 += FUSED_NNZ) {
            __syncthreads();
            {
                constexpr int NE = (FUSED_NNZ + NT - 1) / NT;
                int cc[NE];
                float wv[NE], x[NE][3];
#pragma unroll
                for (int i = 0; i < NE; ++i) {
                    const int e = e0 + tid + i * NT;
                    const bool ok = e < nnz && tid + i * NT < FUSED_NNZ;
                    cc[i] = ok ? reg_col[e] : 0;
                    wv[i] = ok ? reg_val[e] : 0.f;
                }
#pragma unroll
                for (int i = 0; i < NE; ++i)
#pragma unroll
                    for (int c = 0; c < 3; ++c) x[i][c] = ld_f<COH>(vb + 3 * cc[i] + c);
#pragma unroll
                for (int i = 0; i < NE; ++i) {
                    const int e = e0 + tid + i * NT;
                    if (e < nnz && tid + i * NT < FUSED_NNZ) { float* d = sProd + (e - e0) * 3; d[0] = wv[i] * x[i][0]; d[1] = wv[i] * x[i][1]; d[2] = wv[i] * x[i][2]; }
                }
            }
            __syncthreads();
            if (tid < R * 3) {                                                // one thread per (row, coordinate): its segment, in index order
                const int r = tid / 3, c = tid % 3;
                int lo = reg_ptr[r], hi = reg_ptr[r + 1];
                lo = lo < e0 ? e0 : lo;
                hi = hi > e0 + FUSED_NNZ ? e0 + FUSED_NNZ : hi;
                float a = sReg[r][c];
                for (int e = lo; e < hi; ++e) a += sProd[(e - e0) * 3 + c];
                sReg[r][c] = a;
            }
        }
        __syncthreads();
        smpl_stage_tail_image<COH>(m, tail, b, tid, *(float (*)[36][3])sReg, *(float (*)[49][3])sJ, sTab, sTab + 49,
                                   nmk ? (const int32_t*)(sTab + 70) : m.marker_ids);
        __syncthreads();
    }
}

// ---- the stage tail with the joint regression as a CSR gather, ONE launch per stage (was smpl_regress_kernel over the dense [33, 6890] rows for
// B x 33 workgroups + smpl_stage_tail_kernel): the regressors are > 99 % zeros, a per-image workgroup gathers the ~1.2 k products it needs.
#define P3_LDS ((36 + 49) * 3 * 4 + 12 + FUSED_NNZ * 3 * 4 + 256 * 4)
__global__ __launch_bounds__(256) void smpl_tail_csr_kernel(const whmr_smpl_model m, const whmr_stage_tail t, const int32_t* __restrict__ reg_ptr,
                                                            const int32_t* __restrict__ reg_col, const float* __restrict__ reg_val, int B) {
    __shared__ __attribute__((aligned(16))) char smem[P3_LDS];
    fused_phase3<false, 256>(m, t, t.verts, reg_ptr, reg_col, reg_val, B, blockIdx.x, gridDim.x, smem);
}

extern "C" int whmr_smpl_stage_tail_csr(const whmr_smpl_model* m, const whmr_stage_tail* tt, const int32_t* reg_ptr, const int32_t* reg_col,
                                        const float* reg_val, int B, void* stream) {
    const whmr_stage_tail& t = *tt;
    if (B <= 0 || !t.verts || !t.posed_joints || !reg_ptr || !reg_col || !reg_val) return (int)hipErrorInvalidValue;
    if (t.R != 9 && t.R != 33) return (int)hipErrorInvalidValue;
    if (t.smpl_joints45 && t.R != 33) return (int)hipErrorInvalidValue;
    if (t.xc_next && (!t.state || !t.rotmat || !t.bbox_info)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(smpl_tail_csr_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, *m, t, reg_ptr, reg_col, reg_val, B);
    WHMR_CHECK_LAUNCH();
    return 0;
}

// ---- pose-corrective blend shapes + skinning as ONE launch of the per-phase path (was the [B, 207] x [207, 20670] fp32 GEMM into a [B, 20670]
// buffer + smpl_skin_kernel<1>: 16.5 + 17.5 us at batch 64): one workgroup per (64-vertex chunk, 32-image group) runs phase 2 of the one-launch
// kernel above with plain cached accesses -- the offsets on v_mfma_f32_32x32x2_f32 from the re-tiled posedirs copy (same bits as the GEMM), through
// LDS to the skinning layout.  The pose-offset buffer (5.3 MB written + read at batch 64) is gone.
template <bool X3>
__global__ __launch_bounds__(FUSED_NT, 1) void smpl_blend_skin_kernel(const whmr_smpl_model m, const whmr_smpl_call p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // workgroup id -> (vertex chunk, image group) so that the image groups of ONE chunk are neighbours on the SAME XCD (ids with equal id % 8
    // share an XCD and its L2): they fetch the chunk's 160 KB posedirs tile at the same time and the L2 merges the misses -- 17 MB from HBM per
    // call instead of 17 MB per image group
    const int ngrp = (p.B + FUSED_IG - 1) / FUSED_IG, nvb = (NV + FUSED_VT - 1) / FUSED_VT;
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int vb = (k / ngrp) * 8 + xcd;
    if (vb >= nvb) return;
    fused_phase2_item<false, X3>(m, p, vb, (k % ngrp) * FUSED_IG, (float*)smem);
}

static uint32_t* g_blend_stamps = nullptr;          // tools only: whmr_set_option(200, 1) routes workgroup 0's phase stamps into a 64-byte device buffer
extern "C" int whmr_smpl_blend_skin_stamps(uint32_t* buf) { g_blend_stamps = buf; return 0; }

template <bool X3>
static int launch_blend_skin(const whmr_smpl_model* m, const whmr_smpl_call& f, void* stream) {
    const size_t lds = (size_t)P2_FLOATS * 4;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)smpl_blend_skin_kernel<X3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const int nvb = (NV + FUSED_VT - 1) / FUSED_VT, ngrp = (f.B + FUSED_IG - 1) / FUSED_IG;
    hipLaunchKernelGGL(smpl_blend_skin_kernel<X3>, dim3(8 * ((nvb + 7) / 8) * ngrp), dim3(FUSED_NT), lds, (hipStream_t)stream, *m, f);
    WHMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int whmr_smpl_blend_skin(const whmr_smpl_model* m, const float* posedirs_tiled, const float* betas, long beta_stride, const float* pose_feat,
                                    const float* A, int B, float* verts, void* stream) {
    if (B <= 0 || !posedirs_tiled || !betas || !pose_feat || !A || !verts) return (int)hipErrorInvalidValue;
    whmr_smpl_call f = {};
    f.betas = betas; f.beta_stride = beta_stride; f.B = B;
    f.A = const_cast<float*>(A); f.pose_feat = const_cast<float*>(pose_feat); f.verts = verts; f.posedirs_tiled = posedirs_tiled;
    f.barrier = g_blend_stamps;
    return launch_blend_skin<false>(m, f, stream);
}

// the same launch with the pose-corrective offsets on split-bf16 operands (fused_phase2_item<., X3 = true>); posedirs_x3: [108][13][2][2][192][8] bf16
extern "C" int whmr_smpl_blend_skin_x3(const whmr_smpl_model* m, const void* posedirs_x3, const float* betas, long beta_stride, const float* pose_feat,
                                       const float* A, int B, float* verts, void* stream) {
    if (B <= 0 || !posedirs_x3 || !betas || !pose_feat || !A || !verts) return (int)hipErrorInvalidValue;
    whmr_smpl_call f = {};
    f.betas = betas; f.beta_stride = beta_stride; f.B = B;
    f.A = const_cast<float*>(A); f.pose_feat = const_cast<float*>(pose_feat); f.verts = verts; f.posedirs_x3 = (const bf16_t*)posedirs_x3;
    f.barrier = g_blend_stamps;
    return launch_blend_skin<true>(m, f, stream);
}
