// Device-side pieces of the SMPL forward shared by the per-phase kernels (smpl_lbs.hip) and the one-launch call (smpl_fused.hip): the SAME
// inlined code runs in both, so the two paths produce the same bits.
#pragma once
#include "geometry_dev.h"

#define NV 6890
#define NJ 24
#define NPF 207

struct whmr_smpl_model {
    const float* v_template;     // [6890,3]
    const float* shapedirs;      // [30,6890]   = shapedirs[v][c][l] transposed to [(c*10+l)][v] by the host: coalesced over vertices
    const float* posedirs;       // [207, 20670]  (smplx layout)
    const float* lbs_weights;    // [24,6890]   = lbs_weights transposed by the host
    const float* J_template;     // [24,3]      = J_regressor . v_template
    const float* J_shapedirs;    // [24,3,10]   = J_regressor . shapedirs
    const float* J_regressor;    // [24,6890]   (whmr.py:186 smpl_joints; may be null if never requested)
    const float* J_regressor_extra;  // [9,6890]
    const int32_t* parents;      // [24]
    const int32_t* extra_vertex_ids; // [21]
    const int32_t* joint_map;    // [49] into the 54-joint superset
    const int32_t* marker_ids;   // [n_markers]
    int32_t n_markers;
};

// Cross-phase traffic of the one-launch call (smpl_fused.hip): what one phase writes and a later phase of the SAME kernel reads on another CU /
// XCD goes through agent-scope (sc1) stores and loads -- coherent at the device level without flushing the XCD-private L2s at the grid barrier
// (a release / acquire fence pair per workgroup writes back and invalidates the whole L2: measured 226 us per call instead of 60).  COH = false
// (the per-phase kernels: a kernel boundary orders everything) compiles to plain accesses.
template <bool COH> __device__ __forceinline__ void st_f(float* p, float v) {
    if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}
template <bool COH> __device__ __forceinline__ float ld_f(const float* p) {
    if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else return *p;
}

// LDS of one image's chain: rotations, rest joints, world transforms
struct smpl_chain_lds { float sR[NJ][9]; float sJ[NJ][3]; float sG[NJ][12]; };

// One image's pose chain on one wave (lane = 0..63).  Every wave of the workgroup must call it together (it synchronises with __syncthreads: the
// per-phase kernel runs one wave per workgroup, the fused kernel four images per workgroup in lockstep); `valid` = this wave has an image.
template <bool COH = false>
__device__ __forceinline__ void smpl_chain_image(const whmr_smpl_model& m, const float* __restrict__ pose9, long pose_stride,
                                                 const float* __restrict__ betas, long beta_stride, int do_gs,
                                                 float* __restrict__ rotmat, float* __restrict__ aa,
                                                 float* __restrict__ A, float* __restrict__ posed_joints,
                                                 float* __restrict__ pose_feat, int b, int lane, bool valid, smpl_chain_lds& L) {
    float (&sR)[NJ][9] = L.sR;
    float (&sJ)[NJ][3] = L.sJ;
    float (&sG)[NJ][12] = L.sG;
    if (valid && lane < NJ) {
        float r[9], o[9];
        for (int k = 0; k < 9; ++k) r[k] = pose9[(size_t)b * pose_stride + lane * 9 + k];
        if (do_gs) gram_schmidt9(r, o); else for (int k = 0; k < 9; ++k) o[k] = r[k];
        for (int k = 0; k < 9; ++k) { sR[lane][k] = o[k]; if (rotmat) st_f<COH>(rotmat + ((size_t)b * NJ + lane) * 9 + k, o[k]); }
        if (aa) { float a3[3]; rotmat_to_aa3(o, a3); for (int k = 0; k < 3; ++k) st_f<COH>(aa + (size_t)b * 72 + lane * 3 + k, a3[k]); }
        if (lane >= 1 && pose_feat) {
            for (int k = 0; k < 9; ++k)
                st_f<COH>(pose_feat + (size_t)b * NPF + (lane - 1) * 9 + k, o[k] - ((k == 0 || k == 4 || k == 8) ? 1.f : 0.f));
        }
        // joint locations of the shaped rest mesh
        for (int c = 0; c < 3; ++c) {
            float acc = 0.f;
            for (int l = 0; l < 10; ++l) acc = fmaf(m.J_shapedirs[(lane * 3 + c) * 10 + l], betas[(size_t)b * beta_stride + l], acc);
            sJ[lane][c] = m.J_template[lane * 3 + c] + acc;
        }
    }
    __syncthreads();
    // kinematic chain: G_0 = [R_0 | J_0], G_i = G_parent . [R_i | J_i - J_parent]   (lbs.py:41-49); lanes 0..11 own one entry
    const int row = lane / 4, col = lane % 4;
    if (valid && lane < 12) sG[0][lane] = (col < 3) ? sR[0][row * 3 + col] : sJ[0][row];
    __syncthreads();
    // By TREE LEVEL (round 5): the joints of one depth only depend on the level above, so a level is ONE step -- lanes = (slot q = lane / 12 of
    // up to five joints, transform entry lane % 12) -- and the SMPL tree (depths 0..8, at most five joints per level) takes 8 dependent LDS round
    // trips instead of 23.  The depths come from the parents table in registers (lane reads, no memory), a level's joints from a ballot mask; any
    // tree works (a level with more than five joints is walked in several steps).  Per joint the arithmetic is the expression of the serial loop:
    // the same bits.
    const int par = m.parents[lane < NJ ? lane : 0];
    int depth = 0;
#pragma unroll
    for (int i = 1; i < NJ; ++i) {
        const int pi = __builtin_amdgcn_readlane(par, i);
        const int dp = __builtin_amdgcn_readlane(depth, pi < 0 ? 0 : pi);       // parents precede their children (smplx kinematic tree order)
        if (lane == i) depth = dp + 1;
    }
    int maxd = 0;
#pragma unroll
    for (int i = 1; i < NJ; ++i) { const int di = __builtin_amdgcn_readlane(depth, i); maxd = di > maxd ? di : maxd; }
    const int q = lane / 12, e12 = lane - q * 12;
    for (int lev = 1; lev <= maxd; ++lev) {
        // (the mask is built from the tree alone -- NOT from `valid` -- so that a wave without an image runs the SAME number of barriers as one with:
        //  `valid` only gates the LDS writes below)
        unsigned long long mask = __ballot(lane >= 1 && lane < NJ && depth == lev);
        while (mask) {                                                             // (wave-uniform: at most ceil(joints of the level / 5) rounds)
            int i = -1, p = 0;
#pragma unroll
            for (int t = 0; t < 5; ++t) {                                          // the level's next five joints and their parents: scalar work (mask is wave-uniform)
                const int it = mask ? (int)__builtin_ctzll(mask) : -1;
                const int pt = __builtin_amdgcn_readlane(par, it < 0 ? 0 : it);
                if (q == t) { i = it; p = pt; }
                mask &= mask - 1;
            }
            if (valid && i >= 0) {
                const int row2 = e12 / 4, col2 = e12 % 4;
                const float g0 = sG[p][row2 * 4 + 0], g1 = sG[p][row2 * 4 + 1], g2 = sG[p][row2 * 4 + 2], g3 = sG[p][row2 * 4 + 3];
                float v;
                if (col2 < 3) v = g0 * sR[i][col2] + g1 * sR[i][3 + col2] + g2 * sR[i][6 + col2];
                else v = g0 * (sJ[i][0] - sJ[p][0]) + g1 * (sJ[i][1] - sJ[p][1]) + g2 * (sJ[i][2] - sJ[p][2]) + g3;
                sG[i][e12] = v;
            }
            __syncthreads();
        }
    }
    // A_i = G_i with translation  t_i - G_i[:, :3] . J_i   (lbs.py:51-55)
    for (int e = lane; valid && e < NJ * 12; e += 64) {
        const int j = e / 12, k = e % 12, r = k / 4, c = k % 4;
        float v = sG[j][k];
        if (c == 3) {
            v = v - (sG[j][r * 4] * sJ[j][0] + sG[j][r * 4 + 1] * sJ[j][1] + sG[j][r * 4 + 2] * sJ[j][2]);
            if (posed_joints) st_f<COH>(posed_joints + ((size_t)b * NJ + j) * 3 + r, sG[j][k]);
        }
        st_f<COH>(A + (size_t)b * NJ * 12 + e, v);
    }
}

// skinning of one vertex for one image: T = sum_j w_j A_j ; v = T [v_posed; 1]   (lbs.py:67-77); A = this image's 24 x (3x4) transforms (LDS, 16-B aligned)
template <bool COH = false>
__device__ __forceinline__ void smpl_skin_vertex(const float (&w)[NJ], const float* __restrict__ A, float x, float y, float z, float* __restrict__ o) {
    float T[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) T[e] = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const float4* a4 = (const float4*)(A + j * 12);
        const float4 r0 = a4[0], r1 = a4[1], r2 = a4[2];
        T[0] = fmaf(w[j], r0.x, T[0]); T[1] = fmaf(w[j], r0.y, T[1]); T[2] = fmaf(w[j], r0.z, T[2]); T[3] = fmaf(w[j], r0.w, T[3]);
        T[4] = fmaf(w[j], r1.x, T[4]); T[5] = fmaf(w[j], r1.y, T[5]); T[6] = fmaf(w[j], r1.z, T[6]); T[7] = fmaf(w[j], r1.w, T[7]);
        T[8] = fmaf(w[j], r2.x, T[8]); T[9] = fmaf(w[j], r2.y, T[9]); T[10] = fmaf(w[j], r2.z, T[10]); T[11] = fmaf(w[j], r2.w, T[11]);
    }
    st_f<COH>(o, fmaf(T[2], z, fmaf(T[1], y, T[0] * x)) + T[3]);
    st_f<COH>(o + 1, fmaf(T[6], z, fmaf(T[5], y, T[4] * x)) + T[7]);
    st_f<COH>(o + 2, fmaf(T[10], z, fmaf(T[9], y, T[8] * x)) + T[11]);
}

// the same with the image's transforms in an unaligned array (row-major [24][12])
template <bool COH = false>
__device__ __forceinline__ void smpl_skin_vertex_regs(const float (&w)[NJ], const float* __restrict__ A, float x, float y, float z, float* __restrict__ o) {
    float T[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) T[e] = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 12; ++e) T[e] = fmaf(w[j], A[j * 12 + e], T[e]);
    st_f<COH>(o, fmaf(T[2], z, fmaf(T[1], y, T[0] * x)) + T[3]);
    st_f<COH>(o + 1, fmaf(T[6], z, fmaf(T[5], y, T[4] * x)) + T[7]);
    st_f<COH>(o + 2, fmaf(T[10], z, fmaf(T[9], y, T[8] * x)) + T[11]);
}

// v_shaped = T + S . beta for one vertex (verts.py:46-48); s = the vertex's 30 shape directions [(c*10 + l)], beta: stride `bs` floats apart
__device__ __forceinline__ void smpl_shape_vertex(float t0, float t1, float t2, const float (&s)[30], const float* __restrict__ beta, int bs, float (&acc)[3]) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int l = 0; l < 10; ++l) {
        const float be = beta[l * bs];
        a0 = fmaf(s[l], be, a0); a1 = fmaf(s[10 + l], be, a1); a2 = fmaf(s[20 + l], be, a2);
    }
    acc[0] = t0 + a0; acc[1] = t1 + a1; acc[2] = t2 + a2;
}

// ---- fused tail of one regressor stage: joint regression + gather + projections + the next stage's input state --------------------------------
struct whmr_stage_tail {
    // always
    const float* verts; const float* posed_joints; const float* regd; float* joints49; float* smpl_joints45; float* markers; int32_t R;
    // projections (state != null): state rows [pose(216) | shape(10) | cam(3)]
    const float* state; int64_t state_stride; const float* aa; const float* Tz; const float* bbox_h; const float* center; const float* orig_shape;
    float focal0, res_w, res_h; float* theta; float* kp2d; float* kp2d_w; float* cam_t; float* focal;
    // next stage input (xc_next != null): xc_next[b*ld + F .. F+234) = [bbox_info(5) | rotmat(216) | shape(10) | cam(3)]
    const float* bbox_info; const float* rotmat; float* xc_next; int64_t ld_next; int32_t F_next;
};

// Tail of one image on a 256-thread workgroup; sReg [36][3] holds the image's regressed rows (extra rows first, then J_regressor) on entry.
template <bool COH = false>
__device__ __forceinline__ void smpl_stage_tail_image(const whmr_smpl_model& m, const whmr_stage_tail& t, int b, int tid, float (&sReg)[36][3],
                                                      float (&sJ)[49][3], const int32_t* __restrict__ joint_map, const int32_t* __restrict__ extra_ids,
                                                      const int32_t* __restrict__ marker_ids) {
    // joint_map / extra_ids / marker_ids: the model's index tables (global memory in the per-phase kernel, LDS copies in the one-launch kernel: a
    // table lookup in front of a gather is one more dependent memory round trip on a latency-bound tail)
    const float* vb = t.verts + (size_t)b * NV * 3;
    // ---- 54-joint superset -> JOINT_MAP (models/smpl.py:61-83), smpl_joints45 (whmr.py:186-187), markers (whmr.py:184)
    if (tid < 49 * 3) {
        const int j = tid / 3, c = tid % 3;
        const int s = joint_map[j];
        float v;
        if (s < 24) v = ld_f<COH>(t.posed_joints + ((size_t)b * NJ + s) * 3 + c);
        else if (s < 45) v = ld_f<COH>(vb + 3 * extra_ids[s - 24] + c);
        else v = sReg[s - 45][c];
        sJ[j][c] = v;
        if (t.joints49) t.joints49[((size_t)b * 49 + j) * 3 + c] = v;
    }
    if (t.smpl_joints45 && tid < 45 * 3) {
        const int j = tid / 3, c = tid % 3;
        t.smpl_joints45[((size_t)b * 45 + j) * 3 + c] = j < 24 ? sReg[9 + j][c] : ld_f<COH>(vb + 3 * extra_ids[j - 24] + c);
    }
    if (t.markers)
        for (int e = tid; e < m.n_markers * 3; e += 256) t.markers[((size_t)b * m.n_markers) * 3 + e] = ld_f<COH>(vb + 3 * marker_ids[e / 3] + e % 3);
    __syncthreads();
    if (t.state) {
        const float* st = t.state + (size_t)b * t.state_stride;
        const float s = st[226], tx = st[227], ty = st[228];
        const float h = t.bbox_h[b], tz = t.Tz[b];
        const float focal = s * h * tz / 2.f;
        const float H = t.orig_shape[2 * b], W = t.orig_shape[2 * b + 1];
        const float ctx = tx + 2.f * (t.center[2 * b] - W / 2.f) / (s * h);
        const float cty = ty + 2.f * (t.center[2 * b + 1] - H / 2.f) / (s * h);
        if (tid == 0) { t.cam_t[3 * b] = ctx; t.cam_t[3 * b + 1] = cty; t.cam_t[3 * b + 2] = tz; t.focal[b] = focal; }
        if (tid >= 64 && tid < 64 + 85) {
            const int e = tid - 64;
            t.theta[(size_t)b * 85 + e] = e < 3 ? st[226 + e] : (e < 13 ? st[216 + e - 3] : ld_f<COH>(t.aa + (size_t)b * 72 + e - 13));
        }
        if (tid < 49) {
            const float tzw = 2.f * t.focal0 / (t.res_h * s + 1e-9f);
            const float cxw = W / 2.f, cyw = H / 2.f;
            const float x = sJ[tid][0], y = sJ[tid][1], z = sJ[tid][2];
            const float zw = z + tzw;
            t.kp2d[((size_t)b * 49 + tid) * 2] = (t.focal0 * ((x + tx) / zw)) / (t.res_w / 2.f);
            t.kp2d[((size_t)b * 49 + tid) * 2 + 1] = (t.focal0 * ((y + ty) / zw)) / (t.res_h / 2.f);
            const float zf = z + tz;
            t.kp2d_w[((size_t)b * 49 + tid) * 2] = (focal * ((x + ctx) / zf) + cxw) / cxw - 1.f;
            t.kp2d_w[((size_t)b * 49 + tid) * 2 + 1] = (focal * ((y + cty) / zf) + cyw) / cyw - 1.f;
        }
        if (t.xc_next) {
            float* row = t.xc_next + (size_t)b * t.ld_next + t.F_next;
            if (tid < 216) row[5 + tid] = ld_f<COH>(t.rotmat + (size_t)b * 216 + tid);
            else if (tid < 229) row[5 + tid] = st[tid];                       // shape(10) | cam(3) sit at state[216..229)
            else if (tid < 234) row[tid - 229] = t.bbox_info[b * 5 + tid - 229];
        }
    }
}
