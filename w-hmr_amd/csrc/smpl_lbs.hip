// SMPL forward (linear blend skinning) for W-HMR on gfx950, fp32.
//
// Replaces pare.models.SMPL.forward(pose2rot=False) (smplx==0.1.28 lbs) as called at models/whmr.py:132-137,227-232,
// 641-644, plus the surrounding small ops of Regressor.forward: unbiased_gram_schmidt (whmr.py:129-130),
// rotation_matrix_to_angle_axis (whmr.py:174), vertex_joint_selector / J_regressor (whmr.py:186-187), markers
// (whmr.py:184).  In-tree spec of the math: models/smpl_webuser/lbs.py:27-80, verts.py:39-67, models/smpl.py:61-83.
//
// Three launches per SMPL call:
//   1. smpl_pose_chain : per image, one wave -- (optional Gram-Schmidt) -> rotmats, angle-axis, joint locations
//      J = J_template + J_shapedirs.beta (= Jreg.(T + S.beta), regressor folded into the constants once), the 24-joint
//      kinematic chain and the rest-pose-removed skinning transforms A[24][3x4], pose feature (R[1:] - I).
//   2. smpl_skin       : vertex-parallel, HBM/L2-bound -- v_shaped, + posedirs^T.pose_feature (the 17 MB operand is
//      read coalesced, once per block for BT images), T = sum_j w_vj A_j, v = T [v_posed; 1].
//   3. smpl_joints     : per image -- extra-joint regressors over the skinned mesh, vertex picks, JOINT_MAP gather.
// Floating-point contraction per EXPRESSION (the language rule), not across statements after inlining (hipcc's default "fast"): whether a product
// is fused into an fma then depends on the source expression only, not on the kernel it was inlined into -- the per-phase kernels and the
// one-launch kernel share smpl_dev.h / geometry_dev.h and must produce the same bits.
#pragma clang fp contract(on)
#include "smpl_dev.h"

__global__ __launch_bounds__(64) void smpl_pose_chain_kernel(const whmr_smpl_model m, const float* __restrict__ pose9, long pose_stride,
                                                             const float* __restrict__ betas, long beta_stride, int do_gs,
                                                             float* __restrict__ rotmat, float* __restrict__ aa,
                                                             float* __restrict__ A, float* __restrict__ posed_joints,
                                                             float* __restrict__ pose_feat) {
    __shared__ smpl_chain_lds L;
    smpl_chain_image(m, pose9, pose_stride, betas, beta_stride, do_gs, rotmat, aa, A, posed_joints, pose_feat, blockIdx.x, threadIdx.x, true, L);
}

template <int BT>
__global__ __launch_bounds__(128) void smpl_skin_kernel(const whmr_smpl_model m, const float* __restrict__ betas, long beta_stride,
                                                        const float* __restrict__ pose_feat, const float* __restrict__ A,
                                                        const float* __restrict__ pose_off, int B, float* __restrict__ verts) {
    __shared__ float sPF[NPF][BT];          // [k][b]: one ds_read_b128 pair fetches all BT coefficients of step k
    __shared__ float sBeta[10][BT];
    __shared__ __attribute__((aligned(16))) float sA[BT][NJ * 12];
    const int tid = threadIdx.x;
    const int b0 = blockIdx.y * BT;
    const int v = blockIdx.x * 128 + tid;
    for (int e = tid; e < NPF * BT; e += 128) {
        const int k = e / BT, bb = e % BT;
        sPF[k][bb] = (b0 + bb < B) ? pose_feat[(size_t)(b0 + bb) * NPF + k] : 0.f;
    }
    for (int e = tid; e < 10 * BT; e += 128) {
        const int k = e / BT, bb = e % BT;
        sBeta[k][bb] = (b0 + bb < B) ? betas[(size_t)(b0 + bb) * beta_stride + k] : 0.f;
    }
    for (int e = tid; e < BT * NJ * 12; e += 128) {
        const int bb = e / (NJ * 12);
        sA[bb][e % (NJ * 12)] = (b0 + bb < B) ? A[(size_t)(b0 + bb) * NJ * 12 + (e % (NJ * 12))] : 0.f;
    }
    __syncthreads();
    if (v >= NV) return;

    float acc[BT][3];
    {   // v_shaped = T + S . beta   (verts.py:46-48)
        const float t0 = m.v_template[3 * v], t1 = m.v_template[3 * v + 1], t2 = m.v_template[3 * v + 2];
        float s[30];
#pragma unroll
        for (int k = 0; k < 30; ++k) s[k] = m.shapedirs[(size_t)k * NV + v];
#pragma unroll
        for (int bb = 0; bb < BT; ++bb) smpl_shape_vertex(t0, t1, t2, s, &sBeta[0][bb], BT, acc[bb]);
    }
    if (pose_off) {   // pose-corrective offsets precomputed as one [B,207] x [207,20670] GEMM (whmr_gemm_f32): coalesced 12-B reads
#pragma unroll
        for (int bb = 0; bb < BT; ++bb) {
            if (b0 + bb < B) {
                const float* po = pose_off + ((size_t)(b0 + bb) * NV + v) * 3;
                acc[bb][0] += po[0]; acc[bb][1] += po[1]; acc[bb][2] += po[2];
            }
        }
    } else {   // v_posed = v_shaped + posedirs^T . pose_feature   (verts.py:51-53); separate accumulator like the reference's matmul-then-add
        float po[BT][3];
#pragma unroll
        for (int bb = 0; bb < BT; ++bb) po[bb][0] = po[bb][1] = po[bb][2] = 0.f;
        const float* pd = m.posedirs + 3 * v;
#pragma unroll 3
        for (int k = 0; k < NPF; ++k) {
            const float p0 = pd[(size_t)k * (NV * 3)], p1 = pd[(size_t)k * (NV * 3) + 1], p2 = pd[(size_t)k * (NV * 3) + 2];
#pragma unroll
            for (int bb = 0; bb < BT; ++bb) {
                const float f = sPF[k][bb];
                po[bb][0] = fmaf(f, p0, po[bb][0]); po[bb][1] = fmaf(f, p1, po[bb][1]); po[bb][2] = fmaf(f, p2, po[bb][2]);
            }
        }
#pragma unroll
        for (int bb = 0; bb < BT; ++bb) { acc[bb][0] += po[bb][0]; acc[bb][1] += po[bb][1]; acc[bb][2] += po[bb][2]; }
    }
    // skinning: T = sum_j w_j A_j ; v = T [v_posed; 1]   (lbs.py:67-77)
    float w[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) w[j] = m.lbs_weights[(size_t)j * NV + v];
#pragma unroll
    for (int bb = 0; bb < BT; ++bb) {
        if (b0 + bb >= B) break;
        smpl_skin_vertex(w, sA[bb], acc[bb][0], acc[bb][1], acc[bb][2], verts + ((size_t)(b0 + bb) * NV + v) * 3);
    }
}

// Regress `R` rows of a dense [R,6890] matrix over every image's skinned mesh: one workgroup per (image, row) -- 33 rows x B
// images give thousands of independent workgroups instead of one block per image walking the rows serially.
// out[(b*R + r)*3 + c] = sum_v reg[r][v] * verts[b][v][c].  Exact zeros are skipped (the real regressors are >99 % zeros).
__global__ __launch_bounds__(256) void smpl_regress_kernel(const float* __restrict__ reg, int R, const float* __restrict__ verts,
                                                           int B, float* __restrict__ out) {
    // one WORKGROUP per (image, row): the 6890-vertex dot product is split over 4 waves, so a lane walks 3 rounds of
    // [9 weight loads -> (non-zero?) -> 27 vertex loads] instead of 12 -- the two dependent memory round trips per round
    // were the whole run time.
    __shared__ float red[4][3];
    const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = w / R, r = w - b * R;
    const float* rr = reg + (size_t)r * NV;
    const float* vb = verts + (size_t)b * NV * 3;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int v0 = tid; v0 < NV; v0 += 256 * 9) {          // 9 independent coalesced weight loads in flight per lane
        float wv[9];
#pragma unroll
        for (int u = 0; u < 9; ++u) { const int v = v0 + 256 * u; wv[u] = v < NV ? rr[v] : 0.f; }
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            const int v = v0 + 256 * u;
            if (wv[u] != 0.f) { a0 = fmaf(wv[u], vb[3 * v], a0); a1 = fmaf(wv[u], vb[3 * v + 1], a1); a2 = fmaf(wv[u], vb[3 * v + 2], a2); }
        }
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
    if (lane == 0) { red[wave][0] = a0; red[wave][1] = a1; red[wave][2] = a2; }
    __syncthreads();
    if (tid < 3) out[(size_t)w * 3 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

// Gather stage: 54-joint superset (24 posed SMPL joints, 21 picked vertices, 9 regressed; models/smpl.py:61-83) -> JOINT_MAP
// -> joints49; whmr.py:186-187 smpl_joints45 (J_regressor over the POSED mesh + vertex_joint_selector); markers (whmr.py:184).
// `regd` = [B, 9 (+24)] x 3 rows from smpl_regress_kernel (extra first, then J_regressor when requested).
__global__ __launch_bounds__(256) void smpl_joints_kernel(const whmr_smpl_model m, const float* __restrict__ verts,
                                                          const float* __restrict__ posed_joints, const float* __restrict__ regd,
                                                          int R, float* __restrict__ joints49, float* __restrict__ smpl_joints45,
                                                          float* __restrict__ markers) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* vb = verts + (size_t)b * NV * 3;
    const float* rb = regd + (size_t)b * R * 3;
    if (joints49 && tid < 49 * 3) {
        const int j = tid / 3, c = tid % 3;
        const int s = m.joint_map[j];
        float v;
        if (s < 24) v = posed_joints[((size_t)b * NJ + s) * 3 + c];
        else if (s < 45) v = vb[3 * m.extra_vertex_ids[s - 24] + c];
        else v = rb[(s - 45) * 3 + c];
        joints49[((size_t)b * 49 + j) * 3 + c] = v;
    }
    if (smpl_joints45 && tid < 45 * 3) {
        const int j = tid / 3, c = tid % 3;
        smpl_joints45[((size_t)b * 45 + j) * 3 + c] = j < 24 ? rb[(9 + j) * 3 + c] : vb[3 * m.extra_vertex_ids[j - 24] + c];
    }
    if (markers) {
        for (int e = tid; e < m.n_markers * 3; e += 256)
            markers[((size_t)b * m.n_markers) * 3 + e] = vb[3 * m.marker_ids[e / 3] + e % 3];
    }
}

// pose_stride / beta_stride: row strides (elements) of pose9 [B,216] and betas [B,10] -- they may be column slices of the
// regressor state buffer [.., pose(216) | shape(10) | cam(3)].
extern "C" int whmr_smpl_pose_chain(const whmr_smpl_model* m, const float* pose9, long pose_stride, const float* betas,
                                    long beta_stride, int B, int do_gs, float* rotmat, float* aa, float* A, float* posed_joints,
                                    float* pose_feat, void* stream) {
    if (B <= 0 || !A) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(smpl_pose_chain_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, *m, pose9, pose_stride, betas, beta_stride,
                       do_gs, rotmat, aa, A, posed_joints, pose_feat);
    WHMR_CHECK_LAUNCH();
    return 0;
}

// pose_off (nullable): [B, 20670] = pose_feat . posedirs computed by the caller with whmr_gemm_f32 (the 17 MB operand is then
// streamed once by an MFMA GEMM instead of once per 8-image block of this kernel); null = compute it here.
extern "C" int whmr_smpl_skin(const whmr_smpl_model* m, const float* betas, long beta_stride, const float* pose_feat, const float* A,
                              const float* pose_off, int B, float* verts, void* stream) {
    if (B <= 0) return (int)hipErrorInvalidValue;
    // images per block: with the pose-corrective offsets precomputed (pose_off) the per-vertex constants are small, so more, smaller
    // blocks (1 image each: ~7 waves per SIMD at batch 64) hide the LDS / load latency that one 8-image block per 128 vertices exposes
    if (pose_off) {
        constexpr int BT = 1;
        hipLaunchKernelGGL(smpl_skin_kernel<BT>, dim3((NV + 127) / 128, (B + BT - 1) / BT), dim3(128), 0, (hipStream_t)stream, *m,
                           betas, beta_stride, pose_feat, A, pose_off, B, verts);
        WHMR_CHECK_LAUNCH();
        return 0;
    }
    constexpr int BT = 8;
    hipLaunchKernelGGL(smpl_skin_kernel<BT>, dim3((NV + 127) / 128, (B + BT - 1) / BT), dim3(128), 0, (hipStream_t)stream, *m,
                       betas, beta_stride, pose_feat, A, pose_off, B, verts);
    WHMR_CHECK_LAUNCH();
    return 0;
}

// scratch: >= B*33*3 floats of device workspace (regressed rows).  J_regressor_extra and J_regressor must be ONE contiguous
// [33,6890] buffer (extra rows first) when smpl_joints45 is requested; the host wrapper builds it once.
extern "C" int whmr_smpl_joints(const whmr_smpl_model* m, const float* verts, const float* posed_joints, int B,
                                float* joints49, float* smpl_joints45, float* markers, float* scratch, void* stream) {
    if (B <= 0 || !scratch) return (int)hipErrorInvalidValue;
    if (smpl_joints45 && !m->J_regressor) return (int)hipErrorInvalidValue;
    const int R = smpl_joints45 ? 33 : 9;
    if (smpl_joints45 && m->J_regressor != m->J_regressor_extra + 9 * NV) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(smpl_regress_kernel, dim3(B * R), dim3(256), 0, st, m->J_regressor_extra, R, verts, B, scratch);
    hipLaunchKernelGGL(smpl_joints_kernel, dim3(B), dim3(256), 0, st, *m, verts, posed_joints, scratch, R, joints49,
                       smpl_joints45, markers);
    WHMR_CHECK_LAUNCH();
    return 0;
}

// ---- fused tail of one regressor stage: joint regression + gather + projections + the next stage's input state, ONE launch --------------
// smpl_joints + regressor_post + regressor_state (+ a strided-copy of the camera) were dependent launches of ~5 us each on a chain that is
// pure latency (PyMAF loop, whmr.py:580-627).  One workgroup per image, right behind smpl_regress_kernel: the 54-joint superset is gathered
// through LDS, the first wave projects the 49 joints (whmr.py:142-173) and all threads write [bbox_info | rotmat | shape | cam] into the
// NEXT stage's input buffer (whmr.py:105,119).
__global__ __launch_bounds__(256) void smpl_stage_tail_kernel(const whmr_smpl_model m, const whmr_stage_tail t) {
    __shared__ float sReg[36][3];
    __shared__ float sJ[49][3];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int R = t.R;
    // ---- regressed rows of this image (smpl_regress_kernel: B x R workgroups -- a per-image block walking the 33 dense rows itself was 5x slower)
    if (tid < R * 3) sReg[tid / 3][tid % 3] = t.regd[(size_t)b * R * 3 + tid];
    __syncthreads();
    smpl_stage_tail_image(m, t, b, tid, sReg, sJ, m.joint_map, m.extra_vertex_ids, m.marker_ids);
}

// scratch: >= B*33*3 floats (the regressed rows); two launches: smpl_regress_kernel (B x R workgroups) + the tail.
extern "C" int whmr_smpl_stage_tail(const whmr_smpl_model* m, const whmr_stage_tail* tt, int B, float* scratch, void* stream) {
    whmr_stage_tail tv = *tt;
    const whmr_stage_tail* t = &tv;
    if (B <= 0 || !t->verts || !t->posed_joints || !scratch) return (int)hipErrorInvalidValue;
    if (t->R != 9 && t->R != 33) return (int)hipErrorInvalidValue;
    if (t->R == 33 && m->J_regressor != m->J_regressor_extra + 9 * NV) return (int)hipErrorInvalidValue;
    if (t->smpl_joints45 && t->R != 33) return (int)hipErrorInvalidValue;
    if (t->xc_next && (!t->state || !t->rotmat || !t->bbox_info)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(smpl_regress_kernel, dim3(B * t->R), dim3(256), 0, (hipStream_t)stream, m->J_regressor_extra, t->R, t->verts, B, scratch);
    tv.regd = scratch;
    hipLaunchKernelGGL(smpl_stage_tail_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, *m, tv);
    WHMR_CHECK_LAUNCH();
    return 0;
}

// Tail of Regressor.forward (whmr.py:142-174) in one launch per batch: weak-perspective key points (geometry.py:289-307),
// focal length s*h*Tz/2 (whmr.py:147-149), full-image camera translation (geometry.py:139-157), perspective key points in the
// full image normalised by the image centre (whmr.py:165-173) and theta = [cam | shape | angle-axis] (whmr.py:190).
// state rows hold [... pose(216) | shape(10) | cam(3)]; `state` points at the pose column, row stride state_stride.
__global__ __launch_bounds__(64) void regressor_post_kernel(const float* __restrict__ state, long state_stride,
                                                            const float* __restrict__ aa, const float* __restrict__ joints49,
                                                            const float* __restrict__ Tz, const float* __restrict__ bbox_h,
                                                            const float* __restrict__ center, const float* __restrict__ orig_shape,
                                                            float focal0, float res_w, float res_h, float* __restrict__ theta,
                                                            float* __restrict__ kp2d, float* __restrict__ kp2d_w,
                                                            float* __restrict__ cam_t_out, float* __restrict__ focal_out) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float* st = state + (size_t)b * state_stride;
    const float s = st[226], tx = st[227], ty = st[228];
    const float h = bbox_h[b], tz = Tz[b];
    const float focal = s * h * tz / 2.f;
    const float H = orig_shape[2 * b], W = orig_shape[2 * b + 1];
    const float ctx = tx + 2.f * (center[2 * b] - W / 2.f) / (s * h);
    const float cty = ty + 2.f * (center[2 * b + 1] - H / 2.f) / (s * h);
    if (lane == 0) {
        cam_t_out[3 * b] = ctx; cam_t_out[3 * b + 1] = cty; cam_t_out[3 * b + 2] = tz;
        focal_out[b] = focal;
    }
    for (int e = lane; e < 85; e += 64) theta[(size_t)b * 85 + e] = e < 3 ? st[226 + e] : (e < 13 ? st[216 + e - 3] : aa[(size_t)b * 72 + e - 13]);
    const float tzw = 2.f * focal0 / (res_h * s + 1e-9f);
    const float cxw = W / 2.f, cyw = H / 2.f;
    for (int j = lane; j < 49; j += 64) {
        const float* q = joints49 + ((size_t)b * 49 + j) * 3;
        const float x = q[0], y = q[1], z = q[2];
        const float zw = z + tzw;
        kp2d[((size_t)b * 49 + j) * 2] = (focal0 * ((x + tx) / zw)) / (res_w / 2.f);
        kp2d[((size_t)b * 49 + j) * 2 + 1] = (focal0 * ((y + ty) / zw)) / (res_h / 2.f);
        const float zf = z + tz;
        kp2d_w[((size_t)b * 49 + j) * 2] = (focal * ((x + ctx) / zf) + cxw) / cxw - 1.f;
        kp2d_w[((size_t)b * 49 + j) * 2 + 1] = (focal * ((y + cty) / zf) + cyw) / cyw - 1.f;
    }
}

extern "C" int whmr_regressor_post(const float* state, long state_stride, const float* aa, const float* joints49, const float* Tz,
                                   const float* bbox_h, const float* center, const float* orig_shape, int B, float focal0, float res_w,
                                   float res_h, float* theta, float* kp2d, float* kp2d_w, float* cam_t, float* focal, void* stream) {
    if (B <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(regressor_post_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, state, state_stride, aa, joints49, Tz, bbox_h,
                       center, orig_shape, focal0, res_w, res_h, theta, kp2d, kp2d_w, cam_t, focal);
    WHMR_CHECK_LAUNCH();
    return 0;
}


// Regressor input assembly (whmr.py:105,119): xc[b, F..F+234) = [bbox_info(5) | pose(216) | shape(10) | cam(3)] in one launch.
// pose / shape / cam rows may be strided views (row stride 0 = one broadcast row: the mean-parameter initial state).
__global__ __launch_bounds__(256) void regressor_state_kernel(const float* __restrict__ bbox, const float* __restrict__ pose, long ld_p,
                                                              const float* __restrict__ shape, long ld_s, const float* __restrict__ cam,
                                                              long ld_c, float* __restrict__ xc, long ld, int F) {
    const int b = blockIdx.x, t = threadIdx.x;
    float* row = xc + b * ld + F;
    if (t < 216) row[5 + t] = pose[b * ld_p + t];
    else if (t < 226) row[5 + t] = shape[b * ld_s + t - 216];
    else if (t < 229) row[5 + t] = cam[b * ld_c + t - 226];
    else if (t < 234) row[t - 229] = bbox[b * 5 + t - 229];
}

extern "C" int whmr_regressor_state(const float* bbox_info, const float* pose, long ld_p, const float* shape, long ld_s, const float* cam,
                                    long ld_c, int B, float* xc, long ld, int F, void* stream) {
    if (B <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(regressor_state_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, bbox_info, pose, ld_p, shape, ld_s, cam, ld_c, xc, ld, F);
    WHMR_CHECK_LAUNCH();
    return 0;
}


// =====================================================================================================================
// Backward of the SMPL forward above (training; reference: autograd through pare.models.SMPL / smplx lbs as called at
// models/whmr.py:132-137 with pose2rot=False, driven by core/trainer.py:410-470).  Inputs of the differentiated function:
// betas [B,10], rotmats [B,24,3,3]; outputs: verts, joints49 (+ smpl_joints45, markers).  Four launches + one GEMM:
//   smpl_joints_bwd  d(joints49 / smpl_joints45 / markers) -> d(posed joints), d(regressed rows), vertex picks added into d_verts
//   smpl_skin_bwd    vertex-parallel: d_vposed = T_rot^T dv, per-block partial sums of dA_j = sum_v w_vj [dv (x) v_posed | dv]
//   (GEMM)           [B,20670] x [posedirs ; shapedirs]^T -> d(pose feature) [B,207] | d(betas via v_shaped) [B,10]
//   smpl_chain_bwd   per image: partial sums -> dA, reverse kinematic chain -> d(rotmats), d(betas)
// All reductions are fixed-order (no atomics): the vertex picks are distinct indices, so the in-place adds cannot collide.
#define SKIN_BWD_BLOCKS ((NV + 127) / 128)

__global__ __launch_bounds__(256) void smpl_joints_bwd_kernel(const whmr_smpl_model m, const float* __restrict__ d_joints49,
                                                              const float* __restrict__ d_smpl_joints45, const float* __restrict__ d_markers,
                                                              float* __restrict__ d_verts, float* __restrict__ d_posed_joints,
                                                              float* __restrict__ d_regd, int R) {
    __shared__ float d54[54][3];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid < 54 * 3) {
        const int s = tid / 3, c = tid % 3;
        float a = 0.f;
        if (d_joints49)
            for (int j = 0; j < 49; ++j) if (m.joint_map[j] == s) a += d_joints49[((size_t)b * 49 + j) * 3 + c];
        d54[s][c] = a;
    }
    __syncthreads();
    float* dv = d_verts + (size_t)b * NV * 3;
    if (tid < 24 * 3) d_posed_joints[(size_t)b * 72 + tid] = d54[tid / 3][tid % 3];
    if (tid < R * 3) {
        const int r = tid / 3, c = tid % 3;
        d_regd[((size_t)b * R + r) * 3 + c] = r < 9 ? d54[45 + r][c] : d_smpl_joints45[((size_t)b * 45 + (r - 9)) * 3 + c];
    }
    if (tid < 21 * 3) {
        const int i = tid / 3, c = tid % 3;
        float a = d54[24 + i][c];
        if (d_smpl_joints45) a += d_smpl_joints45[((size_t)b * 45 + 24 + i) * 3 + c];
        dv[3 * m.extra_vertex_ids[i] + c] += a;
    }
    __syncthreads();                                    // markers may pick the same vertices as the joint selector: ordered second
    if (d_markers)
        for (int e = tid; e < m.n_markers * 3; e += 256) dv[3 * m.marker_ids[e / 3] + e % 3] += d_markers[(size_t)b * m.n_markers * 3 + e];
}

__global__ __launch_bounds__(128) void smpl_skin_bwd_kernel(const whmr_smpl_model m, const float* __restrict__ betas, long beta_stride,
                                                            const float* __restrict__ A, const float* __restrict__ pose_off,
                                                            const float* __restrict__ d_verts, const float* __restrict__ d_regd, int R,
                                                            float* __restrict__ d_vposed, float* __restrict__ dA_partial) {
    __shared__ __attribute__((aligned(16))) float sA[NJ * 12];
    __shared__ float sBeta[10];
    __shared__ float sReg[33 * 3];
    __shared__ float sQ[12][129];
    __shared__ float sW[NJ][129];
    const int tid = threadIdx.x, b = blockIdx.y, v = blockIdx.x * 128 + tid;
    for (int e = tid; e < NJ * 12; e += 128) sA[e] = A[(size_t)b * NJ * 12 + e];
    if (tid < 10) sBeta[tid] = betas[(size_t)b * beta_stride + tid];
    if (tid < R * 3) sReg[tid] = d_regd[(size_t)b * R * 3 + tid];
    __syncthreads();
    float q[12], w[NJ];
#pragma unroll
    for (int e = 0; e < 12; ++e) q[e] = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) w[j] = 0.f;
    if (v < NV) {
        float vp[3];
        {   // v_posed recomputed like the forward: T + S.beta + pose_off
            float s[30];
#pragma unroll
            for (int k = 0; k < 30; ++k) s[k] = m.shapedirs[(size_t)k * NV + v];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float a = 0.f;
#pragma unroll
                for (int l = 0; l < 10; ++l) a = fmaf(s[c * 10 + l], sBeta[l], a);
                vp[c] = m.v_template[3 * v + c] + a + pose_off[((size_t)b * NV + v) * 3 + c];
            }
        }
        float dv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) dv[c] = d_verts[((size_t)b * NV + v) * 3 + c];
        for (int r = 0; r < R; ++r) {                    // regressed joints: d_verts += reg^T . d_regd
            const float g = m.J_regressor_extra[(size_t)r * NV + v];
            dv[0] = fmaf(g, sReg[r * 3], dv[0]); dv[1] = fmaf(g, sReg[r * 3 + 1], dv[1]); dv[2] = fmaf(g, sReg[r * 3 + 2], dv[2]);
        }
        float T[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) T[e] = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            w[j] = m.lbs_weights[(size_t)j * NV + v];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) T[r * 3 + c] = fmaf(w[j], sA[j * 12 + r * 4 + c], T[r * 3 + c]);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)
            d_vposed[((size_t)b * NV + v) * 3 + c] = T[c] * dv[0] + T[3 + c] * dv[1] + T[6 + c] * dv[2];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            q[r * 4] = dv[r] * vp[0]; q[r * 4 + 1] = dv[r] * vp[1]; q[r * 4 + 2] = dv[r] * vp[2]; q[r * 4 + 3] = dv[r];
        }
    }
#pragma unroll
    for (int e = 0; e < 12; ++e) sQ[e][tid] = q[e];
#pragma unroll
    for (int j = 0; j < NJ; ++j) sW[j][tid] = w[j];
    __syncthreads();
    for (int e = tid; e < NJ * 12; e += 128) {
        const int j = e / 12, k = e % 12;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll 8
        for (int u = 0; u < 128; u += 2) { a0 = fmaf(sW[j][u], sQ[k][u], a0); a1 = fmaf(sW[j][u + 1], sQ[k][u + 1], a1); }
        dA_partial[((size_t)b * gridDim.x + blockIdx.x) * (NJ * 12) + e] = a0 + a1;
    }
}

__global__ __launch_bounds__(256) void smpl_chain_bwd_kernel(const whmr_smpl_model m, const float* __restrict__ rotmat, const float* __restrict__ betas,
                                                            long beta_stride, const float* __restrict__ dA_partial, int nparts,
                                                            const float* __restrict__ d_posed_joints, const float* __restrict__ d_pf_beta,
                                                            float* __restrict__ d_rotmat, float* __restrict__ d_betas) {
    __shared__ float sR[NJ][9], sJ[NJ][3], sGr[NJ][9], sGt[NJ][3];
    __shared__ float dA[NJ][12], dGr[NJ][9], dT[NJ][3], dJ[NJ][3], dR[NJ][9];
    // 256 threads: the partial-sum / element-wise phases use all of them, the serial chain steps only lanes 0..11
    const int b = blockIdx.x, lane = threadIdx.x;
    for (int e = lane; e < NJ * 9; e += 256) sR[e / 9][e % 9] = rotmat[(size_t)b * NJ * 9 + e];
    for (int e = lane; e < NJ * 3; e += 256) {
        float acc = 0.f;
        for (int l = 0; l < 10; ++l) acc = fmaf(m.J_shapedirs[e * 10 + l], betas[(size_t)b * beta_stride + l], acc);
        sJ[e / 3][e % 3] = m.J_template[e] + acc;
    }
    for (int e = lane; e < NJ * 12; e += 256) {
        float a = 0.f;
#pragma unroll 6
        for (int p = 0; p < nparts; ++p) a += dA_partial[((size_t)b * nparts + p) * (NJ * 12) + e];
        dA[e / 12][e % 12] = a;
    }
    __syncthreads();
    // forward chain (global rotations / translations)
    if (lane < 9) sGr[0][lane] = sR[0][lane];
    if (lane < 3) sGt[0][lane] = sJ[0][lane];
    __syncthreads();
    for (int i = 1; i < NJ; ++i) {
        const int p = m.parents[i];
        if (lane < 9) {
            const int r = lane / 3, c = lane % 3;
            sGr[i][lane] = sGr[p][r * 3] * sR[i][c] + sGr[p][r * 3 + 1] * sR[i][3 + c] + sGr[p][r * 3 + 2] * sR[i][6 + c];
        } else if (lane < 12) {
            const int r = lane - 9;
            sGt[i][r] = sGr[p][r * 3] * (sJ[i][0] - sJ[p][0]) + sGr[p][r * 3 + 1] * (sJ[i][1] - sJ[p][1]) + sGr[p][r * 3 + 2] * (sJ[i][2] - sJ[p][2]) + sGt[p][r];
        }
        __syncthreads();
    }
    // A_i = [Gr_i | t_i - Gr_i J_i],  posed_joints_i = t_i
    for (int e = lane; e < NJ * 9; e += 256) {
        const int j = e / 9, r = (e % 9) / 3, c = e % 3;
        dGr[j][r * 3 + c] = dA[j][r * 4 + c] - dA[j][r * 4 + 3] * sJ[j][c];
    }
    for (int e = lane; e < NJ * 3; e += 256) {
        const int j = e / 3, c = e % 3;
        dT[j][c] = dA[j][c * 4 + 3] + (d_posed_joints ? d_posed_joints[(size_t)b * 72 + e] : 0.f);
        dJ[j][c] = -(sGr[j][c] * dA[j][3] + sGr[j][3 + c] * dA[j][7] + sGr[j][6 + c] * dA[j][11]);
    }
    __syncthreads();
    for (int i = NJ - 1; i >= 1; --i) {
        const int p = m.parents[i];
        if (lane < 9) {
            const int r = lane / 3, c = lane % 3;
            dR[i][lane] = sGr[p][r] * dGr[i][c] + sGr[p][3 + r] * dGr[i][3 + c] + sGr[p][6 + r] * dGr[i][6 + c];
            dGr[p][lane] += dGr[i][r * 3] * sR[i][c * 3] + dGr[i][r * 3 + 1] * sR[i][c * 3 + 1] + dGr[i][r * 3 + 2] * sR[i][c * 3 + 2]
                          + dT[i][r] * (sJ[i][c] - sJ[p][c]);
        } else if (lane < 12) {
            const int c = lane - 9;
            const float g = sGr[p][c] * dT[i][0] + sGr[p][3 + c] * dT[i][1] + sGr[p][6 + c] * dT[i][2];
            dJ[i][c] += g;
            dJ[p][c] -= g;
            dT[p][c] += dT[i][c];
        }
        __syncthreads();
    }
    if (lane < 9) dR[0][lane] = dGr[0][lane];
    if (lane < 3) dJ[0][lane] += dT[0][lane];
    __syncthreads();
    for (int e = lane; e < NJ * 9; e += 256)
        d_rotmat[(size_t)b * NJ * 9 + e] = dR[e / 9][e % 9] + (e >= 9 ? d_pf_beta[(size_t)b * 217 + e - 9] : 0.f);
    if (lane < 10) {
        float a = d_pf_beta[(size_t)b * 217 + NPF + lane];
        for (int e = 0; e < NJ * 3; ++e) a = fmaf(m.J_shapedirs[e * 10 + lane], dJ[e / 3][e % 3], a);
        d_betas[(size_t)b * 10 + lane] = a;
    }
}

// d_verts [B,6890,3] is read-modify-write (vertex picks are added in place); d_posed_joints [B,24,3]; d_regd [B,R,3] with
// R = 33 when d_smpl_joints45 is given, else 9.
extern "C" int whmr_smpl_joints_bwd(const whmr_smpl_model* m, const float* d_joints49, const float* d_smpl_joints45, const float* d_markers,
                                    int B, float* d_verts, float* d_posed_joints, float* d_regd, void* stream) {
    if (B <= 0 || !d_verts || !d_posed_joints || !d_regd) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(smpl_joints_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, *m, d_joints49, d_smpl_joints45, d_markers, d_verts,
                       d_posed_joints, d_regd, d_smpl_joints45 ? 33 : 9);
    WHMR_CHECK_LAUNCH();
    return 0;
}

// d_vposed [B,20670]; dA_partial [B, 54, 288] (54 = vertex blocks of 128).  R rows of d_regd (0, 9 or 33; regressor rows as in whmr_smpl_joints).
extern "C" int whmr_smpl_skin_bwd(const whmr_smpl_model* m, const float* betas, long beta_stride, const float* A, const float* pose_off,
                                  const float* d_verts, const float* d_regd, int R, int B, float* d_vposed, float* dA_partial, void* stream) {
    if (B <= 0 || !pose_off || (R != 0 && R != 9 && R != 33)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(smpl_skin_bwd_kernel, dim3(SKIN_BWD_BLOCKS, B), dim3(128), 0, (hipStream_t)stream, *m, betas, beta_stride, A, pose_off, d_verts,
                       d_regd, R, d_vposed, dA_partial);
    WHMR_CHECK_LAUNCH();
    return 0;
}

// d_pf_beta [B,217] = d_vposed . [posedirs (207 rows) ; shapedirs as [10, 20670]]^T (whmr_gemm_f32).
extern "C" int whmr_smpl_chain_bwd(const whmr_smpl_model* m, const float* rotmat, const float* betas, long beta_stride, const float* dA_partial,
                                   const float* d_posed_joints, const float* d_pf_beta, int B, float* d_rotmat, float* d_betas, void* stream) {
    if (B <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(smpl_chain_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, *m, rotmat, betas, beta_stride, dA_partial, SKIN_BWD_BLOCKS,
                       d_posed_joints, d_pf_beta, d_rotmat, d_betas);
    WHMR_CHECK_LAUNCH();
    return 0;
}


// =====================================================================================================================
// Mesh down-sampling sub_verts = Dmap0 . verts, temp_verts = Dmap1 . sub_verts (models/whmr.py:95-98,182-183).  The reference densifies the
// sparse down-sampling matrices of data/mesh_downsampling.npz (1723 x 6890 and 431 x 1723 with ~3 non-zeros per row) and multiplies
// 47.5 MB of zeros per call; here the dense buffer is compressed once per weight version (CSR, built by the host wrapper) and applied as a
// gather: out[b, r, :] = sum_k val[k] * in[b, col[k], :], k in [ptr[r], ptr[r+1]).  The backward is the same kernel on the CSR of D^T.
__global__ __launch_bounds__(256) void csr_apply3_kernel(const int32_t* __restrict__ ptr, const int32_t* __restrict__ col, const float* __restrict__ val,
                                                         const float* __restrict__ in, int n_in, float* __restrict__ out, int n_out, int B,
                                                         int accumulate) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;           // (b, r, c)
    if (i >= (long)B * n_out * 3) return;
    const int c = (int)(i % 3), r = (int)((i / 3) % n_out), b = (int)(i / (3L * n_out));
    const float* ib = in + (size_t)b * n_in * 3 + c;
    float a = 0.f;
    for (int k = ptr[r]; k < ptr[r + 1]; ++k) a = fmaf(val[k], ib[(size_t)col[k] * 3], a);
    out[i] = accumulate ? out[i] + a : a;
}

extern "C" int whmr_csr_apply3(const int32_t* ptr, const int32_t* col, const float* val, const float* in, int n_in, float* out, int n_out, int B,
                               int accumulate, void* stream) {
    if (B <= 0 || n_in <= 0 || n_out <= 0) return (int)hipErrorInvalidValue;
    const long n = (long)B * n_out * 3;
    hipLaunchKernelGGL(csr_apply3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ptr, col, val, in, n_in, out, n_out, B,
                       accumulate);
    WHMR_CHECK_LAUNCH();
    return 0;
}


// =====================================================================================================================
// Training form of the regressor tail (whmr.py:142-173): weak-perspective key points kp_2d (geometry.py:289-307), focal length
// s.detach()*h*Tz/2, full-image camera translation from pred_cam.detach() (geometry.py:139-157), full-image key points kp_2d_w, in ONE launch,
// and their backward in one launch.  cfg.TRAIN.STAGE decides which of the two projections differentiates the joints (whmr.py:142-145,156-163):
// stage 1 -> kp_2d, otherwise kp_2d_w.  One wave per image, lane = joint; the camera / Tz gradients are wave sums (fixed order).
__global__ __launch_bounds__(64) void regressor_post_train_fwd_kernel(const float* __restrict__ joints, const float* __restrict__ cam,
                                                                      const float* __restrict__ Tz, const float* __restrict__ bbox_h,
                                                                      const float* __restrict__ center, const float* __restrict__ orig_shape,
                                                                      int J, float focal0, float res_w, float res_h, float* __restrict__ kp2d,
                                                                      float* __restrict__ kp2d_w, float* __restrict__ cam_t_out,
                                                                      float* __restrict__ focal_out) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float s = cam[3 * b], tx = cam[3 * b + 1], ty = cam[3 * b + 2];
    const float h = bbox_h[b], tz = Tz[b];
    const float focal = s * h * tz / 2.f;
    const float H = orig_shape[2 * b], W = orig_shape[2 * b + 1];
    const float ctx = tx + 2.f * (center[2 * b] - W / 2.f) / (s * h);
    const float cty = ty + 2.f * (center[2 * b + 1] - H / 2.f) / (s * h);
    if (lane == 0) {
        cam_t_out[3 * b] = ctx; cam_t_out[3 * b + 1] = cty; cam_t_out[3 * b + 2] = tz;
        focal_out[b] = focal;
    }
    const float tzw = 2.f * focal0 / (res_h * s + 1e-9f);
    const float cxw = W / 2.f, cyw = H / 2.f;
    for (int j = lane; j < J; j += 64) {
        const float* q = joints + ((size_t)b * J + j) * 3;
        const float x = q[0], y = q[1], z = q[2];
        const float zw = z + tzw;
        kp2d[((size_t)b * J + j) * 2] = (focal0 * ((x + tx) / zw)) / (res_w / 2.f);
        kp2d[((size_t)b * J + j) * 2 + 1] = (focal0 * ((y + ty) / zw)) / (res_h / 2.f);
        const float zf = z + tz;
        kp2d_w[((size_t)b * J + j) * 2] = (focal * ((x + ctx) / zf) + cxw) / cxw - 1.f;
        kp2d_w[((size_t)b * J + j) * 2 + 1] = (focal * ((y + cty) / zf) + cyw) / cyw - 1.f;
    }
}

// d_kp2d / d_kp2d_w [B,J,2], d_cam_t [B,3], d_focal [B] (each nullable) -> d_joints [B,J,3], d_cam [B,3], d_Tz [B].
__global__ __launch_bounds__(64) void regressor_post_train_bwd_kernel(const float* __restrict__ joints, const float* __restrict__ cam,
                                                                      const float* __restrict__ Tz, const float* __restrict__ bbox_h,
                                                                      const float* __restrict__ center, const float* __restrict__ orig_shape,
                                                                      int J, float focal0, float res_w, float res_h, int stage,
                                                                      const float* __restrict__ d_kp2d, const float* __restrict__ d_kp2d_w,
                                                                      const float* __restrict__ d_cam_t, const float* __restrict__ d_focal,
                                                                      float* __restrict__ d_joints, float* __restrict__ d_cam,
                                                                      float* __restrict__ d_Tz) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float s = cam[3 * b], tx = cam[3 * b + 1], ty = cam[3 * b + 2];
    const float h = bbox_h[b], tz = Tz[b];
    const float focal = s * h * tz / 2.f;
    const float H = orig_shape[2 * b], W = orig_shape[2 * b + 1];
    const float ctx = tx + 2.f * (center[2 * b] - W / 2.f) / (s * h);
    const float cty = ty + 2.f * (center[2 * b + 1] - H / 2.f) / (s * h);
    const float den = res_h * s + 1e-9f;
    const float tzw = 2.f * focal0 / den;
    const float cxw = W / 2.f, cyw = H / 2.f;
    const float kx = focal0 / (res_w / 2.f), ky = focal0 / (res_h / 2.f);
    float g_tx = 0.f, g_ty = 0.f, g_tzw = 0.f, g_focal = 0.f, g_tz = 0.f;
    for (int j = lane; j < J; j += 64) {
        const float* q = joints + ((size_t)b * J + j) * 3;
        const float x = q[0], y = q[1], z = q[2];
        float dj[3] = {0.f, 0.f, 0.f};
        if (d_kp2d) {
            const float gx = d_kp2d[((size_t)b * J + j) * 2], gy = d_kp2d[((size_t)b * J + j) * 2 + 1];
            const float zw = z + tzw;
            const float du = kx / zw * gx, dv = ky / zw * gy;
            const float dz = -(kx * (x + tx) * gx + ky * (y + ty) * gy) / (zw * zw);
            g_tx += du; g_ty += dv; g_tzw += dz;
            if (stage == 1) { dj[0] += du; dj[1] += dv; dj[2] += dz; }
        }
        if (d_kp2d_w) {
            const float gx = d_kp2d_w[((size_t)b * J + j) * 2], gy = d_kp2d_w[((size_t)b * J + j) * 2 + 1];
            const float zf = z + tz;
            const float ax = (x + ctx) / (zf * cxw), ay = (y + cty) / (zf * cyw);
            g_focal += ax * gx + ay * gy;
            const float da = focal / (zf * cxw) * gx, db = focal / (zf * cyw) * gy;
            const float dz = -focal * (ax * gx + ay * gy) / zf;
            g_tz += dz;
            if (stage != 1) { dj[0] += da; dj[1] += db; dj[2] += dz; }
        }
        d_joints[((size_t)b * J + j) * 3] = dj[0]; d_joints[((size_t)b * J + j) * 3 + 1] = dj[1]; d_joints[((size_t)b * J + j) * 3 + 2] = dj[2];
    }
    g_tx = wave_sum(g_tx); g_ty = wave_sum(g_ty); g_tzw = wave_sum(g_tzw); g_focal = wave_sum(g_focal); g_tz = wave_sum(g_tz);
    if (lane == 0) {
        if (d_focal) g_focal += d_focal[b];
        if (d_cam_t) g_tz += d_cam_t[3 * b + 2];             // cam_t = [f(cam.detach()), f(cam.detach()), Tz]
        d_cam[3 * b] = g_tzw * (-2.f * focal0 * res_h / (den * den));
        d_cam[3 * b + 1] = g_tx;
        d_cam[3 * b + 2] = g_ty;
        d_Tz[b] = g_tz + g_focal * s * h / 2.f;              // focal = s.detach() * h * Tz / 2
    }
}

extern "C" int whmr_regressor_post_train(const float* joints, const float* cam, const float* Tz, const float* bbox_h, const float* center,
                                         const float* orig_shape, int B, int J, float focal0, float res_w, float res_h, float* kp2d,
                                         float* kp2d_w, float* cam_t, float* focal, void* stream) {
    if (B <= 0 || J <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(regressor_post_train_fwd_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, joints, cam, Tz, bbox_h, center, orig_shape, J,
                       focal0, res_w, res_h, kp2d, kp2d_w, cam_t, focal);
    WHMR_CHECK_LAUNCH();
    return 0;
}

extern "C" int whmr_regressor_post_train_bwd(const float* joints, const float* cam, const float* Tz, const float* bbox_h, const float* center,
                                             const float* orig_shape, int B, int J, float focal0, float res_w, float res_h, int stage,
                                             const float* d_kp2d, const float* d_kp2d_w, const float* d_cam_t, const float* d_focal,
                                             float* d_joints, float* d_cam, float* d_Tz, void* stream) {
    if (B <= 0 || J <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(regressor_post_train_bwd_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, joints, cam, Tz, bbox_h, center, orig_shape, J,
                       focal0, res_w, res_h, stage, d_kp2d, d_kp2d_w, d_cam_t, d_focal, d_joints, d_cam, d_Tz);
    WHMR_CHECK_LAUNCH();
    return 0;
}
