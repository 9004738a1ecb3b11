"""Oracle restatement of the SMPL forward used by W-HMR.  TEST INFRASTRUCTURE ONLY.

The live reference calls ``pare.models.SMPL`` (pare==0.1, environment.yml:118) which
wraps ``smplx.SMPL`` (smplx==0.1.28, environment.yml:167); neither is vendored, so
this arithmetic is "parity unpinned" (SURVEY 8c).  It follows:
  * the in-tree LBS spec models/smpl_webuser/lbs.py:27-80 and verts.py:39-67
    (blend-shape order, J = Jreg . v_shaped, chain G_i = G_parent . [R_i | J_i - J_parent],
    rest-pose removal lbs.py:51-55, T = A . W^T),
  * the commented wrapper models/smpl.py:61-83 (extra joints + JOINT_MAP gather,
    maps in core/constants.py:16-92),
  * smplx conventions restated from the published package: pose feature =
    (R[1:] - I).view(B,207); posedirs stored [207, 20670]; the kinematic tree;
    VertexJointSelector(vertex_ids['smplh']) appending 21 vertices.
Call sites in the reference: models/whmr.py:132-137,227-232,641-644 (pose2rot=False).
"""
import torch

NUM_VERTS = 6890
NUM_JOINTS = 24
PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]

# smplx.vertex_ids.vertex_ids['smplh'] in VertexJointSelector order:
# face (nose, reye, leye, rear, lear), feet (L big/small/heel, R big/small/heel),
# then finger tips l/r x (thumb, index, middle, ring, pinky).
EXTRA_VERTEX_IDS = [332, 6260, 2800, 4071, 583,
                    3216, 3226, 3387, 6617, 6624, 6787,
                    2746, 2319, 2445, 2556, 2673,
                    6191, 5782, 5905, 6016, 6133]

# core/constants.py:16-92: JOINT_NAMES order -> index into the 54-joint superset
# (24 SMPL + 21 selected vertices + 9 J_regressor_extra).
JOINT_MAP_49 = [24, 12, 17, 19, 21, 16, 18, 20, 0, 2, 5, 8, 1, 4, 7,
                25, 26, 27, 28, 29, 30, 31, 32, 33, 34,
                8, 5, 45, 46, 4, 7, 21, 19, 17, 16, 18, 20, 47, 48, 49, 50, 51, 52, 53,
                24, 26, 25, 28, 27]

H36M_TO_J14 = [6, 5, 4, 1, 2, 3, 16, 15, 14, 11, 12, 13, 8, 10]   # models/smpl.py:57-58


def lbs(betas, rotmats, m):
    """betas [B,10], rotmats [B,24,3,3] (global orient first) -> verts [B,6890,3], posed joints [B,24,3].

    ``m`` holds v_template [6890,3], shapedirs [6890,3,10], posedirs [207,20670],
    J_regressor [24,6890], lbs_weights [6890,24].
    """
    B = betas.shape[0]
    v_shaped = m['v_template'].unsqueeze(0) + torch.einsum('bl,mkl->bmk', betas, m['shapedirs'])
    J = torch.einsum('bik,ji->bjk', v_shaped, m['J_regressor'])
    eye = torch.eye(3, dtype=betas.dtype)
    pose_feature = (rotmats[:, 1:] - eye).reshape(B, -1)
    v_posed = v_shaped + torch.matmul(pose_feature, m['posedirs']).view(B, -1, 3)

    # kinematic chain (lbs.py:27-56 / smplx batch_rigid_transform)
    rel = J.clone()
    rel[:, 1:] = J[:, 1:] - J[:, PARENTS[1:]]
    local = torch.zeros(B, NUM_JOINTS, 4, 4, dtype=betas.dtype)
    local[:, :, :3, :3] = rotmats
    local[:, :, :3, 3] = rel
    local[:, :, 3, 3] = 1.0
    chain = [local[:, 0]]
    for i in range(1, NUM_JOINTS):
        chain.append(torch.matmul(chain[PARENTS[i]], local[:, i]))
    G = torch.stack(chain, dim=1)
    posed_joints = G[:, :, :3, 3]
    # remove the rest pose: A = G - pack(G . [J;0])   (lbs.py:51-55)
    Jh = torch.cat([J, torch.zeros(B, NUM_JOINTS, 1, dtype=betas.dtype)], dim=2).unsqueeze(-1)
    A = G.clone()
    A[:, :, :, 3:4] = G[:, :, :, 3:4] - torch.matmul(G, Jh)

    T = torch.matmul(m['lbs_weights'].unsqueeze(0).expand(B, -1, -1), A.view(B, NUM_JOINTS, 16)).view(B, -1, 4, 4)
    vh = torch.cat([v_posed, torch.ones(B, v_posed.shape[1], 1, dtype=betas.dtype)], dim=2)
    verts = torch.matmul(T, vh.unsqueeze(-1))[:, :, :3, 0]
    return verts, posed_joints


def vertex_joint_selector(verts, joints24):
    """smplx VertexJointSelector(vertex_ids['smplh']): 24 -> 45 joints (whmr.py:60,187)."""
    return torch.cat([joints24, verts[:, EXTRA_VERTEX_IDS]], dim=1)


def smpl_forward(betas, rotmats, m):
    """pare.models.SMPL.forward(pose2rot=False): vertices [B,6890,3], joints [B,49,3]."""
    verts, j24 = lbs(betas, rotmats, m)
    j45 = vertex_joint_selector(verts, j24)
    extra = torch.einsum('bik,ji->bjk', verts, m['J_regressor_extra'])
    j54 = torch.cat([j45, extra], dim=1)
    return verts, j54[:, JOINT_MAP_49]
