"""Constants of the hot path (values of core/constants.py:4-8,16-92 and models/smpl.py:57-58 in the reference)."""
FOCAL_LENGTH = 1000.
IMG_NORM_MEAN = [0.485, 0.456, 0.406]
IMG_NORM_STD = [0.229, 0.224, 0.225]

# Index of each of the 49 named joints inside the 54-joint superset (24 SMPL + 21 picked vertices + 9 regressed);
# order = reference JOINT_NAMES (25 OpenPose + 24 dataset joints), values = its JOINT_MAP.
JOINT_MAP_49 = [24, 12, 17, 19, 21, 16, 18, 20, 0, 2, 5, 8, 1, 4, 7,
                25, 26, 27, 28, 29, 30, 31, 32, 33, 34,
                8, 5, 45, 46, 4, 7, 21, 19, 17, 16, 18, 20, 47, 48, 49, 50, 51, 52, 53,
                24, 26, 25, 28, 27]
# smplx vertex_ids['smplh'] in VertexJointSelector order (face, feet, finger tips)
EXTRA_VERTEX_IDS = [332, 6260, 2800, 4071, 583, 3216, 3226, 3387, 6617, 6624, 6787,
                    2746, 2319, 2445, 2556, 2673, 6191, 5782, 5905, 6016, 6133]
SMPL_PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]
H36M_TO_J17 = [6, 5, 4, 1, 2, 3, 16, 15, 14, 11, 12, 13, 8, 10, 0, 7, 9]
H36M_TO_J14 = H36M_TO_J17[:14]
