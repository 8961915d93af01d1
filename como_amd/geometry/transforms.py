"""Rigid-transform helpers (reference como/geometry/transforms.py:6-39)."""
import torch

from como_amd.geometry.lie_algebra import composeSE3, invertSE3, skew_symmetric  # noqa: F401


def get_T_w_curr(T_w_ref, T_curr_ref):
    """World pose of the current frame from its pose relative to a reference frame (transforms.py:6-8)."""
    return composeSE3(T_w_ref, T_curr_ref, 2)


def get_rel_pose(pose1, pose2):
    """T_12 = T_w1^-1 T_w2 (transforms.py:11-13)."""
    return composeSE3(pose1, pose2, 1)


def transform_points_values(Tji, Pi):
    """P_j of `transform_points` alone (the same batched product): the keyframe path never reads the Jacobians."""
    R = Tji[:, None, :3, :3].contiguous()
    t = Tji[:, None, :3, 3:4].contiguous()
    return (R @ Pi[..., None] + t).squeeze(-1)


def transform_points(Tji, Pi):
    """P_j = R P_i + t for every point, with dP_j/dT (left se3 perturbation, [rot | trans] columns) and dP_j/dP_i = R
    (transforms.py:17-39).  Pi (b,n,3) or (1,n,3) broadcast over the poses (B,4,4)."""
    R = Tji[:, None, :3, :3].contiguous()
    t = Tji[:, None, :3, 3:4].contiguous()
    Pj = (R @ Pi[..., None] + t).squeeze(-1)
    J = torch.empty((Tji.shape[0], Pi.shape[1], 3, 6), device=Pi.device, dtype=Tji.dtype)
    J[..., :3] = -(R @ skew_symmetric(Pi))
    J[..., 3:] = R
    return Pj, J, R
