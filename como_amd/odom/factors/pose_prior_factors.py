"""Pose anchor (reference como/odom/factors/pose_prior_factors.py:5-19)."""
import numpy as np
import torch

import como_amd.geometry.lie_algebra as lie


def linearize_pose_prior(pose, meas, H, g, Dpose, sigma):
    info_sqrt = 1.0 / sigma
    xi = -lie.SE3_logmap(lie.invertSE3(pose) @ meas)[0, :]
    # the reference forms J = info_sqrt * eye(6) in float32 (torch.eye default dtype): J^T J is a float32 product.
    # (host-side float32 arithmetic: no host->device copy, so the call is hipGraph-capturable)
    jtj = float(np.float32(info_sqrt) * np.float32(info_sqrt))
    r = info_sqrt * xi
    H[Dpose[0]:Dpose[1], Dpose[0]:Dpose[1]] += jtj * torch.eye(6, dtype=H.dtype, device=H.device)
    g[Dpose[0]:Dpose[1]] -= (info_sqrt * r).to(g.dtype)
    return torch.sum(torch.square(r))
