"""Normal-equation helpers, mirroring the reference's como/odom/backend/linear_system.py.

solve_system / update_vars keep the reference signatures (:101-152).  The dense Cholesky runs on the
device through torch.linalg (rocSOLVER) -- D is 8 B + 8 R + 3 L ~ 760..2400; `info` from the
factorisation is kept in `solve_system.last_info` instead of being swallowed.
"""
import torch

import como_amd.geometry.lie_algebra as lie


def row_col_to_lin_index(row, col, width):
    return row * width + col


def accumulate_gradient_scatter(g_batch, grad, grad_inds):
    grad.index_add_(0, grad_inds, g_batch)


def accumulate_hessian_scatter(H_batch, H, H_inds):
    H.view(-1).index_add_(0, H_inds, H_batch)


def get_gradient(J, r):
    return -torch.sum(J * r[..., None], dim=(1, 2))


def get_hessian_diag_block(J):
    return torch.einsum("bnck,bncl->bkl", J, J)


def get_hessian_off_diag_block(J1, J2):
    return torch.einsum("bnck,bncl->bkl", J1, J2)


def landmark_to_batched_3d_point_inds(landmark_inds, num_kf):
    """(nnz,2) [kf, landmark] pairs (row-major nonzero order) -> (num_kf, 3m) indices 3*l + {0,1,2} (:78-89)."""
    lm = landmark_inds[:, 1].view(num_kf, -1)
    p = 3 * lm.repeat_interleave(3, dim=1)
    return p + torch.arange(3, device=lm.device).repeat(lm.shape[1])[None, :]


def solve_system(H, g):
    L, info = torch.linalg.cholesky_ex(H, upper=False, check_errors=False)
    solve_system.last_info = info
    return torch.cholesky_solve(g[:, None], L, upper=False)


def update_vars(delta, kf_poses, kf_aff_params, kf_inds, recent_poses, recent_aff_params, recent_inds, P,
                landmark_ind_start):
    d = delta.squeeze(-1)
    kd = d[kf_inds]
    kf_poses_new = lie.batch_se3(kf_poses, kd[:, :6])
    kf_aff_new = kf_aff_params + kd[:, 6:, None]
    if recent_inds.shape[0] > 0:
        rd = d[recent_inds]
        rp = lie.batch_se3(recent_poses, rd[:, :6])
        ra = recent_aff_params + rd[:, 6:, None]
    else:
        rp = torch.empty((0), device=d.device, dtype=d.dtype)
        ra = torch.empty((0), device=d.device, dtype=d.dtype)
    return kf_poses_new, kf_aff_new, rp, ra, P + d[landmark_ind_start:].view(-1, 3)
