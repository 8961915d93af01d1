"""Oracle: the small prior factors that make the window's normal equations positive definite
(reference como/odom/factors/*.py, applied in Mapping.iterate, Mapping.py:809-917).
TEST INFRASTRUCTURE (see oracle/__init__.py).

Every factor here is a per-keyframe linear-Gaussian term with residual r (k), Jacobians
J_T (k,6) w.r.t. the keyframe pose and J_P (k,3m) w.r.t. the keyframe's m landmarks, and
per-row information s (k):   H += [J_T J_P]^T diag(s) [J_T J_P],   g -= [J_T J_P]^T (s r).
"""
import torch

from . import geom


def _accumulate(H, g, pose_inds, p_inds, J_T, J_P, r, s):
    """pose_inds (B,6), p_inds (B,3m), J_T (B,k,6), J_P (B,k,3m), r (B,k), s (B,k)."""
    J = torch.cat((J_T, J_P), dim=-1)
    idx = torch.cat((pose_inds, p_inds), dim=-1)
    Hb = torch.einsum("bki,bk,bkj->bij", J, s, J)
    gb = -torch.einsum("bki,bk->bi", J, s * r)
    D = H.shape[1]
    for b in range(J.shape[0]):
        g.index_add_(0, idx[b], gb[b])
        H.view(-1).index_add_(0, (idx[b][:, None] * D + idx[b][None, :]).reshape(-1), Hb[b].reshape(-1))
    return torch.sum(s * r * r)


def _logz_chain(J_logz, dlogzm_dPw, dlogzm_dTwc):
    """J_logz (B,k,m) -> J_T (B,k,6), J_P (B,k,3m) through logz_m(P_w, T_wc)."""
    B, k, m = J_logz.shape
    J_P = (J_logz[..., None] * dlogzm_dPw[:, None, :, 0, :]).reshape(B, k, 3 * m)
    J_T = J_logz @ dlogzm_dTwc[:, :, 0, :]
    return J_T, J_P


def gp_ml_cost(logzm, log_median, L_mm, dlogzm_dPw, dlogzm_dTwc, p_inds, pose_inds, H, g, sigma):
    """GP marginal-likelihood prior r = L^-1 (logz_m - log median); reference gp_priors.py:7-81."""
    B, m, _ = L_mm.shape
    Linv = torch.linalg.solve_triangular(L_mm, torch.eye(m, dtype=L_mm.dtype).expand(B, m, m), upper=False)
    r = (Linv @ (logzm - log_median))[..., 0]
    J_T, J_P = _logz_chain(Linv, dlogzm_dPw, dlogzm_dTwc)
    s = torch.full_like(r, 1.0 / sigma**2)
    return _accumulate(H, g, pose_inds, p_inds, J_T, J_P, r, s)


def log_depth_prior_first_mean(logzm, logz_mean, dlogzm_dPw, dlogzm_dTwc, first_mask, p_inds, pose_inds, H, g, sigma_first):
    """mode="first_mean" of reference depth_prior.py:7-141: only landmarks first seen in the KF."""
    B, m, _ = logzm.shape
    r = (logzm - logz_mean)[..., 0] * first_mask
    J_T, J_P = _logz_chain(torch.eye(m, dtype=logzm.dtype).expand(B, m, m), dlogzm_dPw, dlogzm_dTwc)
    s = first_mask.to(logzm.dtype) / sigma_first**2
    return _accumulate(H, g, pose_inds, p_inds, J_T, J_P, r, s)


def pixel_prior_first(pm, pm_first, dpm_dPw, dpm_dTwc, first_mask, p_inds, pose_inds, H, g, sigma_first):
    """mode="first" of reference pixel_prior.py:6-130: keep a landmark on its first-observation pixel."""
    B, m, _ = pm.shape
    r = ((pm - pm_first) * first_mask[..., None]).reshape(B, 2 * m)
    J_T = dpm_dTwc.reshape(B, 2 * m, 6)
    J_P = torch.zeros((B, m, 2, m, 3), dtype=pm.dtype)
    ar = torch.arange(m)
    J_P[:, ar, :, ar, :] = dpm_dPw.permute(1, 0, 2, 3)
    J_P = J_P.reshape(B, 2 * m, 3 * m)
    s = (first_mask.to(pm.dtype) / sigma_first**2)[..., None].expand(B, m, 2).reshape(B, 2 * m)
    return _accumulate(H, g, pose_inds, p_inds, J_T, J_P, r, s)


def mean_log_depth_cost(logzm, Kt, mean_prior, dlogzm_dPw, dlogzm_dTwc, p_inds, pose_inds, H, g, sigma):
    """Scale prior on the mean predicted log-depth; reference gp_priors.py:84-150.  Kt (B,n,m)."""
    n = Kt.shape[1]
    r = torch.mean(Kt @ logzm, dim=(1, 2))[:, None] - mean_prior.reshape(-1, 1)
    J_logz = (Kt.sum(1) / n)[:, None, :]
    J_T, J_P = _logz_chain(J_logz, dlogzm_dPw, dlogzm_dTwc)
    s = torch.full_like(r, 1.0 / sigma**2)
    return _accumulate(H, g, pose_inds, p_inds, J_T, J_P, r, s)


def so3_log(R, eps=1e-6):
    tr = R[..., 0, 0] + R[..., 1, 1] + R[..., 2, 2]
    t3 = tr - 3.0
    th = torch.acos(0.5 * (tr - 1))
    mag = torch.where(t3 < -eps, th / (2 * torch.sin(th)), 0.5 - t3 / 12 + t3 * t3 / 60)
    return mag[..., None] * torch.stack((R[..., 2, 1] - R[..., 1, 2], R[..., 0, 2] - R[..., 2, 0], R[..., 1, 0] - R[..., 0, 1]), -1)


def se3_log_ref(T, eps=1e-6):
    """The reference's SE3 log (lie_algebra.py:160-176), reproduced as written (B = 1)."""
    w = so3_log(T[:, :3, :3])
    th = torch.clamp(torch.linalg.norm(w, dim=1), min=eps)
    wn = w / th
    t = T[:, :3, 3]
    wxt = torch.linalg.cross(wn, t)
    v = t - (0.5 * t) * wxt + (1 - th / (2 * torch.tan(0.5 * th))) * torch.linalg.cross(wn, wxt)
    return torch.cat((w, v), -1)


def pose_prior(pose, meas, H, g, i0, sigma):
    """reference pose_prior_factors.py:5-19."""
    xi = -se3_log_ref(geom.invert_pose(pose) @ meas)[0]
    info_sqrt = 1.0 / sigma
    # the reference builds J = info_sqrt * eye(6) in float32 (torch.eye default dtype), so J^T J is a float32 product
    JtJ = (torch.tensor(info_sqrt, dtype=torch.float32) * torch.eye(6, dtype=torch.float32))
    JtJ = (JtJ.T @ JtJ).to(H.dtype)
    H[i0:i0 + 6, i0:i0 + 6] += JtJ
    r = info_sqrt * xi
    g[i0:i0 + 6] -= info_sqrt * r
    return (r * r).sum()


def scalar_prior(x, meas, H, g, inds, sigma):
    """reference scalar_prior_factors.py:4-34 (single scalar or a vector of independent scalars)."""
    info = 1.0 / sigma**2
    r = (x - meas).reshape(-1)
    g[inds] += -info * r
    H[inds, inds] += info
    return info * (r * r).sum()
