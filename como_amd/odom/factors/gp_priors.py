"""GP priors on the inducing log-depths (reference como/odom/factors/gp_priors.py), same signatures."""
import torch

from como_amd.odom.factors.prior_accumulate import accumulate, logz_chain
from como_amd.utils.lin_alg import trsm_lower


_linv_cache = {}


def _linv(L_mm):
    """L_mm^-1 (B,m,m).  L_mm changes only when a keyframe is added, so the triangular inverse is cached per tensor
    (data pointer + version) instead of being recomputed every GN iteration as the reference does (gp_priors.py:22-23)."""
    key = (L_mm.data_ptr(), L_mm._version, tuple(L_mm.shape))
    hit = _linv_cache.get("k")
    if hit is not None and hit[0] == key:
        return hit[1]
    B, m, _ = L_mm.shape
    eye = torch.eye(m, dtype=L_mm.dtype, device=L_mm.device).expand(B, m, m)
    Linv = trsm_lower(L_mm, eye.contiguous()) if L_mm.is_cuda else torch.linalg.solve_triangular(L_mm, eye, upper=False)
    _linv_cache["k"] = (key, Linv)
    return Linv


def gp_ml_cost(logzm, log_median_depths, L_mm, dlogzm_dPw, dlogzm_dTwc, p_inds_batched, pose_inds_batched, H, g, sigma):
    """r = L_mm^-1 (logz_m - log median depth), information 1/sigma^2 (gp_priors.py:7-81)."""
    Linv = _linv(L_mm)
    r = (Linv @ (logzm - log_median_depths))[..., 0]
    J_T, J_P = logz_chain(Linv, dlogzm_dPw, dlogzm_dTwc)
    return accumulate(H, g, pose_inds_batched, p_inds_batched, J_T, J_P, r, torch.full_like(r, 1.0 / sigma**2))


def mean_log_depth_cost(logzm, Knm_Kmminv, mean_log_depth_prior, dlogzm_dPw, dlogzm_dTwc, p_inds_batched,
                        pose_inds_batched, H, g, sigma):
    """Scale prior on the mean predicted log-depth of a keyframe (gp_priors.py:84-150). Knm_Kmminv (B,n,m)."""
    n = Knm_Kmminv.shape[1]
    r = torch.mean(Knm_Kmminv @ logzm, dim=(1, 2))[:, None] - mean_log_depth_prior.reshape(-1, 1)
    J_logz = (Knm_Kmminv.sum(1) / n)[:, None, :]
    J_T, J_P = logz_chain(J_logz, dlogzm_dPw, dlogzm_dTwc)
    return accumulate(H, g, pose_inds_batched, p_inds_batched, J_T, J_P, r, torch.full_like(r, 1.0 / sigma**2))
