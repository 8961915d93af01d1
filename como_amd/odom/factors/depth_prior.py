"""Log-depth prior towards the (log) median depth (reference como/odom/factors/depth_prior.py:7-141)."""
import torch

from como_amd.odom.factors.prior_accumulate import accumulate, logz_chain


def log_depth_prior(logzm_curr, logzm_mean, dlogzm_dPwm, dlogzm_dTwc, obs_ref_mask, p_inds_batched, pose_inds_batched,
                    H, g, mode, sigma_first=None, sigma_all=None):
    """Modes of the reference: first_mean, first_curr, all_curr, all_mean, first_plus_rest_mean, first_plus_rest_curr."""
    B, m, _ = logzm_curr.shape
    dt = logzm_curr.dtype
    first = obs_ref_mask.to(dt)
    rest = 1.0 - first
    i_first = 1.0 / sigma_first**2 if sigma_first is not None else 0.0
    i_all = 1.0 / sigma_all**2 if sigma_all is not None else 0.0
    r = (logzm_curr - logzm_mean)[..., 0]
    if mode == "first_mean":
        s, rmask = i_first * first, first
    elif mode == "first_curr":
        s, rmask = i_first * first, torch.zeros_like(first)
    elif mode == "all_curr":
        s, rmask = i_all * torch.ones_like(first), torch.zeros_like(first)
    elif mode == "all_mean":
        s, rmask = i_all * torch.ones_like(first), torch.ones_like(first)
    elif mode == "first_plus_rest_mean":
        s, rmask = i_first * first + i_all * rest, torch.ones_like(first)
    elif mode == "first_plus_rest_curr":
        s, rmask = i_first * first + i_all * rest, first
    else:
        raise ValueError("pixel_prior_cost mode: " + mode + " is not implemented.")
    eye = torch.eye(m, dtype=dt, device=logzm_curr.device).expand(B, m, m)
    J_T, J_P = logz_chain(eye, dlogzm_dPwm, dlogzm_dTwc)
    return accumulate(H, g, pose_inds_batched, p_inds_batched, J_T, J_P, r * rmask, s)
