"""Shared accumulation for the small prior factors (reference como/odom/factors/*.py).

Every prior of Mapping.iterate (Mapping.py:809-917) is a per-keyframe linear-Gaussian term: residual r (k),
Jacobian J_T (k,6) w.r.t. the keyframe pose, J_P (k,3m) w.r.t. its m landmarks, per-row information s (k):
    H += [J_T J_P]^T diag(s) [J_T J_P]          g -= [J_T J_P]^T (s r)
O(B m^2) work: batched torch GEMMs + one index_add on the device (not a throughput target).
"""
import torch


def accumulate(H, g, pose_inds, p_inds, J_T, J_P, r, s):
    """pose_inds (B,6) p_inds (B,3m) J_T (B,k,6) J_P (B,k,3m) r (B,k) s (B,k); returns sum s r^2."""
    J = torch.cat((J_T, J_P), dim=-1).to(H.dtype)
    idx = torch.cat((pose_inds, p_inds), dim=-1)
    s = s.to(H.dtype)
    r = r.to(H.dtype)
    Hb = torch.einsum("bki,bk,bkj->bij", J, s, J)
    gb = -torch.einsum("bki,bk->bi", J, s * r)
    D = H.shape[1]
    g.index_add_(0, idx.reshape(-1), gb.reshape(-1))
    lin = (idx[:, :, None] * D + idx[:, None, :]).reshape(-1)
    H.view(-1).index_add_(0, lin, Hb.reshape(-1))
    return torch.sum(s * r * r)


def logz_chain(J_logz, dlogzm_dPw, dlogzm_dTwc):
    """J_logz (B,k,m) -> (J_T (B,k,6), J_P (B,k,3m)) through logz_m(P_w, T_wc)."""
    B, k, m = J_logz.shape
    J_P = (J_logz[..., None] * dlogzm_dPw[:, None, :, 0, :]).reshape(B, k, 3 * m)
    J_T = J_logz @ dlogzm_dTwc[:, :, 0, :]
    return J_T, J_P
