"""Oracle: landmark projection, pixel sub-selection and dense reference points
(reference como/odom/backend/sparse_map.py).  TEST INFRASTRUCTURE (see oracle/__init__.py).
"""
import torch

from . import geom


def batched_landmarks(P_all, corr):
    """(L,3), corr (B,L) bool with m True per row -> (B,m,3) in ascending landmark order,
    plus landmark ids (B,m).  reference sparse_map.py:73-112 (no padding: same m per KF)."""
    B = corr.shape[0]
    ids = torch.stack([torch.nonzero(corr[k])[:, 0] for k in range(B)])
    return P_all[ids], ids


def project_landmarks(Twc, Pw, K, reinit_P, median_depths):
    """reference sparse_map.py:18-60.  Twc (B,4,4), Pw (B,m,3)."""
    B, m = Pw.shape[:2]
    Tcw = geom.invert_pose(Twc)
    dTcw_dTwc = -geom.adjoint(Twc)
    Rcw = Tcw[:, :3, :3]

    def to_cam(P):
        Pc = geom.rigid_apply(Tcw, P)
        dPc_dTcw = torch.cat((-torch.einsum("bij,bnjk->bnik", Rcw, geom.skew(P)),
                              Rcw[:, None].expand(B, m, 3, 3)), dim=-1)
        return Pc, dPc_dTcw

    Pc, dPc_dTcw = to_cam(Pw)
    z_mask = Pc[..., 2] < (0.1 * median_depths)[:, None]
    if z_mask.any():
        rPc, rdPc = to_cam(reinit_P)
        Pc = torch.where(z_mask[..., None], rPc, Pc)
        dPc_dTcw = torch.where(z_mask[..., None, None], rdPc, dPc_dTcw)
    z = Pc[..., 2:3]
    logz = torch.log(z)
    dlogz_dz = (1.0 / z)[..., None]                       # (B,m,1,1)
    u, v = geom.project(K, Pc)
    p = torch.stack((u, v), dim=-1)
    X, Y, Z = Pc[..., 0], Pc[..., 1], Pc[..., 2]
    zero = torch.zeros_like(Z)
    dp_dPc = torch.stack((K[0, 0] / Z, zero, -(K[0, 0] * X / Z) / Z,
                          zero, K[1, 1] / Z, -(K[1, 1] * Y / Z) / Z), dim=-1).reshape(B, m, 2, 3)
    dPc_dTwc = torch.einsum("bnij,bjk->bnik", dPc_dTcw, dTcw_dTwc)
    dz_dPw = Rcw[:, None, 2:3, :]                         # (B,1,1,3)
    dz_dTwc = dPc_dTwc[:, :, 2:3, :]                      # (B,m,1,6)
    dp_dPw = torch.einsum("bnij,bjk->bnik", dp_dPc, Rcw)
    dp_dTwc = dp_dPc @ dPc_dTwc
    return p, logz, z_mask, dlogz_dz, dz_dPw, dz_dTwc, dp_dPw, dp_dTwc


def subselect_pixels(img_and_grads, window):
    """Pixel of max gradient magnitude per window x window cell; reference sparse_map.py:116-142.
    Returns coords (B, n, 2) long (row, col).  Ties: first in row-major scan (max_pool2d)."""
    B, c3, H, W = img_and_grads.shape
    c = c3 // 3                                        # gradient norm over all channels, sparse_map.py:127-132
    gn = torch.sqrt(torch.sum(img_and_grads[:, c:2 * c] ** 2 + img_and_grads[:, 2 * c:] ** 2, dim=1))
    _, idx = torch.nn.functional.max_pool2d(gn[:, None], kernel_size=window, return_indices=True)
    idx = idx.reshape(B, -1)
    return torch.stack((idx // W, idx % W), dim=-1)


def dense_reference(logzm, Twc, Kt_rows, coords_n, K, dlogzm_dTwc, dlogzm_dzm):
    """Dense reference points and their Jacobians; reference sparse_map.py:184-194, 212-230.

    logzm (B,m,1), Kt_rows (B,n,m) = K~ rows of the selected pixels, coords_n (B,n,2) (row,col),
    dlogzm_dTwc (B,m,1,6), dlogzm_dzm (B,m,1,1).
    Returns Pwn (B,n,3), dPwn_dTwc (B,n,3,6), dPwn_dzm (B,n,3,m,1), median depth (B,), logz_n.
    """
    dt = logzm.dtype
    logzn = Kt_rows @ logzm                                # (B,n,1)
    zn = torch.exp(logzn)
    ray = torch.stack(((coords_n[..., 1].to(dt) - K[0, 2]) / K[0, 0],
                       (coords_n[..., 0].to(dt) - K[1, 2]) / K[1, 1],
                       torch.ones(coords_n.shape[:2], dtype=dt)), dim=-1)   # (B,n,3)
    Pc = zn * ray
    dPc_dlogzm = (ray * zn)[..., None] * Kt_rows[:, :, None, :]             # (B,n,3,m)
    dPc_dTwc = dPc_dlogzm @ dlogzm_dTwc[:, None, :, 0, :]                   # (B,n,3,6)
    dPc_dzm = dPc_dlogzm * dlogzm_dzm[:, None, None, :, 0, 0]               # (B,n,3,m)
    med = torch.median(Pc[:, :, 2], dim=1).values
    R = Twc[:, :3, :3]
    Pw = geom.rigid_apply(Twc, Pc)
    dPw_dTwc = torch.cat((-torch.einsum("bij,bnjk->bnik", R, geom.skew(Pc)),
                          R[:, None].expand(-1, Pc.shape[1], 3, 3)), dim=-1)
    dPw_dzm = torch.einsum("bij,bnjm->bnim", R, dPc_dzm)[..., None]
    dPw_dTwc_full = dPw_dTwc + torch.einsum("bij,bnjk->bnik", R, dPc_dTwc)
    return Pw, dPw_dTwc_full, dPw_dzm, med, logzn
