"""Oracle: inverse-compositional photometric tracking GN
(reference como/odom/frontend/photo_tracking.py).  TEST INFRASTRUCTURE (see oracle/__init__.py).
Gray images (vals (N,), J (N,8), img (H,W)) or c channels (vals (N,c), J (N,c,8), img (c,H,W)).
"""
import torch

from . import geom


def ic_jacobians(dI_dw, P, vals, K):
    """Reference-side Jacobians at theta = 0, photo_tracking.py:46-74.
    dI_dw (N,2) = image gradient at ref pixels, P (N,3), vals (N,) -> (N,8); c channels: dI_dw (N,c,2), vals (N,c) -> (N,c,8)."""
    if vals.dim() == 2:
        return torch.stack([ic_jacobians(dI_dw[:, ch], P, vals[:, ch], K) for ch in range(vals.shape[1])], dim=1)
    X, Y, Z = P[:, 0], P[:, 1], P[:, 2]
    fx, fy = K[0, 0], K[1, 1]
    zero = torch.zeros_like(Z)
    dp_dP = torch.stack((fx / Z, zero, -(fx * X / Z) / Z, zero, fy / Z, -(fy * Y / Z) / Z), -1).reshape(-1, 2, 3)
    dP_dT = torch.cat((-geom.skew(P), torch.eye(3, dtype=P.dtype).expand(P.shape[0], 3, 3)), dim=-1)
    dI_dT = torch.einsum("nk,nkl->nl", dI_dw, dp_dP @ dP_dT)
    return torch.cat((dI_dT, vals[:, None], torch.ones_like(vals)[:, None]), dim=-1)


def project_ic(K, Tji, Pi):
    """transform_project, reference camera.py:57-68: (K T[:3,:]) applied with sequential dot products."""
    Pm = torch.zeros((3, 4), dtype=Pi.dtype)
    for i in range(3):
        for j in range(4):
            Pm[i, j] = geom.dot3_seq(K[i, 0], K[i, 1], K[i, 2], Tji[0, j], Tji[1, j], Tji[2, j])
    ph = [geom.dot3_seq(Pm[i, 0], Pm[i, 1], Pm[i, 2], Pi[:, 0], Pi[:, 1], Pi[:, 2]) + Pm[i, 3] for i in range(3)]
    depth = ph[2]
    return ph[0] / depth, ph[1] / depth, depth


def tracking_iter(Tji, Pi, K, img_j, aff, vals_i, J8):
    """One GN iteration, photo_tracking.py:117-143.
    Tji (4,4), Pi (N,3), img_j (H,W), aff (2,), vals_i (N,), J8 (N,8) (column 6 is overwritten).
    c channels: img_j (c,H,W), vals_i (N,c), J8 (N,c,8); every (pixel, channel) residual enters the median and the sums,
    the mean error divides by the valid PIXELS (photo_tracking.py:83-85).
    Returns dict with T_new, aff_new, delta, mse, grad_norm, u, v, valid, depth, sigma, H, g."""
    gray = img_j.dim() == 2
    if gray:
        img_j, vals_i, J8 = img_j[None], vals_i[:, None], J8[:, None]
    H_, W_ = img_j.shape[-2:]
    u, v, depth = project_ic(K, Tji, Pi)
    valid = geom.in_image(u, v, H_, W_) & (depth > 0)
    It = geom.bilinear_zeros(img_j, u, v).T                                  # (N,c)
    tmp = torch.exp(-aff[0]) * It
    J = J8.clone()
    J[..., 6] = -tmp
    r = (tmp + aff[1]) - vals_i
    sigma = 1.4826 * torch.median(r[valid].abs())
    wr = r * (1.0 / sigma)
    w = torch.where(valid[:, None], geom.huber_weight(wr), torch.zeros_like(wr))     # photo_tracking.py:77-81
    total = torch.sum(w * wr * wr)
    mse = total / valid.sum()
    JW = J * w[..., None]
    g = (JW * r[..., None]).sum((0, 1))
    Hm = torch.einsum("nck,ncl->kl", JW, J)
    if gray:
        r = r[:, 0]
    L, _ = torch.linalg.cholesky_ex(Hm, check_errors=False)
    delta = torch.cholesky_solve(g[:, None], L)[:, 0]
    T_new = Tji @ geom.se3_exp(-delta[:6])
    aff_new = aff - delta[6:8]
    return {"T": T_new, "aff": aff_new, "delta": delta, "mse": mse, "grad_norm": torch.linalg.norm(g),
            "u": u, "v": v, "valid": valid, "depth": depth, "sigma": sigma, "H": Hm, "g": g, "r": r}


def level_tracking(Tji, aff, vals_i, Pi, J8, img_j, K, term):
    """photo_tracking.py:147-185. Returns T, aff, number of iterations."""
    it = 0
    prev = float("inf")
    while True:
        o = tracking_iter(Tji, Pi, K, img_j, aff, vals_i, J8)
        Tji, aff = o["T"], o["aff"]
        it += 1
        rel = abs((prev - o["mse"].item()) / prev) if prev != float("inf") else float("nan")
        stop = (it >= term["max_iter"] or torch.linalg.norm(o["delta"]).item() < term["delta_norm"]
                or rel < term["rel_tol"] or o["grad_norm"].item() < term["grad_norm"])
        prev = o["mse"].item()
        if stop:
            return Tji, aff, it


def pyramid_tracking(Tji, aff, vals_l, P_l, J_l, masks, K_l, img_l, term):
    """Coarse-to-fine, photo_tracking.py:10-42 (lists ordered coarse -> fine)."""
    iters = []
    for l in range(len(vals_l)):
        mk = masks[l]
        Tji, aff, it = level_tracking(Tji, aff, vals_l[l][mk], P_l[l][mk], J_l[l][mk], img_l[l], K_l[l], term)
        iters.append(it)
    return Tji, aff, iters
