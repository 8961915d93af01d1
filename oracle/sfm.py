"""Oracle: two-frame SfM photometric system (6-DoF + m log-depths), forward-additive.
reference como/odom/frontend/two_frame_sfm.py:180-285.  TEST INFRASTRUCTURE (see oracle/__init__.py).
"""
import torch

from . import geom


def construct_photo_system(Tji, logz_m, coords_i, vals_i, Kt, img_and_grads_j, K, H, g):
    """reference two_frame_sfm.py:232-269.  Tji (4,4), logz_m (m,), coords_i (N,2) (row,col),
    vals_i (N,) [c channels: (c,N)], Kt (N,m) = K~ rows, img_and_grads_j (3c,H,W); H (6+m,6+m), g (6+m) accumulated.
    Returns dict(err, log_depth, valid, Pi, sigma, r, J)."""
    gray = vals_i.dim() == 1
    if gray:
        vals_i = vals_i[None]
    c = vals_i.shape[0]
    dt = vals_i.dtype
    m = logz_m.shape[0]
    Hh, Ww = img_and_grads_j.shape[-2:]
    logz = Kt @ logz_m                                              # depth.py:22-24
    z = torch.exp(logz)
    ray = torch.stack(((coords_i[:, 1].to(dt) - K[0, 2]) / K[0, 0], (coords_i[:, 0].to(dt) - K[1, 2]) / K[1, 1],
                       torch.ones_like(z)), dim=-1)                 # camera.py:43-54
    Pi = z[:, None] * ray
    Pj = geom.rigid_apply(Tji[None], Pi[None])[0]                   # two_frame_sfm.py:183
    u, v = geom.project(K, Pj)
    valid = geom.in_image(u, v, Hh, Ww) & (Pj[:, 2] > 0)            # two_frame_sfm.py:192-193
    # A_norm is float32 there (1.0 / int tensor, two_frame_sfm.py:187-190) even when everything else is float64
    ax = (1.0 / torch.tensor(float(Ww), dtype=torch.float32)).to(dt)
    ay = (1.0 / torch.tensor(float(Hh), dtype=torch.float32)).to(dt)
    s = geom.bilinear_zeros(img_and_grads_j, u, v, ax, ay)
    I, gx, gy = s[:c].T, s[c:2 * c].T, s[2 * c:].T                  # (N,c) each, two_frame_sfm.py:194-202
    r = I - vals_i.T                                                # two_frame_sfm.py:205
    X, Y, Z = Pj[:, 0:1], Pj[:, 1:2], Pj[:, 2:3]
    fx, fy = K[0, 0], K[1, 1]
    dI_dPj = torch.stack((gx * fx / Z, gy * fy / Z, -(gx * fx * X / Z + gy * fy * Y / Z) / Z), -1)     # (N,c,3)
    R = Tji[:3, :3]
    dPj_dT = torch.cat((-torch.einsum("ij,njk->nik", R, geom.skew(Pi)), R[None].expand(Pi.shape[0], 3, 3)), -1)
    J = torch.empty((Pi.shape[0], c, 6 + m), dtype=dt)
    J[..., :6] = torch.einsum("nck,nkl->ncl", dI_dPj, dPj_dT)
    dI_dPi = dI_dPj @ R
    J[..., 6:] = ((dI_dPi * ray[:, None]).sum(-1) * z[:, None])[..., None] * Kt[:, None]      # dI/dPi . ray z K~[n,:]
    sigma = 1.4826 * torch.median(r[valid].abs())                   # over the valid pixels, all channels
    wr = r * (1.0 / sigma)
    w = torch.where(valid[:, None], geom.huber_weight(wr), torch.zeros_like(wr))
    sw = torch.sqrt(w) * (1.0 / sigma)
    err = torch.sum(torch.square(torch.sqrt(w) * wr))
    Jt, rt = J * sw[..., None], r * sw
    g += -(Jt * rt[..., None]).sum((0, 1))                          # two_frame_sfm.py:219-229
    H += torch.einsum("nck,ncl->kl", Jt, Jt)
    if gray:
        r = r[:, 0]
    return {"err": err, "log_depth": logz, "valid": valid, "Pi": Pi, "sigma": sigma, "r": r, "u": u, "v": v}
