"""Oracle: DepthCov covariance-kernel assembly and GP conditioning.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Two kernel formulas exist in the reference
and are NOT bit-identical:
  * native op  como/backend/src/cov_cpu.cpp:17-64 (+ cov_gpu.cu:17-55): the 1e-8 sits inside
    sqrt(1/det + 1e-8); float32 only on CPU          -> cross_cov_native()
  * Python twin como/depth_cov/core/kernels.py:22-88: C = 2 det1^(1/4) det2^(1/4) / sqrt(det + 1e-8),
    coordinate differences cast to float32 (kernels.py:25)  -> cross_cov_py()
"""
import math

import numpy as np
import torch

SQRT3_NATIVE = np.float32(1.73205080757)


def cross_cov_native(x1, E1, x2, E2, scale):
    """float32 restatement of cov_cpu.cpp:17-64.  x (B,N,2), E (B,N,2,2) -> (B,N,M).

    Every intermediate is float32 as in the C++ (the `1.0/`, `0.5*`, `2.0*` literals are
    double in C++, so those products are evaluated in double and rounded once on store).
    """
    f32, f64 = np.float32, np.float64
    x1, E1, x2, E2 = (np.asarray(a, dtype=f32) for a in (x1, E1, x2, E2))
    dx = x1[:, :, None, 0] - x2[:, None, :, 0]
    dy = x1[:, :, None, 1] - x2[:, None, :, 1]
    e00 = E1[:, :, None, 0, 0] + E2[:, None, :, 0, 0]
    e01 = E1[:, :, None, 0, 1] + E2[:, None, :, 0, 1]
    e11 = E1[:, :, None, 1, 1] + E2[:, None, :, 1, 1]
    det = e00 * e11 - e01 * e01
    det_inv = (f64(1.0) / det.astype(f64)).astype(f32)
    two = f32(2)
    q = (e11 * dx * dx) - two * (e01 * dx * dy) + (e00 * dy * dy)
    q = (q.astype(f64) * (f64(0.5) * det_inv.astype(f64))).astype(f32)
    d1 = (E1[..., 0, 0] * E1[..., 1, 1] - E1[..., 0, 1] * E1[..., 1, 0])[:, :, None]
    d2 = (E2[..., 0, 0] * E2[..., 1, 1] - E2[..., 0, 1] * E2[..., 1, 0])[:, None, :]
    safe = lambda v: np.sqrt(v.astype(f64) + 1e-8).astype(f32)   # float safe_sqrt(float): sqrt(x + 1e-8) in double
    powq = np.power(d1 * d2, f32(0.25)).astype(f32)
    c = ((f64(2.0) * powq.astype(f64)) * safe(det_inv).astype(f64)).astype(f32)
    tmp = (f64(1.73205080757) * safe(q).astype(f64)).astype(f32)
    mat = ((f32(1) + tmp) * np.exp(-tmp)).astype(f32)            # float (1+tmp)*expf(-tmp)
    return torch.from_numpy((f32(scale) * c * mat).astype(f32))


def cross_cov_py(x1, E1, x2, E2, scale):
    """Python-twin kernel (CrossCovarianceModule), kernels.py:22-88 + covariance.py:29-39.
    Works in the dtype of E (float64 in mapping) with the float32 cast of the coordinate diff."""
    diff = (x1[:, :, None, :] - x2[:, None, :, :]).float()
    e00 = E1[..., 0, 0][:, :, None] + E2[..., 0, 0][:, None, :]
    e01 = E1[..., 0, 1][:, :, None] + E2[..., 0, 1][:, None, :]
    e11 = E1[..., 1, 1][:, :, None] + E2[..., 1, 1][:, None, :]
    q = e11 * torch.square(diff[..., 0])
    q = q + (-2 * e01 * diff[..., 0] * diff[..., 1])
    q = q + e00 * torch.square(diff[..., 1])
    det = e00 * e11 - torch.square(e01)
    q = (q / det) * 0.5
    d1 = (E1[..., 0, 0] * E1[..., 1, 1] - E1[..., 0, 1] * E1[..., 1, 0]) ** 0.25
    d2 = (E2[..., 0, 0] * E2[..., 1, 1] - E2[..., 0, 1] * E2[..., 1, 0]) ** 0.25
    c = 2.0 * d1[:, :, None] * d2[:, None, :] / torch.sqrt(det + 1e-8)
    tmp = math.sqrt(3.0) * torch.sqrt(q + 1e-8)
    return ((1 + tmp) * torch.exp(-tmp)) * c * scale


def interp_cov_params(cov_img, coords_norm):
    """Bilinear, border-padded lookup of the (B,4,H,W) covariance image at normalised
    (row, col) coordinates; reference gaussian_kernel.py:52-79.  -> (B,N,2,2)."""
    B, C, H, W = cov_img.shape
    y = ((coords_norm[..., 0] + 1) * H - 1) / 2
    x = ((coords_norm[..., 1] + 1) * W - 1) / 2
    x = x.clamp(0, W - 1)
    y = y.clamp(0, H - 1)
    x0 = torch.floor(x)
    y0 = torch.floor(y)
    wx, wy = x - x0, y - y0
    x0l, y0l = x0.long(), y0.long()
    x1l, y1l = (x0l + 1).clamp(max=W - 1), (y0l + 1).clamp(max=H - 1)
    out = torch.zeros((B, coords_norm.shape[1], C), dtype=cov_img.dtype)
    for b in range(B):
        img = cov_img[b]
        t = (img[:, y0l[b], x0l[b]] * ((1 - wy[b]) * (1 - wx[b])) + img[:, y0l[b], x1l[b]] * ((1 - wy[b]) * wx[b])
             + img[:, y1l[b], x0l[b]] * (wy[b] * (1 - wx[b])) + img[:, y1l[b], x1l[b]] * (wy[b] * wx[b]))
        out[b] = t.T
    return out.reshape(B, -1, 2, 2)


def normalize_coords(px, size):
    """x_norm = 2 A x + A - 1 with A = 1/size; reference utils/coords.py:12-15."""
    A = 1.0 / torch.as_tensor(size, dtype=px.dtype)
    return 2 * A * px + A - 1


def prep_predictor(cov_img, coords_m, scale, photo_size=None):
    """K_mm^-1, L_mm and K~ = K_nm K_mm^-1 for ALL pixels; reference Mapping.py:430-468.
    cov_img (B,4,H,W), coords_m (B,m,2) pixel (row,col).  -> (B,m,m), (B,m,m), (B,H,W,m)."""
    B, _, H, W = cov_img.shape
    Hp, Wp = photo_size or (H, W)
    dt = cov_img.dtype
    m = coords_m.shape[1]
    cm = normalize_coords(coords_m, (H, W))
    Em = interp_cov_params(cov_img, cm)
    rows, cols = torch.meshgrid(torch.arange(Hp), torch.arange(Wp), indexing="ij")
    cn = torch.stack((rows.reshape(-1), cols.reshape(-1)), -1).to(dt)[None].expand(B, -1, -1)
    cn = normalize_coords(cn, (H, W))
    En = interp_cov_params(cov_img, cn)
    Kmm = cross_cov_py(cm, Em, cm, Em, scale)
    # the reference adds the jitter as a float32 tensor (torch.ones default dtype), Mapping.py:450
    Kmm = Kmm + torch.diag_embed((1e-6 * torch.ones(B, m)).to(dt))
    L, _ = torch.linalg.cholesky_ex(Kmm)
    Kinv = torch.cholesky_solve(torch.eye(m, dtype=dt).expand(B, m, m), L)
    Knm = cross_cov_py(cn, En, cm, Em, scale)
    return Kinv, L, (Knm @ Kinv).reshape(B, Hp, Wp, m)


def chol_append(L, obs_info, var, k_ni, k_id, k_ii, N):
    """Rank-1 Cholesky row append + obs_info row + variance downdate, in place.
    reference cov_cpu.cpp:66-85 / cov_gpu.cu:132-215.  L (B,n,n) obs_info (B,n,d) var (B,d)
    k_ni (B,N,1) k_id (B,1,d)."""
    l_ni = torch.linalg.solve_triangular(L[:, :N, :N], k_ni, upper=False)
    l_ii = torch.sqrt(k_ii - (l_ni * l_ni).sum(1, keepdim=True))
    new = (k_id - (l_ni * obs_info[:, :N, :]).sum(1, keepdim=True)) / l_ii
    L[:, N, :N] = l_ni[:, :, 0]
    L[:, N, N] = l_ii[:, 0, 0]
    obs_info[:, N, :] = new[:, 0, :]
    var -= (new * new)[:, 0, :]


def greedy_sampler(cov_img, num, scale, border, dist_thresh, cross_cov=cross_cov_native, append=chol_append):
    """Greedy conditional-entropy inducing-point sampler from scratch (no existing coords,
    no early termination); reference samplers.py:36-107, 143-282 with fixed_var=None.
    cov_img (1,4,H,W) float32.  Returns domain indices (num,), pixel coords (num,2) (row,col)."""
    B, _, H, W = cov_img.shape
    assert B == 1
    rows, cols = torch.meshgrid(torch.arange(border, H - border), torch.arange(border, W - border), indexing="ij")
    dom = torch.stack((rows.reshape(-1), cols.reshape(-1)), -1)
    dn = normalize_coords(dom.float(), (H, W))[None]
    Ed = cov_img[0][:, dom[:, 0], dom[:, 1]].T.reshape(1, -1, 2, 2).contiguous()
    d = dom.shape[0]
    L = torch.eye(num)[None].clone()
    obs = torch.zeros((1, num, d))
    first = torch.argmax(Ed[0, :, 0, 0] * Ed[0, :, 1, 1] - Ed[0, :, 0, 1] * Ed[0, :, 1, 0])
    idx = [int(first)]
    xs = dn[:, idx[0]:idx[0] + 1].clone()
    Es = Ed[:, idx[0]:idx[0] + 1].clone()
    L[:, :1, :1] = torch.linalg.cholesky(cross_cov(xs, Es, xs.clone(), Es.clone(), scale))
    obs[:, :1] = torch.linalg.solve_triangular(L[:, :1, :1], cross_cov(xs, Es, dn, Ed, scale), upper=False)
    var = scale - (obs[:, :1] * obs[:, :1]).sum(1)

    def next_ind(var, chosen):
        sd = torch.sqrt(var)
        sd[sd.isnan()] = 0.0
        sd = sd + 1e-10
        d2 = ((chosen[:, :, None, :] - dn[:, None, :, :]) ** 2).sum(-1)
        ok = (d2 > dist_thresh * dist_thresh).all(1)
        return int(torch.argmax(sd * ok, dim=1))

    best = next_ind(var, xs)
    for i in range(1, num):
        xi, Ei = dn[:, best:best + 1], Ed[:, best:best + 1]
        idx.append(best)
        k_ni = cross_cov(xs, Es, xi, Ei, scale)
        k_id = cross_cov(xi, Ei, dn, Ed, scale)
        xs = torch.cat((xs, xi), 1)
        Es = torch.cat((Es, Ei), 1)
        append(L, obs, var, k_ni, k_id, float(scale), i)
        best = next_ind(var, xs)
    idx = torch.tensor(idx)
    return idx, dom[idx], L, obs, var
