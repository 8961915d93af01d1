"""Landmark projection, pixel sub-selection and dense reference points: the mirror of the reference's
como/odom/backend/sparse_map.py (same function names and return tuples).

O(B*m) landmark bookkeeping stays as small torch ops on the device; the per-pixel work of
`setup_test_points` (the (B,n,3,m) Jacobian the reference materialises) has a factored HIP counterpart
`dense_reference_factored` that never builds it.  Mask-feeding arithmetic (pose inverse, rigid transform,
projection) is written with the reference's operation order -- elementwise torch ops round each
multiply/add separately, exactly like its small-matmul CPU kernels.
"""
import torch

import como_amd.geometry.lie_algebra as lie


def _dot3(a0, a1, a2, x, y, z):
    return (a0 * x + a1 * y) + a2 * z


def invert_pose_exact(T):
    """T^-1 with -(R^T t) accumulated in the reference's order (lie_algebra.py:83-93)."""
    R, t = T[:, :3, :3], T[:, :3, 3]
    Ti = torch.zeros_like(T)
    Ti[:, :3, :3] = R.transpose(1, 2)
    for i in range(3):
        Ti[:, i, 3] = -_dot3(R[:, 0, i], R[:, 1, i], R[:, 2, i], t[:, 0], t[:, 1], t[:, 2])
    Ti[:, 3, 3] = 1.0
    return Ti


def rigid_apply_exact(T, P):
    """R P + t per batch with sequential accumulation (transforms.py:17-23). T (b,4,4), P (b,n,3)."""
    return torch.stack([_dot3(T[:, None, i, 0], T[:, None, i, 1], T[:, None, i, 2], P[..., 0], P[..., 1], P[..., 2])
                        + T[:, None, i, 3] for i in range(3)], dim=-1)


def get_batch_remap_function(correspondence_mask):
    """(B,L) bool mask -> (remap(variable, default), [landmark_inds (nnz,2), batch_inds (nnz,2)]); sparse_map.py:73-112."""
    seq = torch.cumsum(correspondence_mask, dim=1, dtype=torch.long) - 1
    lm = torch.nonzero(correspondence_mask)
    bi = torch.stack((lm[:, 0], seq[lm[:, 0], lm[:, 1]]), dim=1)
    max_depth = int(seq[:, -1].max()) + 1

    def remap(variable, default_val=-1):
        out = torch.full((variable.shape[0], max_depth) + tuple(variable.shape[2:]), default_val,
                         device=variable.device, dtype=variable.dtype)
        out[bi[:, 0], bi[:, 1], ...] = variable[lm[:, 0], lm[:, 1], ...]
        return out

    return remap, [lm, bi]


def project_landmarks(Twc, Pw, intrinsics, reinit_P, median_depths):
    """sparse_map.py:18-60.  Returns (p, logz, z_mask, dlogz_dz, dz_dPw, dz_dTwc, dp_dPw, dp_dTwc)."""
    B, m = Pw.shape[:2]
    K = intrinsics
    Tcw = invert_pose_exact(Twc)
    dTcw_dTwc = -lie.adjoint_matrix(Twc)
    Rcw = Tcw[:, :3, :3]

    def to_cam(P):
        Pc = rigid_apply_exact(Tcw, P)
        J = torch.cat((-(Rcw[:, None] @ lie.skew_symmetric(P)), Rcw[:, None].expand(B, m, 3, 3)), dim=-1)
        return Pc, J

    Pc, dPc_dTcw = to_cam(Pw)
    z_mask = Pc[..., 2] < (0.1 * median_depths)[:, None]
    rPc, rJ = to_cam(reinit_P)                      # branch-free (no host sync); identical when nothing is re-initialised
    Pc = torch.where(z_mask[..., None], rPc, Pc)
    dPc_dTcw = torch.where(z_mask[..., None, None], rJ, dPc_dTcw)
    z = Pc[..., 2:3]
    logz = torch.log(z)
    dlogz_dz = (1.0 / z)[..., None]
    X, Y, Z = Pc[..., 0], Pc[..., 1], Pc[..., 2]
    p = torch.stack(((K[0, 0] * X) / Z + K[0, 2], (K[1, 1] * Y) / Z + K[1, 2]), dim=-1)
    zero = torch.zeros_like(Z)
    dp_dPc = torch.stack((K[0, 0] / Z, zero, -(K[0, 0] * X / Z) / Z, zero, K[1, 1] / Z, -(K[1, 1] * Y / Z) / Z),
                         dim=-1).reshape(B, m, 2, 3)
    dPc_dTwc = dPc_dTcw @ dTcw_dTwc[:, None]
    dz_dPw = Rcw[:, None, 2:3, :]
    dz_dTwc = dPc_dTwc[:, :, 2:3, :]
    dp_dPw = dp_dPc @ Rcw[:, None]
    dp_dTwc = dp_dPc @ dPc_dTwc
    return p, logz, z_mask, dlogz_dz, dz_dPw, dz_dTwc, dp_dPw, dp_dTwc


def setup_point_to_frame(Pw_all, Twc, remap_variable_to_batch, K, reinit_P, median_depths):
    """sparse_map.py:197-209."""
    B = Twc.shape[0]
    Pwm = remap_variable_to_batch(Pw_all.unsqueeze(0).expand(B, -1, -1), -1)
    rPwm = remap_variable_to_batch(reinit_P.unsqueeze(0).expand(B, -1, -1), -1)
    return project_landmarks(Twc, Pwm, K[0, ...], rPwm, median_depths)


def subselect_pixels(kf_img_and_grads, photo_window_size, want_pixidx=False):
    """Pixel of maximum gradient magnitude per window (sparse_map.py:116-142).
    Returns coords (B,n,2) long (row, col) and batch_inds (B,n) [and the int32 linear pixel index for
    como_dense_ref's K~ row lookup].  Gray CUDA tensors run csrc/image.hip; colour falls back to the torch formula."""
    B, c3, H, W = kf_img_and_grads.shape
    c = c3 // 3
    if c == 1 and kf_img_and_grads.is_cuda:
        from como_amd import _lib
        x = kf_img_and_grads.contiguous()
        n = (H // photo_window_size) * (W // photo_window_size)
        coords = torch.empty((B, n, 2), dtype=torch.long, device=x.device)
        pixidx = torch.empty((B, n), dtype=torch.int32, device=x.device)
        fn = getattr(_lib.lib(), "como_subselect_pixels_" + _lib.suffix(x.dtype))
        _lib.check(fn(x.data_ptr(), B, H, W, int(photo_window_size), coords.data_ptr(), pixidx.data_ptr(),
                      _lib.stream_ptr(x.device)), "como_subselect_pixels")
    else:
        gn = torch.sqrt(torch.sum(kf_img_and_grads[:, c:2 * c] ** 2 + kf_img_and_grads[:, 2 * c:] ** 2, dim=1))
        _, idx = torch.nn.functional.max_pool2d(gn[:, None], kernel_size=photo_window_size, return_indices=True)
        idx = idx.reshape(B, -1)
        coords = torch.stack((idx // W, idx % W), dim=-1)
        pixidx = idx.to(torch.int32)
    bi = torch.arange(B, device=coords.device)[:, None].expand(-1, coords.shape[1])
    return (coords, bi, pixidx) if want_pixidx else (coords, bi)


def backproject_cloud(logz_m, Knm_Kmminv, coords_n, intrinsics):
    """sparse_map.py:184-194."""
    dt = logz_m.dtype
    logz_n = Knm_Kmminv @ logz_m
    z_n = torch.exp(logz_n)
    K = intrinsics
    ray = torch.stack(((coords_n[..., 1].to(dt) - K[0, 2]) / K[0, 0], (coords_n[..., 0].to(dt) - K[1, 2]) / K[1, 1],
                       torch.ones(coords_n.shape[:2], dtype=dt, device=coords_n.device)), dim=-1)
    Pc_n = z_n * ray
    dlogzn_dlogzm = Knm_Kmminv.unsqueeze(-2)
    dPcn_dlogzm = (ray * z_n)[..., None] * dlogzn_dlogzm
    return Pc_n, dPcn_dlogzm, dlogzn_dlogzm, logz_n


def setup_test_points(pm, logzm, Twc, Knm_Kmminv, coords_n, K, dlogzm_dTwc, dlogzm_dzm):
    """sparse_map.py:212-230 with the reference's return tuple (materialises dPwn_dzm (B,n,3,m,1))."""
    Pc_n, dPcn_dlogzm, dlogzn_dlogzm, logzn = backproject_cloud(logzm, Knm_Kmminv, coords_n, K[0, ...])
    dPcn_dTwc = dPcn_dlogzm @ dlogzm_dTwc[:, None, :, 0, :]
    dPcn_dzm = dPcn_dlogzm * dlogzm_dzm[:, None, None, :, 0, 0]
    median_depths = torch.median(Pc_n[:, :, 2], dim=1).values
    R = Twc[:, :3, :3]
    Pw_n = rigid_apply_exact(Twc, Pc_n)
    n = Pc_n.shape[1]
    dPwn_dTwc = torch.cat((-(R[:, None] @ lie.skew_symmetric(Pc_n)), R[:, None].expand(-1, n, 3, 3)), dim=-1)
    dPwn_dzm = (R[:, None] @ dPcn_dzm)[..., None]
    dPwn_dTwc_full = dPwn_dTwc + R[:, None] @ dPcn_dTwc
    return Pw_n, dPwn_dTwc_full, dPwn_dzm, median_depths, dlogzn_dlogzm, logzn


def dense_reference_factored_torch(logzm, Twc, Kt_full, pixidx, coords_n, K, dlogzm_dTwc):
    """Factored dense reference (no (B,n,3,m) tensor): Pwn, dPwn_dTwc (full), uvec = R_wc ray z_n, median depth.
    Kt_full (B,HW,m), pixidx (B,n) rows.  Small-GEMM torch form; the HIP kernel `como_dense_ref_*` replaces it."""
    dt = logzm.dtype
    B, n = pixidx.shape
    rows = torch.gather(Kt_full, 1, pixidx.long()[..., None].expand(-1, -1, Kt_full.shape[-1]))
    logzn = rows @ logzm
    zn = torch.exp(logzn)
    ray = torch.stack(((coords_n[..., 1].to(dt) - K[0, 2]) / K[0, 0], (coords_n[..., 0].to(dt) - K[1, 2]) / K[1, 1],
                       torch.ones((B, n), dtype=dt, device=logzm.device)), dim=-1)
    Pc = zn * ray
    med = torch.median(Pc[:, :, 2], dim=1).values
    R = Twc[:, :3, :3]
    Pw = rigid_apply_exact(Twc, Pc)
    uvec = (R[:, None] @ (ray * zn)[..., None])[..., 0]
    dl = rows @ dlogzm_dTwc[:, :, 0, :]                                  # (B,n,6) = dlogz_n/dT_wc
    dPw_dTwc = torch.cat((-(R[:, None] @ lie.skew_symmetric(Pc)), R[:, None].expand(-1, n, 3, 3)), dim=-1)
    dPw_dTwc = dPw_dTwc + uvec[..., None] * dl[:, :, None, :]
    return Pw, dPw_dTwc, uvec, med, logzn
