"""Sparse-point correspondence and initialisation for a new keyframe (reference como/odom/frontend/corr.py).

`track_and_init` keeps the reference's signature and return values.  What runs where:
  * reprojection of the m inducing points and of the dense depth image of the previous keyframe: elementwise torch ops on
    the device (O(N) once per keyframe),
  * latent log-depths in the new frame: `distill_depth_from_scratch` / `distill_conditional_depth_from_scratch`
    (kernel matrices from the HIP covariance modules),
  * pruning of the tracked points and sampling of new ones: `sample_sparse_coords` (HIP greedy sampler, csrc/cov.hip),
  * Scharr gradient of the reference log-depth: `ImageGradientModule` (csrc/image.hip).
"""
import torch
import torch.nn.functional as F

from como_amd.depth_cov.core.distill_depth import distill_conditional_depth_from_scratch, distill_depth_from_scratch
from como_amd.depth_cov.core.samplers import sample_sparse_coords
from como_amd.geometry.camera import backprojection, backprojection_points
from como_amd.geometry.lie_algebra import composeSE3, invertSE3
from como_amd.utils.coords import get_test_coords, normalize_coordinates, swap_coords_xy
from como_amd.utils.image_processing import ImageGradientModule


def filter_reproj_coords(coords, P, img_size, min_depth):
    """Keep points at least one pixel inside the image and deeper than min_depth (corr.py:17-29).  coords (1,n,2) row/col."""
    r, c = coords[0, :, 0], coords[0, :, 1]
    keep = (c >= 1) & (c < img_size[-1] - 1) & (r >= 1) & (r < img_size[-2] - 1) & (P[0, :, 2] > min_depth)
    idx = torch.nonzero(keep)[:, 0]                      # index_select: far faster than boolean indexing of a middle dimension
    return coords.index_select(1, idx), P.index_select(1, idx), keep


def reproject_and_filter(coords_i, zi, Tji, K, img_size=None, min_depth=0.0, grid_width=None, want_idx=False, compact=True,
                         host_idx=False):
    """reproject_points (+ filter_reproj_coords when img_size is given) as ONE launch (csrc/trackref.hip
    `como_reproject_points_*`): coords_i (1,n,2) row/col or None with grid_width = W (the points are the pixel grid), zi (1,n,1).
    Returns (coords_j (1,k,2), P_j (1,k,3), keep (n,) bool or None): the kept points in index order.
    compact=False: ALL n points come back (coords_j (1,n,2), P_j (1,n,3), keep (n,) bool) -- the caller carries `keep` as a row
    mask (zero weights) instead of gathering the kept rows: no `nonzero`, no host synchronisation, no 300k-row gathers.
    host_idx: the kept indices are read back as a Python list (the one synchronisation `nonzero` costs anyway) and returned
    as (..., keep, idx device tensor, idx list)."""
    from como_amd import _lib
    dt, dev = zi.dtype, zi.device
    n = zi.shape[1]
    L = _lib.lib()
    rc = torch.empty((1, n, 2), dtype=dt, device=dev)
    P = torch.empty((1, n, 3), dtype=dt, device=dev)
    keep = torch.empty((n,), dtype=torch.uint8, device=dev) if img_size is not None else None
    c = None if coords_i is None else coords_i.to(dt).contiguous()
    h, w = (int(img_size[-2]), int(img_size[-1])) if img_size is not None else (1, 1)
    rcode = getattr(L, "como_reproject_points_" + _lib.suffix(dt))(
        _lib.ptr(c), zi.contiguous().data_ptr(), Tji.to(dt).reshape(4, 4).contiguous().data_ptr(), K[0].to(dt).contiguous().data_ptr(), n,
        int(grid_width or 0), h, w, float(min_depth), rc.data_ptr(), P.data_ptr(), _lib.ptr(keep), _lib.stream_ptr(dev))
    _lib.check(rcode, "como_reproject_points")
    if keep is None:
        return rc, P, None
    keep = keep.view(torch.bool)
    if not compact:
        return rc, P, keep
    if host_idx:
        from como_amd.utils.hostlist import to_device
        idx_list = [j for j, kp in enumerate(keep.tolist()) if kp]
        idx = to_device(idx_list, torch.int64, dev)
        return rc.index_select(1, idx), P.index_select(1, idx), keep, idx, idx_list
    idx = torch.nonzero(keep)[:, 0]
    if want_idx:
        return rc.index_select(1, idx), P.index_select(1, idx), keep, idx
    return rc.index_select(1, idx), P.index_select(1, idx), keep


_MASKED_DENSE = __import__("os").environ.get("COMO_KF_MASKED_DENSE", "1") != "0"     # 0: compact the reprojected dense points (A/B)


def _kernel_path(z):
    return z.is_cuda and z.dtype in (torch.float32, torch.float64) and z.shape[0] == 1


def condition_depth(logz_m, Knm_Kmminv):
    return Knm_Kmminv @ logz_m


def reproject_points(coords_i, zi, Tji, K):
    """Row/col coords + depths of frame i -> row/col coords and camera points in frame j (corr.py:37-43).
    Values only: `transform_points` / `projection` also build the (n,3,6) and (n,2,3) Jacobians nobody reads here -- for the
    307,200 points of the dense depth image that was two batched products over n tiny matrices (0.5 ms each) and ~40 MB of
    stores per call.  The rigid transform is ONE (n,3) x (3,3) product, the projection keeps the reference's operation order
    (camera.py:20-26: f X / Z + c)."""
    Pi, _ = backprojection(K[0], swap_coords_xy(coords_i), zi)
    Pj = Pi @ Tji[:, :3, :3].transpose(-1, -2) + Tji[:, None, :3, 3]
    Kc = K[0]
    rc = torch.stack((Kc[1, 1] * Pj[..., 1] / Pj[..., 2] + Kc[1, 2], Kc[0, 0] * Pj[..., 0] / Pj[..., 2] + Kc[0, 2]), dim=-1)
    return rc, Pj


def get_correspondence_errors(P_reproj, P_new, mode):
    """corr.py:47-59"""
    if mode == "z":
        return torch.abs(P_reproj[..., 2:3] - P_new[..., 2:3])
    if mode in ("logz", "logr"):
        return torch.abs(torch.log(P_reproj[..., 2:3]) - torch.log(P_new[..., 2:3]))
    if mode == "3d":
        return torch.linalg.norm(P_reproj - P_new, dim=-1, keepdim=True)
    raise ValueError(f"unknown correspondence mode {mode!r}")


def _grad_mag(gx, gy):
    """sqrt(gx^2 + gy^2): one launch (csrc/kfglue.hip kg_grad_mag_kernel, the same three roundings per pixel) instead of four."""
    from como_amd.depth_cov.core import distill_depth as _dd
    if (_dd.KF_GLUE and gx.is_cuda and gx.dtype in (torch.float32, torch.float64) and gy.dtype == gx.dtype and gx.shape == gy.shape and
            gx.is_contiguous() and gy.is_contiguous() and gx.numel() > 0):
        from como_amd import _lib
        out = torch.empty_like(gx)
        _lib.check(getattr(_lib.lib(), "como_kf_grad_mag_" + _lib.suffix(gx.dtype))(gx.data_ptr(), gy.data_ptr(), gx.numel(), out.data_ptr(),
                                                                                   _lib.stream_ptr(gx.device)), "como_kf_grad_mag")
        return out
    return torch.sqrt(gx * gx + gy * gy)


def _sample_at(img, coords, size):
    """Bilinear look-up of a (1,1,h,w) image at row/col coords (1,k,2) -> (1,k,1) (zeros outside, align_corners=False)."""
    grid = normalize_coordinates(coords, size, swap=True).unsqueeze(1)
    out = F.grid_sample(img, grid, mode="bilinear", padding_mode="zeros", align_corners=False)
    return out.reshape(1, 1, coords.shape[1]).permute(0, 2, 1)


def prepare_track_and_init(pose1, pose2, coords_m1, z_m1, z_img1, K, cov_size, corr_params):
    """The part of `track_and_init` that does not read the new frame's covariance image: reprojection of the previous keyframe's
    sparse points and dense depth into the new frame (corr.py:86-101 of the reference) and the depth-discontinuity measure of the
    reference depth at the kept points (:135-141).  `Mapping.add_keyframe` issues it BEFORE the network, so that the host-side
    synchronisation of the reprojection (the kept points are counted) does not wait for the network, and the network's 0.7 ms of
    GPU time run under the host work that follows.  Returns a dict for `track_and_init(..., prepared=...)`."""
    dev = coords_m1.device
    b = z_img1.shape[0]
    N = z_img1.shape[-2] * z_img1.shape[-1]
    min_d = corr_params["min_obs_depth"]
    Tji = composeSE3(pose2, pose1, 1)
    z_n1 = z_img1.reshape(b, 1, N).permute(0, 2, 1)
    fused = _kernel_path(z_n1) and z_m1.dtype == z_n1.dtype
    idx_list, keep_n = None, None
    if fused:
        # the m sparse points: compacted (their number shapes the small systems), the kept indices ALSO as a host list (this is the
        # keyframe path's first synchronisation either way); the H W dense points: NOT compacted -- `keep_n` travels as a row mask
        cj_m, Pj_m, keep_m, idx_m, idx_list = reproject_and_filter(coords_m1, z_m1, Tji, K, cov_size, min_d, host_idx=True)
        if _MASKED_DENSE:
            cj_n, Pj_n, keep_n = reproject_and_filter(None, z_n1, Tji, K, cov_size, min_d, grid_width=z_img1.shape[-1], compact=False)
        else:
            cj_n, Pj_n, _ = reproject_and_filter(None, z_n1, Tji, K, cov_size, min_d, grid_width=z_img1.shape[-1])
    else:
        coords_n1 = get_test_coords(z_img1.shape[-2:], device=dev, batch_size=b)
        cj_m, Pj_m = reproject_points(coords_m1, z_m1, Tji, K)
        cj_n, Pj_n = reproject_points(coords_n1, z_n1, Tji, K)
        cj_m, Pj_m, keep_m = filter_reproj_coords(cj_m, Pj_m, cov_size, min_d)
        cj_n, Pj_n, _ = filter_reproj_coords(cj_n, Pj_n, cov_size, min_d)
        idx_m = torch.nonzero(keep_m)[:, 0]
    # depth discontinuities of the reference: |grad log z| at the original sparse coordinates that were kept
    gx, gy = ImageGradientModule(channels=1, device=dev, dtype=z_img1.dtype)(torch.log(z_img1))
    grad_ref = _sample_at(_grad_mag(gx, gy), coords_m1.index_select(1, idx_m), cov_size)
    return {"Tji": Tji, "Tij": invertSE3(Tji), "fused": fused, "cj_m": cj_m, "Pj_m": Pj_m, "keep_m": keep_m, "cj_n": cj_n,
            "zj_n": Pj_n[:, :, 2:3], "grad_ref": grad_ref, "z_dtype": z_n1.dtype, "idx_m": idx_m, "idx_m_list": idx_list,
            "keep_n": keep_n}


def track_and_init(pose1, pose2, coords_m1, z_m1, z_img1, cov_params_img2, K, model, corr_params, sampling_params,
                   rgb_img_size, rgb1=None, rgb2=None, prepared=None):
    """corr.py:62-242.  Returns (coords_2, z2, corr_mask, coords_all, z_all):
    the newly sampled points and their depths, which of the m previous points are kept as correspondences, and the full
    inducing set of the new keyframe (kept correspondences first).  prepared: the result of `prepare_track_and_init` on the same
    arguments, when the caller already issued that part."""
    dev = coords_m1.device
    b, _, h, w = cov_params_img2.shape
    if b != 1:
        raise RuntimeError("track_and_init: batch 1 only")
    cov_size = (h, w)
    min_d = corr_params["min_obs_depth"]
    pre = prepared if prepared is not None else prepare_track_and_init(pose1, pose2, coords_m1, z_m1, z_img1, K, cov_size, corr_params)
    cj_m, Pj_m, keep_m, cj_n, zj_n, grad_ref = pre["cj_m"], pre["Pj_m"], pre["keep_m"], pre["cj_n"], pre["zj_n"], pre["grad_ref"]
    keep_n = pre.get("keep_n")                             # row mask of the (uncompacted) dense points, or None (compacted)
    track_and_init.corr_host = None

    # latent depths of the reprojected sparse points under the NEW frame's covariance, from the reprojected dense depths
    logz_m, logz_res = distill_depth_from_scratch(cj_m, cj_n, zj_n, cov_params_img2, model,
                                                  distill_with_prior=corr_params["distill_with_prior"], min_depth=min_d,
                                                  obs_mask=keep_n, masked_residual=keep_n is not None)
    z_m = torch.exp(logz_m)
    P_m = backprojection_points(K[0], swap_coords_xy(cj_m), z_m)

    # back into frame 1: compare with the interpolated reference depth there
    if pre["fused"] and z_m.dtype == pre["z_dtype"] and cj_m.shape[1] > 0:
        ci_m, Pi_m, _ = reproject_and_filter(cj_m, z_m, pre["Tij"], K)
    else:
        ci_m, Pi_m = reproject_points(cj_m, z_m, pre["Tij"], K)
    P_proj = backprojection_points(K[0], swap_coords_xy(ci_m), _sample_at(z_img1, ci_m, cov_size))

    mode = corr_params["corr_mode"]
    from como_amd.depth_cov.core import distill_depth as _dd
    gr = grad_ref.reshape(-1)
    pts = (P_proj, Pi_m, Pj_m, P_m)
    if (_dd.KF_GLUE and mode in ("logz", "logr") and gr.is_contiguous() and gr.dtype == torch.float64 and gr.is_cuda and
            all(t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and t.shape == (1, gr.shape[0], 3) for t in pts)):
        # the two correspondence errors, their maximum and both threshold tests in ONE launch (twelve as torch ops on ~64 points)
        from como_amd import _lib
        good = torch.empty((gr.shape[0],), dtype=torch.bool, device=gr.device)
        _lib.check(_lib.lib().como_kf_corr_good_f64(P_proj.data_ptr() + 16, Pi_m.data_ptr() + 16, Pj_m.data_ptr() + 16,
                                                    P_m.data_ptr() + 16, 3, gr.data_ptr(), gr.shape[0],
                                                    float(corr_params["corr_thresh"]), float(corr_params["logz_grad_mag_thresh"]),
                                                    good.data_ptr(), _lib.stream_ptr(gr.device)), "como_kf_corr_good_f64")
    else:
        err = torch.maximum(get_correspondence_errors(P_proj, Pi_m, mode), get_correspondence_errors(Pj_m, P_m, mode))
        good = ((err < corr_params["corr_thresh"]) & (grad_ref < corr_params["logz_grad_mag_thresh"]))[0, :, 0]

    # (index lists instead of boolean masks from here on: every boolean-mask selection / assignment synchronises with the host to
    # learn its size -- one read-back does, the rest are gathers and scatters with the same element order; the read-back returns
    # the VALUES, so that the caller can keep its landmark bookkeeping on the host: `track_and_init.corr_host`)
    from como_amd.utils.hostlist import to_device
    host = pre.get("idx_m_list") is not None and good.is_cuda
    if host:
        gi_list = [j for j, g_ in enumerate(good.tolist()) if g_]
        gi = to_device(gi_list, torch.int64, dev)
    else:
        gi = torch.nonzero(good)[:, 0]
        gi_list = None
    coords_1 = cj_m.index_select(1, gi)
    z1 = Pj_m[:, :, 2:3].index_select(1, gi)
    n_max = sampling_params["max_num_coords"]
    if coords_1.shape[1] > 0:
        # thin the tracked points with the same greedy criterion (the sampler reorders: only its index set is used)
        with torch.no_grad():
            _, picked = sample_sparse_coords(cov_params_img2, n_max, "greedy_conditional_entropy",
                                             sampling_params["max_stdev_thresh"], border=sampling_params["border"],
                                             terminate_early=True, dist_thresh=sampling_params["dist_thresh"],
                                             signal_var=model.get_scale(-1), fixed_var=sampling_params["fixed_var"],
                                             coords_domain=coords_1)
        picked_list = getattr(sample_sparse_coords, "last_picked_list", None) if host else None
        if picked_list is not None and len(picked_list) == picked.shape[1]:
            ps_list = sorted(picked_list)                    # the picked points in their original order (= a boolean mask's order)
            ps = to_device(ps_list, torch.int64, dev)
            gi_list = [gi_list[p_] for p_ in ps_list]
        else:
            ps = torch.sort(picked[0, :])[0]
            gi_list = None
        coords_1 = coords_1.index_select(1, ps)
        z1 = z1.index_select(1, ps)
        gi = gi.index_select(0, ps)
    corr_mask = torch.zeros_like(keep_m)
    if gi_list is not None:
        corr_list = [False] * int(keep_m.shape[0])
        for j in gi_list:
            corr_list[pre["idx_m_list"][j]] = True
        track_and_init.corr_host = corr_list                 # which of the previous keyframe's m points stay (host copy)
        corr_mask = to_device(corr_list, torch.bool, dev)
    else:
        corr_mask[pre["idx_m"].index_select(0, gi)] = True

    if coords_1.shape[1] < n_max:
        with torch.no_grad():
            coords_2, _ = sample_sparse_coords(cov_params_img2, n_max, sampling_params["mode"],
                                               sampling_params["max_stdev_thresh"], border=sampling_params["border"],
                                               terminate_early=False, dist_thresh=sampling_params["dist_thresh"],
                                               signal_var=model.get_scale(-1), fixed_var=sampling_params["fixed_var"],
                                               curr_coords=coords_1)
            coords_2 = coords_2.to(dtype=coords_1.dtype)
        coords_all = torch.cat((coords_1, coords_2), dim=1)
        # depths of the new points conditioned on the tracked ones; observation noise = spread of the first fit
        stdev_obs = logz_res.std() if hasattr(logz_res, "okm") else torch.std(logz_res)
        logz_2 = distill_conditional_depth_from_scratch(coords_all, z1, cj_n, cov_params_img2, zj_n, model, min_depth=0.0,
                                                        stdev_obs=stdev_obs, obs_mask=keep_n)
        z2 = torch.exp(logz_2)
        z_all = torch.cat((z1, z2), dim=1)
    else:
        coords_all, z_all = coords_1.clone(), z1.clone()
        coords_2 = torch.empty((1, 0, 2), device=dev, dtype=coords_1.dtype)
        z2 = torch.empty((1, 0, 1), device=dev, dtype=z1.dtype)
    return coords_2, z2, corr_mask, coords_all, z_all
