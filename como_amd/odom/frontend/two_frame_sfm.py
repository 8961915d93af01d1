"""Two-frame photometric SfM initialisation on the GPU (reference como/odom/frontend/two_frame_sfm.py).

construct_photo_system (:232-269) is expressed through the SAME HIP kernels as the window BA: with T_wi := Tji and
T_wj := I the pair (i -> j) of batch_photo_cost has Pw = Tji Pi = Pj, its reference-pose block is d r / d Tji (right
perturbation) and its depth columns, with invz = 1 and dlogz_m/dT = 0, are d r / d(log-depth codes) = (dI/dPi . ray z) K~.
The system is assembled into a scratch (6 + 2 + 8 + m + 1) layout and the (6 + m) rows / columns are gathered from it.
The priors (:115-177), the solve (:288-293) and the update (:296-303) are O(m^2) torch ops on the device.
"""
import torch

from como_amd import _lib
from como_amd.geometry import lie_algebra as lie
from como_amd.odom.backend import photo
from como_amd.odom.backend.dense_ref import dense_reference_factored
from como_amd.utils.lin_alg import chol_small, cholesky_solve_many, trsm_lower

_tables = {}
_LEGACY_LOOP = __import__("os").environ.get("COMO_SFM_PER_ITERATION", "0") == "1"


def _table(m, device, c=1):
    """Pair arrays of the one (i -> j) pair; c image channels = c entries, one per channel (como_ba_args.channels)."""
    key = (m, str(device), c)
    t = _tables.get(key)
    if t is None:
        i32 = lambda v: torch.tensor(v * c, dtype=torch.int32, device=device)
        D = 16 + m + 1
        lm = torch.full((1, 3 * m), D - 1, dtype=torch.long, device=device)       # d-columns -> 16+k; the two unused
        lm[0, 0::3] = torch.arange(16, 16 + m, device=device)                      # "landmark" axes land in a trash row
        sel = torch.cat((torch.arange(6, device=device), torch.arange(16, 16 + m, device=device)))
        t = {"ref_slot": i32([0]), "ref_aff": i32([0]), "tgt_aff": i32([1]), "tgt_pose": i32([1]),
             "tgt_img": torch.zeros(c, dtype=torch.int64, device=device),
             "pose_ref": torch.arange(8, device=device).reshape(1, 8).repeat(c, 1),
             "pose_tgt": torch.arange(8, 16, device=device).reshape(1, 8).repeat(c, 1),
             "lm": lm.repeat(c, 1), "sel": sel, "D": D,
             "chan": torch.arange(c, dtype=torch.int32, device=device) if c > 1 else None}
        _tables[key] = t
    return t


def photo_statics(test_coords_i, vals_i, Knm_Kmminv, img_and_grads_j, intrinsics):
    """Everything `construct_photo_system` derives from its iteration-independent arguments (one pyramid level of one frame pair):
    built once per level by `two_frame_sfm` instead of once per Gauss-Newton iteration (pixel indices, the value layout the kernels
    read, image / intrinsics in the system's type, the constant pose / affine / Jacobian stubs -- ~25 launches and a host-to-device
    copy per iteration)."""
    dt, dev = Knm_Kmminv.dtype, Knm_Kmminv.device
    N, m = Knm_Kmminv.shape[1], Knm_Kmminv.shape[2]
    Ww = img_and_grads_j.shape[-1]
    c = img_and_grads_j.shape[1] // 3
    poses = torch.zeros((2, 4, 4), dtype=dt, device=dev)
    poses[1] = torch.eye(4, dtype=dt, device=dev)
    K = intrinsics
    ray = torch.stack(((test_coords_i[..., 1].to(dt) - K[0, 2]) / K[0, 0], (test_coords_i[..., 0].to(dt) - K[1, 2]) / K[1, 1],
                       torch.ones((1, N), dtype=dt, device=dev)), dim=-1)
    return {"pixcoord": (test_coords_i[..., 0] * Ww + test_coords_i[..., 1]).to(torch.int32).reshape(1, N).contiguous(),
            "zeros6": torch.zeros((1, m, 6), dtype=dt, device=dev), "Kt": Knm_Kmminv.contiguous(), "poses": poses,
            "aff0": torch.zeros((2, 2), dtype=dt, device=dev), "ones": torch.ones((1, m), dtype=dt, device=dev),
            "dzdP": torch.tensor([[1.0, 0.0, 0.0]], dtype=dt, device=dev),
            "vals": vals_i.reshape(1, c, N).transpose(1, 2).to(dt).contiguous(),        # (1,N,c): the kernels' (slots,n,c) layout
            "img": img_and_grads_j.to(dt).contiguous(), "K": intrinsics.to(dt).contiguous(), "ray": ray}


def photo_points(st, Pj, logz, c, N):
    """The per-point return values of `construct_photo_system` from the most recent linearisation (photo.last_aux):
    (coords_j, depths_j, valid (1,N), Pi (1,N,3)) -- reference two_frame_sfm.py:262-269."""
    aux = photo.last_aux
    valid = aux["valid"].reshape(c, N)[:1].bool()              # the mask does not depend on the channel
    # Pi = Tji^-1 Pj is not needed by the kernels; the reference returns it (two_frame_sfm.py:269) -> rebuild from logz
    Pi = torch.exp(logz.reshape(1, N, 1)) * st["ray"]
    pj = aux["pj"].reshape(c, N, 2)[:1]
    vi = torch.nonzero(valid[0])[:, 0]                          # (one index list for both gathers)
    coords_j = torch.stack((pj[0, :, 1].index_select(0, vi), pj[0, :, 0].index_select(0, vi)), dim=-1)[None]   # swap_coords_xy(pj)[valid]
    depths_j = Pj[0, 2].index_select(0, vi).reshape(1, -1, 1)
    return coords_j, depths_j, valid, Pi


def construct_photo_system(Tji, sparse_log_depth, aff, test_coords_i, vals_i, Knm_Kmminv, img_and_grads_j, intrinsics,
                           photo_sigma, H, g, statics=None, want_points=True):
    """reference two_frame_sfm.py:232-269, same arguments and return tuple
    (total_err, log_depth_i (1,N,1), coords_j, depths_j, valid_mask (1,N), Pi (1,N,3)); H (6+m,6+m), g (6+m) accumulated.
    vals_i (1,c,N) and img_and_grads_j (1,3c,H,W) as the reference passes them (c = 1 gray, 3 rgb: linearize_photo :180-200);
    `aff` and `photo_sigma` are unused there as well.
    statics: `photo_statics(...)` of the same arguments (the Gauss-Newton loop builds it once per level); want_points = False: the
    four per-point values are not extracted (None) and log_depth_i is a view of the kernel's buffer, valid until the next call --
    the loop reads them once, after its last iteration (`photo_points`)."""
    _lib.require_cuda(Tji, sparse_log_depth, test_coords_i, vals_i, Knm_Kmminv, img_and_grads_j, intrinsics, H, g)
    dt, dev = Knm_Kmminv.dtype, Knm_Kmminv.device
    N, m = Knm_Kmminv.shape[1], Knm_Kmminv.shape[2]
    Hh, Ww = img_and_grads_j.shape[-2:]
    c = img_and_grads_j.shape[1] // 3
    if img_and_grads_j.shape[1] != 3 * c or vals_i.numel() != c * N:
        raise RuntimeError("como_amd two_frame_sfm: img_and_grads_j must be (1,3c,H,W) and vals_i (1,c,N)")
    tb = _table(m, dev, c)
    st = statics if statics is not None else photo_statics(test_coords_i, vals_i, Knm_Kmminv, img_and_grads_j, intrinsics)
    Kt = st["Kt"]
    Pj, dPj_dT, uvec, _, logz = dense_reference_factored(sparse_log_depth.reshape(1, m), Tji.reshape(1, 4, 4), Kt, None,
                                                         intrinsics, st["zeros6"], Ww, pixcoord=st["pixcoord"], compact=True)
    poses = st["poses"]
    poses[0].copy_(Tji.reshape(4, 4))
    D = tb["D"]
    # order-independent assembly (exact integer atomics into the fixed-point system buffer, then ONE conversion) -- the
    # float-atomic accumulation into H cost 213 us per call here: every workgroup's 70 x 70 block lands on the same entries
    Lb = _lib.lib()
    fp = tb.get("fix_plane")
    if fp is None:
        fp = tb["fix_plane"] = int(Lb.como_sys_fix_plane_elems(D))
        tb["sysfix"] = torch.zeros((2 * fp,), dtype=torch.int64, device=dev)
        tb["sys64"] = torch.empty((D * D + D + 8,), dtype=torch.float64, device=dev)
    sysfix, sys64 = tb["sysfix"], tb["sys64"]
    sysfix.zero_()
    photo.linearize(dtype=dt, b=c, n=N, m=m, H_img=Hh, W_img=Ww, zmode=2, ref_pose=tb["ref_slot"], Pwn=Pj, vals=st["vals"], channels=c,
                    pair_chan=tb["chan"],
                    dPwn_dTwc=dPj_dT, zjac=Kt, uvec=uvec, pixidx=None, invz=st["ones"], kt_slot_stride=Kt.stride(0), poses_all=poses,
                    aff_all=st["aff0"], img_base=st["img"], K=st["K"],
                    ref_slot=tb["ref_slot"], ref_aff=tb["ref_aff"], tgt_aff=tb["tgt_aff"], tgt_pose=tb["tgt_pose"],
                    tgt_img=tb["tgt_img"], pose_ref_inds=tb["pose_ref"], pose_tgt_inds=tb["pose_tgt"], landmark_inds=tb["lm"],
                    dzdP=st["dzdP"], H=None, g=None, err_out=None, want_pj=True, anorm_f32=True, sysfix=sysfix, fix_plane=fp, D=D)
    Hs, gs, err8 = sys64[:D * D].view(D, D), sys64[D * D:D * D + D], sys64[D * D + D:]
    _lib.check(Lb.como_sys_finalize(sysfix.data_ptr(), fp, D, Hs.data_ptr(), gs.data_ptr(), err8.data_ptr(), _lib.stream_ptr(dev)),
               "como_sys_finalize")
    err = err8[0]
    sel = tb["sel"]
    H += Hs[sel][:, sel].to(H.dtype)
    g += gs[sel].to(g.dtype)
    if not want_points:
        return err.to(dt), logz.reshape(1, N, 1), None, None, None, (st, Pj, logz, c, N)
    coords_j, depths_j, valid, Pi = photo_points(st, Pj, logz, c, N)
    return err.to(dt), logz.reshape(1, N, 1).clone(), coords_j, depths_j, valid, Pi


def linearize_sparse_depth_prior(L_mm):
    """:115-125"""
    B, m, _ = L_mm.shape
    eye = torch.eye(m, device=L_mm.device).unsqueeze(0).repeat(B, 1, 1)
    dr_dd = trsm_lower(L_mm, eye.to(L_mm.dtype))                   # L_mm^-1 (csrc/smallsolve.hip)
    return dr_dd, torch.einsum("hjk,hjl->hkl", dr_dd, dr_dd)


def linearize_mean_log_depth_prior_system(Knm_Kmminv):
    """:128-134"""
    N = Knm_Kmminv.shape[1]
    dr_dd = torch.sum(Knm_Kmminv, dim=1, keepdim=True) / N
    return dr_dd, torch.einsum("hjk,hjl->hkl", dr_dd, dr_dd)


def construct_sparse_depth_prior_system(sparse_log_depth_ref, H, g, dr_dd, H_d_d):
    """:137-148"""
    r = torch.matmul(dr_dd, sparse_log_depth_ref)
    total_err = torch.sum(torch.square(r), dim=(1, 2))
    g[6:] -= torch.sum(dr_dd * r, dim=1).flatten().squeeze(0)
    H[6:, 6:] += H_d_d.squeeze(0)
    return total_err


def construct_mean_log_depth_prior_system(log_depth, H, g, dr_dd, H_d_d, sigma):
    """:151-165"""
    info_sqrt = 1.0 / sigma
    r = torch.mean(log_depth, dim=(1, 2), keepdim=True) * info_sqrt
    total_err = torch.sum(torch.square(r), dim=(1, 2))
    g[6:] -= torch.sum(info_sqrt * dr_dd * r, dim=1).flatten()
    H[6:, 6:] += (info_sqrt * info_sqrt) * torch.block_diag(*H_d_d)
    return total_err


def solve_delta(H, g):
    """:288-293"""
    return chol_small(H, want_L=False, rhs=g[:, None])["X"]        # (6 + m) system: factor + substitutions in one launch


def update_vars(T, sparse_log_depth, aff, delta):
    """:296-303"""
    return lie.batch_se3(T, delta[:6, 0].unsqueeze(0)), sparse_log_depth + delta[6:], aff


def two_frame_sfm(Tji_init, sparse_log_depth_init, aff_init, test_coords_i, vals_i, Knm_Kmminv, img_and_grads_j, dr_prior_dd,
                  H_prior_d_d, intrinsics, sigmas, term_criteria, init_cfg):
    """:306-392 (one host read-back per iteration for the stop test, as the reference)."""
    dev, dt = Tji_init.device, Tji_init.dtype
    m = sparse_log_depth_init.shape[1]
    D = 6 + m
    Tji, d, aff = Tji_init.clone(), sparse_log_depth_init.clone(), aff_init.clone()
    dr_mean_dd, H_mean_d_d = linearize_mean_log_depth_prior_system(Knm_Kmminv)
    # (COMO_SFM_PER_ITERATION=1: round 4's loop -- statics rebuilt, per-point values extracted and two read-backs in EVERY
    # iteration -- for A/B timing, scripts/init_time.py; same numbers either way)
    legacy = _LEGACY_LOOP
    statics = None if legacy else photo_statics(test_coords_i, vals_i, Knm_Kmminv, img_and_grads_j, intrinsics)   # once per level
    it, prev = 0, float("inf")
    while True:
        H = torch.zeros((D, D), device=dev, dtype=dt)
        g = torch.zeros((D,), device=dev, dtype=dt)
        photo_err, log_depth, cj_, dj_, vm_, last = construct_photo_system(
            Tji, d, aff, test_coords_i, vals_i, Knm_Kmminv, img_and_grads_j, intrinsics, sigmas["photo"], H, g, statics=statics,
            want_points=legacy)
        e_d = construct_sparse_depth_prior_system(d, H, g, dr_prior_dd, H_prior_d_d)
        e_m = construct_mean_log_depth_prior_system(log_depth, H, g, dr_mean_dd, H_mean_d_d, sigma=1e0)
        total_t = photo_err + e_d + e_m
        if legacy:
            total = float(total_t)
        delta = solve_delta(H, g)
        Tji, d, aff = update_vars(Tji, d, aff, delta)
        it += 1
        if legacy:
            dn = float(torch.norm(delta[:6]))
        else:
            total, dn = torch.stack((total_t.reshape(-1)[0], torch.norm(delta[:6]).to(total_t.dtype))).tolist()   # ONE read-back
        dec = prev - total
        rel = abs(dec) / prev
        if it >= init_cfg["max_iter"] or dn < init_cfg["delta_norm"] or (rel < init_cfg["rel_tol"] and dec > 0):
            break
        prev = total
    # the per-point values of the LAST linearisation (what the reference's last construct_photo_system call returned)
    mean_log_depth = torch.mean(log_depth, dim=(1, 2), keepdim=True)
    coords_j, depths_j = (cj_, dj_) if legacy else photo_points(*last)[:2]
    two_frame_sfm.last_iters = it
    return Tji, d, aff, coords_j, depths_j, mean_log_depth


def two_frame_sfm_pyr(Tji_init, sparse_log_depth_init, aff_init, test_coords_i, vals_i, Knm_Kmminv, img_and_grads_j,
                      dr_prior_dd, H_prior_d_d, intrinsics, sigmas, term_criteria, init_cfg):
    """:15-52 (lists ordered coarse -> fine)."""
    Tji, d, aff = Tji_init.clone(), sparse_log_depth_init.clone(), aff_init.clone()
    for l in range(len(vals_i)):
        Tji, d, aff, coords_j, depths_j, mld = two_frame_sfm(Tji, d, aff, test_coords_i[l], vals_i[l], Knm_Kmminv[l],
                                                             img_and_grads_j[l], dr_prior_dd, H_prior_d_d, intrinsics[l],
                                                             sigmas, term_criteria, init_cfg)
    return Tji, d, aff, coords_j, depths_j, mld


def setup_reference(img_and_grads, sparse_coords_norm, model, cov_params_img, intrinsics):
    """:55-112.  Per pyramid level (coarse -> fine) of the reference frame: intensities, test coordinates, the predictor
    K~ = K_nm K_mm^-1 of the inducing points (kernel matrices from the HIP covariance modules; no jitter on K_mm here, as
    the reference), plus the level intrinsics and the sparse-depth prior.  Test coordinates are ALL pixels in row-major
    order (the reference draws the same set in a random order -- `torch.multinomial` without replacement over a mask of
    ones -- which only permutes the terms of the sums)."""
    from como_amd.depth_cov.core.gaussian_kernel import interpolate_kernel_params
    from como_amd.utils.coords import get_test_coords, normalize_coordinates
    from como_amd.utils.image_processing import IntrinsicsPyramidModule
    c = img_and_grads[-1].shape[-3] // 3
    dev, dt = img_and_grads[-1].device, img_and_grads[-1].dtype
    intrinsics_pyr = IntrinsicsPyramidModule(0, len(img_and_grads), dev)(intrinsics, [1.0, 1.0])
    E_m = interpolate_kernel_params(cov_params_img, sparse_coords_norm)
    K_mm = model.cov_modules[-1](sparse_coords_norm, E_m)
    L_mm = chol_small(K_mm, want_L=True)["L"]
    dr_prior_dd, H_prior_d_d = linearize_sparse_depth_prior(L_mm)
    vals_pyr, coords_pyr, Kt_pyr, sizes = [], [], [], []
    for lvl in img_and_grads:
        h, w = lvl.shape[-2:]
        coords = get_test_coords((h, w), device=dev, batch_size=1)
        cn = normalize_coordinates(coords.to(dt), (h, w))
        E_n = interpolate_kernel_params(cov_params_img, cn)
        K_nm = model.cross_cov_modules[-1](cn, E_n, sparse_coords_norm, E_m)
        Kt_pyr.append(cholesky_solve_many(K_nm.transpose(-2, -1).contiguous(), L_mm).transpose(-2, -1).contiguous())
        vals_pyr.append(lvl[:, :c].reshape(lvl.shape[0], c, h * w).contiguous())
        coords_pyr.append(coords)
        sizes.append((h, w))
    return vals_pyr, coords_pyr, Kt_pyr, sizes, intrinsics_pyr, dr_prior_dd, H_prior_d_d
