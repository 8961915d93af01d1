"""Oracle: window-BA photometric linearisation (reference como/odom/backend/photo.py).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Materialised per-pixel Jacobian rows,
one batched Gram product per pair, index_add assembly, dense Cholesky -- the same
algorithmic structure the reference runs on its CPU path; also used (timed) as
bench.py's `cpu_baseline` ("port").
"""
import torch

from . import geom


def pair_rows(vals_i, aff_i, Pwn, Twcj, aff_j, img_and_grads_j, dPwn_dTwci, dPwn_dzm, K):
    """Un-robustified residuals and Jacobian rows of every (pair, ref pixel).

    Follows reference photo.py:104-149; c image channels (gray 1, rgb 3: photo.py:24-27, 44-52).
    Shapes: vals_i (b,n,c) aff (b,2,1) Pwn (b,n,3) Twcj (b,4,4) img (b,3c,H,W) = [I | dI/dx | dI/dy]
            dPwn_dTwci (b,n,3,6) dPwn_dzm (b,n,3,m,1).
    Returns r (b,n,c), valid (b,n) bool, J (b,n,c,16+m) with column order
            [ref pose 6, ref aff 2 | target pose 6, target aff 2 | m log-depth cols]; for c = 1 the channel axis is dropped
            (r (b,n), J (b,n,16+m)).
    """
    b, n = vals_i.shape[:2]
    m = dPwn_dzm.shape[3]
    H, W = img_and_grads_j.shape[-2:]
    Tcw = geom.invert_pose(Twcj)                      # photo.py:105, lie_algebra.py:83-95
    Pc = geom.rigid_apply(Tcw, Pwn)                   # photo.py:106, transforms.py:17-23
    u, v = geom.project(K, Pc)                        # photo.py:30, camera.py:20-26
    valid = geom.in_image(u, v, H, W) & (Pc[..., 2] > 0)   # photo.py:15-21
    c = vals_i.shape[2]
    samp = torch.stack([geom.bilinear_zeros(img_and_grads_j[k], u[k], v[k]) for k in range(b)])  # (b,3c,n)
    I_t, gx, gy = (samp[:, k * c:(k + 1) * c].transpose(1, 2) for k in range(3))                 # (b,n,c) each, photo.py:44-52
    # residual, photo.py:114-118 (one affine pair per frame, shared by the channels)
    scale = torch.exp(aff_j[:, 0:1, :] - aff_i[:, 0:1, :])            # (b,1,1)
    Iref_s = scale * vals_i
    r = I_t - Iref_s + (aff_j[:, 1:2, :] - aff_i[:, 1:2, :])
    # dI/dPc = [gx gy] dp/dPc, camera.py:28-35
    X, Y, Z = Pc[..., 0:1], Pc[..., 1:2], Pc[..., 2:3]                # (b,n,1)
    fx, fy = K[0, 0], K[1, 1]
    dI_dPc = torch.stack((gx * fx / Z, gy * fy / Z, -(gx * fx * X / Z + gy * fy * Y / Z) / Z), dim=-1)  # (b,n,c,3)
    Rcw = Tcw[:, :3, :3]
    dI_dPw = torch.einsum("bnck,bkl->bncl", dI_dPc, Rcw)             # photo.py:135
    # target pose: dPc/dTcw = [-Rcw [Pw]x, Rcw]; dTcw/dTwc = -Ad(Twc)  (photo.py:107, transforms.py:25-29)
    dPc_dTcw = torch.cat((-torch.einsum("bij,bnjk->bnik", Rcw, geom.skew(Pwn)),
                          Rcw[:, None].expand(b, n, 3, 3)), dim=-1)  # (b,n,3,6)
    dPc_dTwc = torch.einsum("bnij,bjk->bnik", dPc_dTcw, -geom.adjoint(Twcj))
    J = torch.empty((b, n, c, 16 + m), dtype=vals_i.dtype)
    J[..., 0:6] = torch.einsum("bnck,bnkl->bncl", dI_dPw, dPwn_dTwci)  # photo.py:145
    J[..., 6] = Iref_s                                                 # photo.py:121
    J[..., 7] = -1.0
    J[..., 8:14] = torch.einsum("bnck,bnkl->bncl", dI_dPc, dPc_dTwc)   # photo.py:146
    J[..., 14] = -Iref_s
    J[..., 15] = 1.0
    J[..., 16:] = torch.einsum("bnck,bnkm->bncm", dI_dPw, dPwn_dzm[..., 0])  # photo.py:137-139
    if c == 1:
        return r[..., 0], valid, J[:, :, 0]
    return r, valid, J


def robust_scale(r, valid):
    """sigma = 1.4826 * lower-median(|r| over the valid pixels, all channels of them), photo.py:124-128."""
    return 1.4826 * torch.median(r[valid].abs())


def pair_blocks(r, valid, J, sigma):
    """Huber-whitened Gram blocks per pair: G (b,D,D), gv (b,D) = -J~^T r~, err.

    photo.py:66-80 (robustify) + linear_system.py:24-38 (blocks).
    """
    if r.dim() == 2:                                   # c = 1 without the channel axis
        r, J = r[..., None], J[:, :, None]
    wr = r * (1.0 / sigma)
    w = geom.huber_weight(wr)
    w = torch.where(valid[..., None], w, torch.zeros_like(w))        # one weight per (pixel, channel); photo.py:71-73
    s = torch.sqrt(w) * (1.0 / sigma)
    Jt = J * s[..., None]
    rt = r * s
    G = torch.einsum("bnck,bncl->bkl", Jt, Jt)         # linear_system.py:31-38: summed over pixels AND channels
    gv = -torch.einsum("bnck,bnc->bk", Jt, rt)
    err = torch.sum(torch.square(torch.sqrt(w) * wr))
    return G, gv, err


def assemble(G, gv, dzm_dPwm, pose_ref_inds, pose_target_inds, landmark_inds, H, g):
    """Scatter pair blocks into the dense normal equations (both triangles), photo.py:169-231.

    The m log-depth columns are expanded to 3m landmark coordinates with the per-frame
    constant dz/dP_w (b,1,1,3) ("sleight of hand", photo.py:169-182).
    """
    b = G.shape[0]
    m = G.shape[1] - 16
    dz = dzm_dPwm[:, 0, 0, :]                                   # (b,3)
    Hf = H.view(-1)
    D = H.shape[1]
    for p in range(b):
        ri, ti, li = pose_ref_inds[p], pose_target_inds[p], landmark_inds[p]
        pi = torch.cat((ri, ti))
        g.index_add_(0, pi, gv[p, :16])
        g.index_add_(0, li, (gv[p, 16:, None] * dz[p][None, :]).reshape(-1))
        Hf.index_add_(0, (pi[:, None] * D + pi[None, :]).reshape(-1), G[p, :16, :16].reshape(-1))
        Hpz = (G[p, :16, 16:, None] * dz[p][None, None, :]).reshape(16, 3 * m)
        Hf.index_add_(0, (pi[:, None] * D + li[None, :]).reshape(-1), Hpz.reshape(-1))
        Hf.index_add_(0, (li[:, None] * D + pi[None, :]).reshape(-1), Hpz.T.reshape(-1))
        Hzz = (dz[p][None, :, None, None] * G[p, 16:, None, 16:, None] * dz[p][None, None, None, :]).reshape(3 * m, 3 * m)
        Hf.index_add_(0, (li[:, None] * D + li[None, :]).reshape(-1), Hzz.reshape(-1))


def batch_photo_cost(vals_i, aff_params_i, Pwn, Twcj, aff_params_j, img_and_grads_j, dPwn_dTwci,
                     dPwn_dzm, dzm_dPwm, pose_ref_inds, pose_target_inds, landmark_inds, intrinsics, H, g,
                     return_aux=False):
    """Same contract as reference photo.py:83-233: accumulates into H, g; returns total error."""
    r, valid, J = pair_rows(vals_i, aff_params_i, Pwn, Twcj, aff_params_j, img_and_grads_j,
                            dPwn_dTwci, dPwn_dzm, intrinsics)
    sigma = robust_scale(r, valid)
    G, gv, err = pair_blocks(r, valid, J, sigma)
    assemble(G, gv, dzm_dPwm, pose_ref_inds, pose_target_inds, landmark_inds, H, g)
    if return_aux:
        return err, {"r": r, "valid": valid, "sigma": sigma, "G": G, "gv": gv}
    return err


def consecutive_pairs(B):
    """Forward then backward consecutive-keyframe edges, graph_pair_construction.py:5-15,155-182
    (radius/degree thresholds are 0 in config/como.yml:40-41 -> no extra keyframe edges)."""
    ref = list(range(0, B - 1)) + list(range(1, B))
    tgt = list(range(1, B)) + list(range(0, B - 1))
    return ref, tgt


def one_way_temporal(kf_ts, recent_ts):
    """Temporal one-way neighbours (each recent frame <-> the keyframes it lies between),
    graph_pair_construction.py:97-133."""
    nk, nr = len(kf_ts), len(recent_ts)
    kf_ids, r_ids = [], []
    k = -1
    while recent_ts[0] > kf_ts[k + 1]:
        k += 1
        if k == nk - 1:
            break
    j = 0
    if k < nk - 1:
        while j < nr:
            if recent_ts[j] > kf_ts[k + 1]:
                k += 1
            if k >= nk - 1:
                break
            kf_ids += [k, k + 1]
            r_ids += [j, j]
            j += 1
    while j < nr:
        kf_ids.append(k)
        r_ids.append(j)
        j += 1
    return kf_ids, r_ids


def create_photo_system(kf_poses, kf_aff, recent_poses, recent_aff, Pwn, dPwn_dTwc, dPwn_dzm, dzm_dPwm,
                        vals_n, kf_img_and_grads, recent_img_and_grads, kf_ts, recent_ts, K, H, g,
                        batch_size, kf_inds, recent_inds, landmark_inds):
    """Pair-graph construction + batching, reference photo.py:236-353 (median is per batch)."""
    B = kf_poses.shape[0]
    ref, tgt = consecutive_pairs(B)
    ow_kf, ow_t = ([], [])
    if recent_poses.shape[0] > 0:
        ow_kf, ow_t = one_way_temporal(kf_ts.tolist(), recent_ts.tolist())
    all_ref = ref + ow_kf
    nkf = len(ref)
    err = 0.0
    for b1 in range(0, len(all_ref), batch_size):
        b2 = min(b1 + batch_size, len(all_ref))
        rid = torch.tensor(all_ref[b1:b2])
        kt = torch.tensor(tgt[b1:min(b2, nkf)], dtype=torch.long)
        rt = torch.tensor(ow_t[max(b1, nkf) - nkf:max(b2, nkf) - nkf], dtype=torch.long)
        t_pose = torch.cat((kf_poses[kt], recent_poses[rt])) if rt.numel() else kf_poses[kt]
        t_aff = torch.cat((kf_aff[kt], recent_aff[rt])) if rt.numel() else kf_aff[kt]
        t_img = torch.cat((kf_img_and_grads[kt], recent_img_and_grads[rt])) if rt.numel() else kf_img_and_grads[kt]
        t_ind = torch.cat((kf_inds[kt], recent_inds[rt])) if rt.numel() else kf_inds[kt]
        err = err + batch_photo_cost(vals_n[rid], kf_aff[rid], Pwn[rid], t_pose, t_aff, t_img, dPwn_dTwc[rid],
                                     dPwn_dzm[rid], dzm_dPwm[rid], kf_inds[rid], t_ind, landmark_inds[rid],
                                     K, H, g)
    return err, [ref, tgt], [ow_kf, ow_t]


def solve_system(H, g):
    """reference linear_system.py:101-112 (Cholesky errors ignored there; here info is returned)."""
    L, info = torch.linalg.cholesky_ex(H, upper=False, check_errors=False)
    return torch.cholesky_solve(g[:, None], L, upper=False), info


def update_vars(delta, kf_poses, kf_aff, kf_inds, recent_poses, recent_aff, recent_inds, P, lm_start):
    """reference linear_system.py:115-152."""
    d = delta[:, 0]
    kd = d[kf_inds]
    kf_poses_new = kf_poses @ geom.se3_exp(kd[:, :6])
    kf_aff_new = kf_aff + kd[:, 6:, None]
    if recent_inds.numel() > 0:
        rd = d[recent_inds]
        rp = recent_poses @ geom.se3_exp(rd[:, :6])
        ra = recent_aff + rd[:, 6:, None]
    else:
        rp, ra = recent_poses, recent_aff
    return kf_poses_new, kf_aff_new, rp, ra, P + d[lm_start:].view(-1, 3)
