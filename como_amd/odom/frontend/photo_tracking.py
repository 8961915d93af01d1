"""Drop-in for the reference's como/odom/frontend/photo_tracking.py on MI355X.

Same names, argument meaning and return values as the reference functions; the per-pixel work of
`tracking_iter` (warp, sampling, residual, validity mask, exact median, Huber-weighted 8x8 normal
equations, solve, pose update) runs in the HIP kernel chain of csrc/track.hip through the C ABI
(`como_track_iter_*`).  Gray images (c = 1) only; no CPU fallback.
"""
import torch

from como_amd import _lib
from como_amd.geometry.lie_algebra import skew_symmetric

_ws_cache = {}


def _workspace(device, dtype, N):
    key = (str(device), dtype, N)
    ws = _ws_cache.get(key)
    if ws is None:
        L = _lib.lib()
        ws = {
            "r": torch.empty(N, device=device, dtype=dtype),
            "hists": torch.empty(L.como_select_workspace_bytes() // 4, device=device, dtype=torch.int32),
            "partials": torch.empty(L.como_track_partials_bytes() // 8, device=device, dtype=torch.float64),
        }
        _ws_cache.clear()
        _ws_cache[key] = ws
    return ws


def precalc_jacobians(dI_dw, P, vals, intrinsics):
    """Inverse-compositional Jacobians at theta = 0 (reference photo_tracking.py:46-74).
    dI_dw (B,N,c,2), P (B,N,3), vals (B,N,c) -> (B,N,c,8).  Gray CUDA input runs csrc/image.hip."""
    if P.is_cuda and dI_dw.shape[2] == 1:
        B, N = P.shape[:2]
        dt = P.dtype
        J = torch.empty((B, N, 1, 8), dtype=dt, device=P.device)
        fn = getattr(_lib.lib(), "como_track_precalc_jac_" + _lib.suffix(dt))
        _lib.check(fn(dI_dw.to(dt).contiguous().data_ptr(), P.contiguous().data_ptr(), vals.to(dt).contiguous().data_ptr(),
                      intrinsics.to(dt).contiguous().data_ptr(), J.data_ptr(), B * N, _lib.stream_ptr(P.device)),
                   "como_track_precalc_jac")
        return J
    fx, fy = intrinsics[0, 0], intrinsics[1, 1]
    X, Y, Z = P[..., 0], P[..., 1], P[..., 2]
    zero = torch.zeros_like(Z)
    dp_dP = torch.stack((fx / Z, zero, -(fx * X / Z) / Z, zero, fy / Z, -(fy * Y / Z) / Z), dim=-1)
    dp_dP = dp_dP.reshape(P.shape[:-1] + (2, 3))
    eye = torch.eye(3, dtype=P.dtype, device=P.device).expand(P.shape[:-1] + (3, 3))
    dP_dT = torch.cat((-skew_symmetric(P), eye), dim=-1)
    dI_dT = dI_dw @ (dp_dP @ dP_dT)
    return torch.cat((dI_dT, vals.unsqueeze(-1), torch.ones_like(vals).unsqueeze(-1)), dim=-1)


def tracking_iter_raw(Tji, Pi, intrinsics, img_j, aff, vals_i, dI_dT, want_proj=True):
    """Enqueue one GN iteration; returns (out[105], valid u8 (N,), pj (N,2) or None, depth (N,) or None)."""
    _lib.require_cuda(Tji, Pi, intrinsics, img_j, aff, vals_i, dI_dT)
    if img_j.shape[0] != 1 or img_j.shape[1] != 1 or Pi.shape[0] != 1:
        raise RuntimeError("como_amd tracking: batch 1, gray (c = 1) only")
    dt, dev = Pi.dtype, Pi.device
    N = Pi.shape[1]
    H, W = img_j.shape[-2:]
    for t in (Tji, Pi, intrinsics, img_j, aff, vals_i, dI_dT):
        if not t.is_contiguous() or t.dtype != dt:
            raise RuntimeError("como_amd tracking: inputs must be contiguous and share one dtype")
    ws = _workspace(dev, dt, N)
    out = torch.empty(105, device=dev, dtype=dt)
    valid = torch.empty(N, device=dev, dtype=torch.uint8)
    pj = torch.empty((N, 2), device=dev, dtype=dt) if want_proj else None
    depth = torch.empty(N, device=dev, dtype=dt) if want_proj else None
    fn = getattr(_lib.lib(), "como_track_iter_" + _lib.suffix(dt))
    rc = fn(_lib.ptr(Tji), _lib.ptr(intrinsics), _lib.ptr(aff), _lib.ptr(Pi), _lib.ptr(vals_i), _lib.ptr(img_j), H, W, N,
            _lib.ptr(dI_dT), _lib.ptr(ws["r"]), _lib.ptr(valid), _lib.ptr(pj), _lib.ptr(depth), _lib.ptr(ws["hists"]),
            _lib.ptr(ws["partials"]), _lib.ptr(out), _lib.stream_ptr(dev))
    _lib.check(rc, "como_track_iter")
    return out, valid, pj, depth


def tracking_iter(Tji, Pi, intrinsics, img_j, aff, vals_i, dI_dT, photo_sigma, A_norm):
    """reference photo_tracking.py:117-143.  Returns the same 8-tuple:
    (Tji_new (1,4,4), aff_new (1,2,1), delta (1,8,1), mean_sq_err, grad_norm, pj (1,N,2), valid_mask (1,N), depth_j (1,N,1)).
    `photo_sigma` and `A_norm` are accepted for signature parity (the reference ignores photo_sigma too; A_norm = 1/size is
    recomputed in-kernel in the same dtype)."""
    out, valid, pj, depth = tracking_iter_raw(Tji, Pi, intrinsics, img_j, aff, vals_i, dI_dT)
    return (out[80:96].reshape(1, 4, 4), out[96:98].reshape(1, 2, 1), out[72:80].reshape(1, 8, 1), out[98], out[99],
            pj[None], valid[None].bool(), depth[None, :, None])


def photo_level_tracking(Tji_init, aff_init, vals_i, Pi, dI_dT, img_j, intrinsics, photo_sigma, term_criteria):
    """reference photo_tracking.py:147-185 (one host read-back of 4 scalars per iteration for the stop test)."""
    Tji = Tji_init.clone()
    aff = aff_init.clone()
    it = 0
    prev = float("inf")
    while True:
        out, _, _, _ = tracking_iter_raw(Tji, Pi, intrinsics, img_j, aff, vals_i, dI_dT, want_proj=False)
        Tji = out[80:96].reshape(1, 4, 4)
        aff = out[96:98].reshape(1, 2, 1)
        mse, gnorm, dnorm = (float(v) for v in out[[98, 99, 103]].tolist())
        it += 1
        rel = abs((prev - mse) / prev) if prev != float("inf") else float("nan")
        if (it >= term_criteria["max_iter"] or dnorm < term_criteria["delta_norm"] or rel < term_criteria["rel_tol"]
                or gnorm < term_criteria["grad_norm"]):
            break
        prev = mse
    photo_level_tracking.last_iters = it
    return Tji.clone(), aff.clone()


def photo_tracking_pyr(Tji_init, aff_init, vals_i, Pi, dI_dT, masks, intrinsics, img_j, photo_sigma, term_criteria):
    """reference photo_tracking.py:10-42 (lists ordered coarse -> fine)."""
    Tji = Tji_init.clone()
    aff = aff_init.clone()
    for l in range(len(vals_i)):
        mk = masks[l]
        Tji, aff = photo_level_tracking(Tji, aff, vals_i[l][None, mk, :].contiguous(), Pi[l][None, mk, :].contiguous(),
                                        dI_dT[l][None, mk, :, :].contiguous(), img_j[l], intrinsics[l], photo_sigma,
                                        term_criteria)
    return Tji, aff
