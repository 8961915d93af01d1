"""Latent log-depth distillation at the inducing points (reference como/depth_cov/core/distill_depth.py).

Same functions and arguments; the kernel matrices come from the HIP modules of the model mirror
(`model.cov_modules[-1]`, `cross_cov_modules`, `diagonal_cov_modules`), the m x m solves are torch ops on the device
(once per keyframe, O(n m^2)).
"""
import torch

from como_amd.depth_cov.core.gaussian_kernel import interpolate_kernel_params
from como_amd.utils.coords import normalize_coordinates
from como_amd.utils.lin_alg import chol_small, trsm_lower


def _gram(A, b, chunks=256):
    """A^T A and A^T b for a tall A (B,n,m), n >> m.  As one GEMM the m x m output is a single tile -- one workgroup walking
    the whole K = n dimension (20 ms at n = 300k in float64) -- so the rows are split into `chunks` slabs whose partial
    products run as a batched GEMM and are summed."""
    B, n, m = A.shape
    if n < 16 * chunks:
        return A.mT @ A, A.mT @ b
    per = -(-n // chunks)
    pad = per * chunks - n
    if pad:
        A = torch.cat((A, A.new_zeros((B, pad, m))), dim=1)
        b = torch.cat((b, b.new_zeros((B, pad, b.shape[2]))), dim=1)
    Ac = A.reshape(B * chunks, per, m)
    bc = b.reshape(B * chunks, per, b.shape[2])
    return ((Ac.mT @ Ac).reshape(B, chunks, m, m).sum(1), (Ac.mT @ bc).reshape(B, chunks, m, b.shape[2]).sum(1))


_gram_ws = {}


def gram_weighted(A, w, y, c=None, want_stats=False):
    """Weighted normal equations with the rows read in place (csrc/gram.hip `como_gram_f64`): A (1,n,m) float64 on the GPU (a
    row-strided view is fine), w (1,n[,1]) row weights or None, y (1,n,1), c (1,m[,1]) or None:
        r = y - A c,   AtA = A^T diag(w) A (1,m,m),   Atb = A^T diag(w) r (1,m,1),   stats = {sum w, sum w r, sum w r^2, #(w != 0)}.
    Rows with w == 0 contribute nothing whatever they hold (no boolean-mask gather of the valid rows, no concatenation)."""
    from como_amd import _lib
    _lib.require_cuda(A, y)
    B, n, m = A.shape
    if B != 1 or A.dtype != torch.float64 or m > 64 or m % 4 or A.stride(2) != 1 or A.stride(1) % 2 or A.data_ptr() % 16:
        raise RuntimeError("como_amd gram_weighted: needs a (1,n,m) float64 tensor with m <= 64, m % 4 == 0 and 16-byte aligned rows")
    L = _lib.lib()
    dev = A.device
    key = f"{dev}:{torch.cuda.current_stream(dev).cuda_stream}"     # (one scratch per stream: two mappers on one device must not share it)
    ws = _gram_ws.get(key)
    if ws is None:
        ws = _gram_ws[key] = torch.empty(L.como_gram_workspace_bytes() // 8, dtype=torch.float64, device=dev)
    AtA = torch.empty((1, m, m), dtype=torch.float64, device=dev)
    Atb = torch.empty((1, m, 1), dtype=torch.float64, device=dev)
    stats = torch.empty(4, dtype=torch.float64, device=dev) if want_stats else None
    wv = None if w is None else w.reshape(n).to(torch.float64).contiguous()
    yv = y.reshape(n).to(torch.float64).contiguous()
    cv = None if c is None else c.reshape(m).to(torch.float64).contiguous()
    rc = L.como_gram_f64(A.data_ptr(), A.stride(1), n, m, _lib.ptr(wv), yv.data_ptr(), _lib.ptr(cv), AtA.data_ptr(), Atb.data_ptr(),
                         _lib.ptr(stats), ws.data_ptr(), _lib.stream_ptr(dev))
    _lib.check(rc, "como_gram_f64")
    return (AtA, Atb, stats) if want_stats else (AtA, Atb)


_GRAM_KERNEL = __import__("os").environ.get("COMO_GRAM_KERNEL", "1") != "0"     # 0: gather + concatenate + slab GEMMs (A/B)
# 1 (default): the element-wise chains around the distillation's kernels as ONE launch each (csrc/kfglue.hip: the same operations
# in the same order); 0: the torch chains (A/B, tests)
KF_GLUE = __import__("os").environ.get("COMO_KF_GLUE", "1") != "0"


def _glue_ok(*ts):
    return KF_GLUE and all(t is None or (t.is_cuda and t.is_contiguous()) for t in ts)


def _stdev_ok(stdev_obs):
    """A fixed observation stdev the fused path can take: None, a Python number, or a float64 scalar on the device."""
    return (not torch.is_tensor(stdev_obs)) or (stdev_obs.is_cuda and stdev_obs.dtype == torch.float64 and stdev_obs.numel() == 1)


def distill_prep(z_obs, obs_mask, min_depth, sinv, stdev_obs, weighted, want_zs=False):
    """ok = z > min_depth [& obs_mask]; y = log(ok ? z : 1); w = ok ? s^2 : 0 (weighted) or ok as 0 / 1 -- the validity test of the
    observations as zero weights (distill_depth.py:96-111, 152-166), ONE launch (csrc/kfglue.hip kg_distill_prep_kernel).
    z_obs (1,n,1) float64 (any stride along n: e.g. the depth column of (1,n,3) points), obs_mask (n,) bool or None; s = sinv (1,n,1)
    per row, or -- sinv None -- 1 / stdev_obs for a fixed observation stdev: a Python number, or a float64 scalar ON THE DEVICE
    (the residual spread of an earlier distillation), which is inverted inside the launch: no read-back.
    Returns (okm (1,n,1) bool, zs (1,n,1) or None, y (1,n,1), w (1,n,1))."""
    from como_amd import _lib
    n = z_obs.shape[1]
    dev = z_obs.device
    okm = torch.empty((1, n, 1), dtype=torch.bool, device=dev)
    y = torch.empty((1, n, 1), dtype=torch.float64, device=dev)
    w = torch.empty((1, n, 1), dtype=torch.float64, device=dev)
    zs = torch.empty((1, n, 1), dtype=torch.float64, device=dev) if want_zs else None
    sdev, sval = None, 0.0
    if sinv is None and stdev_obs is not None:
        if torch.is_tensor(stdev_obs):
            sdev = stdev_obs
        else:
            sval = 1.0 / float(stdev_obs)
    _lib.check(_lib.lib().como_kf_distill_prep_f64(z_obs.data_ptr(), int(z_obs.stride(1)) if n > 1 else 1, _lib.ptr(obs_mask), n,
                                                   float(min_depth), _lib.ptr(sinv), sval, _lib.ptr(sdev), 1 if weighted else 0,
                                                   okm.data_ptr(), _lib.ptr(zs), y.data_ptr(), w.data_ptr(), _lib.stream_ptr(dev)),
               "como_kf_distill_prep_f64")
    return okm, zs, y, w


def padded_predictor(Kt):
    """The (B,n,mp) buffer behind a `get_predictor(pad4=True)` result (the tensor itself when no padding was needed)."""
    return getattr(Kt, "_como_padded", Kt)


def _fast(Kt):
    Kt = padded_predictor(Kt)
    return _GRAM_KERNEL and Kt.is_cuda and Kt.dtype == torch.float64 and Kt.shape[0] == 1 and 0 < Kt.shape[2] <= 64 and \
        Kt.shape[2] % 4 == 0 and Kt.stride(2) == 1 and Kt.stride(1) % 2 == 0 and Kt.data_ptr() % 16 == 0


def lstsq_chol(A, b):
    """como/utils/lin_alg.py:82-87: normal equations + Cholesky."""
    AtA, Atb = _gram(A, b)
    return chol_small(AtA, want_L=False, rhs=Atb)["X"]        # factor + both substitutions in one launch


def calc_kernel_matrices(coords_m, coords_n, cov_params_img, model):
    """:8-27 -> K_mm (B,m,m), K_nm (B,n,m), K_nn_diag (B,n)."""
    size = cov_params_img.shape[-2:]
    dt = cov_params_img.dtype
    mods = (model.cov_modules[-1], model.cross_cov_modules[-1], model.diagonal_cov_modules[-1])
    if (cov_params_img.is_cuda and dt in (torch.float32, torch.float64) and coords_m.dtype == dt and coords_n.dtype == dt and
            cov_params_img.shape[1] == 4 and len({float(x.scale) for x in mods}) == 1 and type(mods[2]).__name__ == "DiagonalCovarianceModule"):
        # one native call (csrc/densify.hip `como_kernel_matrices_*`: 5 launches) for ~45 torch launches, value for value
        from como_amd import _lib
        B, m, n = coords_m.shape[0], coords_m.shape[1], coords_n.shape[1]
        dev = cov_params_img.device
        e = lambda *shape: torch.empty(shape, dtype=dt, device=dev)
        cm, Em, cn, En = e(B, m, 2), e(B, m, 2, 2), e(B, n, 2), e(B, n, 2, 2)
        K_mm, K_nm, K_d = e(B, m, m), e(B, n, m), e(B, n)
        rc = getattr(_lib.lib(), "como_kernel_matrices_" + _lib.suffix(dt))(
            cov_params_img.contiguous().data_ptr(), int(size[0]), int(size[1]), coords_m.contiguous().data_ptr(), m,
            coords_n.contiguous().data_ptr(), n, float(mods[0].scale), cm.data_ptr(), Em.data_ptr(), cn.data_ptr(), En.data_ptr(),
            K_mm.data_ptr(), K_nm.data_ptr(), K_d.data_ptr(), B, _lib.stream_ptr(dev))
        _lib.check(rc, "como_kernel_matrices")
        return K_mm, K_nm, K_d
    cm = normalize_coordinates(coords_m, size)
    Em = interpolate_kernel_params(cov_params_img, cm)
    cn = normalize_coordinates(coords_n, size)
    En = interpolate_kernel_params(cov_params_img, cn)
    return (model.cov_modules[-1](cm, Em), model.cross_cov_modules[-1](cn, En, cm, Em), model.diagonal_cov_modules[-1](cn, En))


class MaskedResidual:
    """The residuals of a distillation over ALL rows together with the row mask of the valid ones -- what
    `distill_depth_from_scratch(..., masked_residual=True)` returns instead of the gathered valid residuals (whose number only the
    device knows: the gather costs a `nonzero` + a host synchronisation).  `std()` is torch.std of the valid entries (unbiased),
    from masked sums: equal to the gathered form up to the rounding of a different summation order."""

    def __init__(self, res, okm):
        self.res, self.okm = res, okm

    def std(self):
        r, o = self.res, self.okm
        if (KF_GLUE and r.is_cuda and r.dtype == torch.float64 and o.dtype == torch.bool and r.is_contiguous() and o.is_contiguous() and
                r.numel() == o.numel() and r.numel() > 0):
            # one launch, fixed summation order (csrc/kfglue.hip kg_masked_std_kernel) instead of thirteen
            from como_amd import _lib
            out = torch.empty((), dtype=torch.float64, device=r.device)
            _lib.check(_lib.lib().como_kf_masked_std_f64(r.data_ptr(), o.data_ptr(), r.numel(), out.data_ptr(), _lib.stream_ptr(r.device)),
                       "como_kf_masked_std_f64")
            return out
        zero = torch.zeros_like(self.res)
        n = self.okm.sum().to(self.res.dtype)
        mean = torch.where(self.okm, self.res, zero).sum() / n          # (selects: a masked row may hold anything)
        d = torch.where(self.okm, self.res - mean, zero)
        return torch.sqrt((d * d).sum() / (n - 1.0))

    def compact(self):
        return self.res.index_select(1, torch.nonzero(self.okm[0, :, 0])[:, 0])


def get_predictor(K_mm, K_nm, K_nn_diag, pad4=False, row_mask=None, want_sinv=True):
    """:30-48 -> Knm_Kmminv (B,n,m), L_mm, 1/stdev of the conditional variance (B,n,1).
    row_mask (1,n) bool or None: the rows that count (the others ride along with zero weights in the caller's normal equations):
    the smallest conditional variance -- the shift that keeps every variance positive -- is taken over THEM only.
    pad4: Knm_Kmminv is returned as the leading m columns of a (B,n,mp) buffer, mp = m rounded up to a multiple of 4, whose other
    columns are exactly zero (K_mm^-1 padded with zero columns before the product) -- 16-byte aligned rows for `gram_weighted`
    whatever the number of tracked points; `padded_predictor(Kt)` recovers the buffer.
    want_sinv=False: the third result is None (a caller with a fixed observation stdev does not read it)."""
    f = chol_small(K_mm, want_L=True, want_inv=True)          # csrc/smallsolve.hip: L_mm and K_mm^-1 in one launch
    L_mm = f["L"]
    m = K_mm.shape[-1]
    if (_GRAM_KERNEL and K_nm.is_cuda and K_nm.dtype == torch.float64 and K_nm.shape[0] == 1 and 0 < m <= 64 and K_nm.shape[1] > 0):
        # product, elementwise product and row reduction in ONE pass over K_nm (csrc/gram.hip `como_predictor_f64`) instead of a
        # library GEMM and two more passes over both n x m matrices
        from como_amd import _lib
        n = K_nm.shape[1]
        mp = (m + 3) // 4 * 4 if pad4 else m
        full = torch.empty((1, n, mp), dtype=K_nm.dtype, device=K_nm.device)
        var_n = torch.empty((1, n), dtype=K_nm.dtype, device=K_nm.device)
        Kc, ic, dc = K_nm.contiguous(), f["inv"].contiguous(), K_nn_diag.reshape(1, n).contiguous()
        _lib.check(_lib.lib().como_predictor_f64(Kc.data_ptr(), ic.data_ptr(), dc.data_ptr(), n, m, mp, full.data_ptr(), var_n.data_ptr(),
                                                 _lib.stream_ptr(K_nm.device)), "como_predictor_f64")
        Kt = full[:, :, :m] if mp != m else full
        if mp != m:
            Kt._como_padded = full
        if not want_sinv:
            return Kt, L_mm, None
        rmask = None if row_mask is None else row_mask.reshape(n)
        if _glue_ok(rmask) and (rmask is None or rmask.dtype == torch.bool):
            # min over the rows that count, the shift and 1 / sqrt in two launches (seven as torch ops), the same values
            sinv = torch.empty((1, n, 1), dtype=K_nm.dtype, device=K_nm.device)
            part = torch.empty(64, dtype=K_nm.dtype, device=K_nm.device)
            _lib.check(_lib.lib().como_kf_predictor_sinv_f64(var_n.data_ptr(), _lib.ptr(rmask), n, part.data_ptr(), sinv.data_ptr(),
                                                             _lib.stream_ptr(K_nm.device)), "como_kf_predictor_sinv_f64")
            return Kt, L_mm, sinv
        vmin = torch.min(var_n) if row_mask is None else torch.min(torch.where(row_mask.reshape(1, n), var_n, torch.full_like(var_n, float("inf"))))
        var_n = var_n + (vmin + 1e-8)
        return Kt, L_mm, 1.0 / torch.sqrt(var_n.unsqueeze(-1))
    if pad4 and m % 4:
        full = K_nm @ torch.nn.functional.pad(f["inv"], (0, 4 - m % 4))
        Kt = full[:, :, :m]
        Kt._como_padded = full
    else:
        Kt = K_nm @ f["inv"]
    var_n = K_nn_diag - torch.sum(K_nm * Kt, dim=2)
    vmin = torch.min(var_n) if row_mask is None else torch.min(torch.where(row_mask.reshape(var_n.shape), var_n, torch.full_like(var_n, float("inf"))))
    var_n = var_n + (vmin + 1e-8)
    return Kt, L_mm, 1.0 / torch.sqrt(var_n.unsqueeze(-1))


def distill_depth(Knm_Kmminv, z_obs, with_prior, L_mm=None, stdev_inv_obs=None):
    """:52-84: argmax p(d_n | d_m) p(d_m) in log-depth."""
    B, n, m = Knm_Kmminv.shape
    logz_obs = torch.log(z_obs)
    if not with_prior:
        logz_m = lstsq_chol(Knm_Kmminv, logz_obs)
    else:
        eye = torch.eye(m, device=Knm_Kmminv.device, dtype=Knm_Kmminv.dtype).reshape(1, m, m).repeat(B, 1, 1)
        A = torch.cat((trsm_lower(L_mm, eye), stdev_inv_obs * Knm_Kmminv), dim=1)
        b = torch.cat((torch.zeros((B, m, 1), device=A.device, dtype=A.dtype), stdev_inv_obs * logz_obs), dim=1)
        logz_m = lstsq_chol(A, b)
    return logz_m, Knm_Kmminv @ logz_m - logz_obs


def distill_depth_from_scratch(coords_m, coords_n, z_obs, cov_params_img, model, distill_with_prior, min_depth, stdev_obs=None,
                               obs_mask=None, masked_residual=False):
    """:88-118.  obs_mask (n,) bool or None: the observation rows that exist at all (a caller that did NOT gather the points that
    reproject into the image passes their mask: every row rides along, the masked ones with zero weight).  masked_residual: the
    second result is a `MaskedResidual` (all rows + the mask of the valid ones) instead of the gathered valid residuals."""
    assert coords_m.shape[0] == 1
    rm = None if obs_mask is None else obs_mask.reshape(1, -1)
    pre = KF_GLUE and z_obs.is_cuda and z_obs.dtype == torch.float64 and z_obs.dim() == 3 and z_obs.shape[2] == 1 and \
        z_obs.shape[1] > 0 and z_obs.stride(1) >= 1 and _glue_ok(obs_mask) and (obs_mask is None or obs_mask.dtype == torch.bool) and \
        _stdev_ok(stdev_obs)
    Kt, L_mm, sinv = get_predictor(*calc_kernel_matrices(coords_m, coords_n, cov_params_img, model), pad4=_GRAM_KERNEL, row_mask=rm,
                                   want_sinv=not (pre and stdev_obs is not None))
    glue = pre and _fast(Kt) and _glue_ok(sinv)
    if stdev_obs is not None and not glue:
        sinv = (1.0 / stdev_obs) * torch.ones((Kt.shape[0], Kt.shape[1], 1), dtype=Kt.dtype, device=Kt.device)
    if obs_mask is not None and not _fast(Kt):
        # (the gathering form below needs gathered inputs)
        sel = torch.nonzero(obs_mask)[:, 0]
        Kt, z_obs, sinv, obs_mask = Kt.index_select(1, sel), z_obs.index_select(1, sel), sinv.index_select(1, sel), None
    if _fast(Kt):
        # the same normal equations with the rows read in place: the validity test becomes a zero weight (no gather of the
        # valid rows, no [prior ; observations] concatenation, no library GEMM with a single output tile)
        m = Kt.shape[2]
        if glue:
            okm, _, y, wgt = distill_prep(z_obs, None if obs_mask is None else obs_mask.reshape(-1), min_depth,
                                          sinv if stdev_obs is None else None, stdev_obs, bool(distill_with_prior))
        else:
            okm = z_obs[:, :, 0:1] > min_depth
            if obs_mask is not None:
                okm = okm & obs_mask.reshape(1, -1, 1)
            y = torch.log(torch.where(okm, z_obs[:, :, 0:1], torch.ones_like(z_obs[:, :, 0:1])))
            # (a select, not a product: a masked row may hold anything -- a point behind the camera reprojects to non-finite coordinates)
            wgt = torch.where(okm, sinv * sinv, torch.zeros_like(sinv)) if distill_with_prior else okm.to(Kt.dtype)
        AtA, Atb = gram_weighted(padded_predictor(Kt), wgt, y)
        if AtA.shape[1] != m:                                 # (zero-padded columns: their rows / columns of the products are zero)
            AtA, Atb = AtA[:, :m, :m].contiguous(), Atb[:, :m].contiguous()
        if distill_with_prior:
            eye = torch.eye(m, device=Kt.device, dtype=Kt.dtype).reshape(1, m, m)
            Lm1 = trsm_lower(L_mm, eye)                       # the prior rows L_mm^-1 (distill_depth.py:60-63)
            AtA = AtA + Lm1.mT @ Lm1
        logz_m = chol_small(AtA, want_L=False, rhs=Atb)["X"]
        # residuals K~ logz_m - y: the predicted log-depths come from the depth-only pass of the dense-reference kernel (one
        # streaming pass, csrc/densify.hip) instead of a (n x m)(m x 1) library product
        from como_amd.odom.backend.dense_ref import depth_image
        Kp = padded_predictor(Kt)
        lz = logz_m.reshape(1, m)
        if Kp.shape[2] != m:
            lz = torch.nn.functional.pad(lz, (0, Kp.shape[2] - m))
        pred = torch.empty((1, Kp.shape[1]), dtype=Kp.dtype, device=Kp.device)
        depth_image(Kp, lz, logz_out=pred)
        res = MaskedResidual(pred.unsqueeze(-1) - y, okm)
        return logz_m, (res if masked_residual else res.compact())
    ok = torch.nonzero(z_obs[0, :, 0] > min_depth)[:, 0]
    logz_m, res = distill_depth(Kt.index_select(1, ok), z_obs.index_select(1, ok), distill_with_prior, L_mm=L_mm,
                                stdev_inv_obs=sinv.index_select(1, ok))
    return logz_m, (MaskedResidual(res, torch.ones_like(res, dtype=torch.bool)) if masked_residual else res)


def distill_conditional_depth_with_scale_prior(Knm_Kmminv, z_obs, z1, stdev_inv_obs):
    """:122-148: new inducing depths given the kept ones, pulled towards the median log-depth."""
    B, n, m = Knm_Kmminv.shape
    assert B == 1
    m1 = z1.shape[1]
    m2 = m - m1
    dev, dt = Knm_Kmminv.device, Knm_Kmminv.dtype
    s = torch.log(torch.median(z_obs))
    sp = 1.0 / 5e-2
    A = torch.cat((sp * torch.eye(m2, device=dev, dtype=dt).unsqueeze(0), stdev_inv_obs * Knm_Kmminv[:, :, m1:]), dim=1)
    b = torch.cat(((sp * s) * torch.ones((B, m2, 1), device=dev, dtype=dt),
                   stdev_inv_obs * (torch.log(z_obs) - Knm_Kmminv[:, :, :m1] @ torch.log(z1))), dim=1)
    return lstsq_chol(A, b)


def distill_conditional_depth_from_scratch(coords_m, z_m1, coords_n, cov_params_img, z_obs, model, min_depth, stdev_obs, obs_mask=None):
    """:152-175.  obs_mask: as in distill_depth_from_scratch."""
    assert coords_m.shape[0] == 1
    pre = KF_GLUE and z_obs.is_cuda and z_obs.dtype == torch.float64 and z_obs.dim() == 3 and z_obs.shape[2] == 1 and \
        z_obs.shape[1] > 0 and z_obs.stride(1) >= 1 and _glue_ok(obs_mask) and (obs_mask is None or obs_mask.dtype == torch.bool) and \
        _stdev_ok(stdev_obs)
    Kt, L_mm, sinv = get_predictor(*calc_kernel_matrices(coords_m, coords_n, cov_params_img, model), pad4=_GRAM_KERNEL, want_sinv=not pre)
    glue = pre and _fast(Kt)
    if not glue:
        sinv = (1.0 / stdev_obs) * torch.ones((Kt.shape[0], Kt.shape[1], 1), dtype=Kt.dtype, device=Kt.device)
    m, m1 = Kt.shape[2], z_m1.shape[1]
    if obs_mask is not None and not _fast(Kt):
        sel = torch.nonzero(obs_mask)[:, 0]
        Kt, z_obs, sinv, obs_mask = Kt.index_select(1, sel), z_obs.index_select(1, sel), sinv.index_select(1, sel), None
    if _fast(Kt):
        # [sp I ; sinv K~[:, m1:]] x = [sp s ; sinv (log z_obs - K~[:, :m1] log z_1)] as weighted normal equations of the rows in place:
        # the known columns enter through c = [log z_1 ; 0] (r = y - K~ c), the unknown block is the lower-right corner
        from como_amd.utils.select import masked_median
        if glue:
            okm, zs, logzs, wgt = distill_prep(z_obs, None if obs_mask is None else obs_mask.reshape(-1), min_depth, None,
                                               stdev_obs, True, want_zs=True)
        else:
            okm = z_obs[:, :, 0:1] > min_depth
            if obs_mask is not None:
                okm = okm & obs_mask.reshape(1, -1, 1)
            zs = torch.where(okm, z_obs[:, :, 0:1], torch.ones_like(z_obs[:, :, 0:1]))
            logzs, wgt = None, None
        s_med = torch.log(masked_median(zs[0, :, 0], okm[0, :, 0]))
        sp2 = (1.0 / 5e-2) ** 2
        Kp = padded_predictor(Kt)
        m2 = m - m1
        small = glue and m2 > 0 and z_m1.dtype == torch.float64 and z_m1.is_cuda and z_m1.is_contiguous() and s_med.dtype == torch.float64
        if small:
            from como_amd import _lib
            Lb, st = _lib.lib(), _lib.stream_ptr(Kt.device)
            c = torch.empty((1, Kp.shape[2], 1), device=Kt.device, dtype=Kt.dtype)
            _lib.check(Lb.como_kf_cond_c_f64(z_m1.data_ptr(), m1, Kp.shape[2], c.data_ptr(), st), "como_kf_cond_c_f64")
        else:
            c = torch.cat((torch.log(z_m1).reshape(1, m1, 1), torch.zeros((1, Kp.shape[2] - m1, 1), device=Kt.device, dtype=Kt.dtype)), dim=1)
        if not glue:
            logzs, wgt = torch.log(zs), torch.where(okm, sinv * sinv, torch.zeros_like(sinv))
        AtA, Atb = gram_weighted(Kp, wgt, logzs, c=c)
        if small and AtA.is_contiguous() and Atb.is_contiguous():
            # A22 = AtA[m1:, m1:] + sp2 I and b2 = Atb[m1:] + sp2 s in ONE launch (eye, two products, two sums, two copies as torch ops)
            A22 = torch.empty((1, m2, m2), device=Kt.device, dtype=Kt.dtype)
            b2 = torch.empty((1, m2, 1), device=Kt.device, dtype=Kt.dtype)
            _lib.check(Lb.como_kf_cond_system_f64(AtA.data_ptr(), Atb.data_ptr(), AtA.shape[2], m1, m2, sp2, s_med.data_ptr(),
                                                  A22.data_ptr(), b2.data_ptr(), st), "como_kf_cond_system_f64")
            return chol_small(A22, want_L=False, rhs=b2)["X"]
        A22 = AtA[:, m1:m, m1:m] + sp2 * torch.eye(m2, device=Kt.device, dtype=Kt.dtype)
        b2 = Atb[:, m1:m] + sp2 * s_med
        return chol_small(A22.contiguous(), want_L=False, rhs=b2.contiguous())["X"]
    ok = torch.nonzero(z_obs[0, :, 0] > min_depth)[:, 0]
    return distill_conditional_depth_with_scale_prior(Kt.index_select(1, ok), z_obs.index_select(1, ok), z_m1,
                                                      sinv.index_select(1, ok))
