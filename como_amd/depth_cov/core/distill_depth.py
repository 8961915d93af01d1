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


def lstsq_chol(A, b):
    """como/utils/lin_alg.py:82-87: normal equations + Cholesky."""
    AtA, Atb = _gram(A, b)
    return chol_small(AtA, want_L=False, rhs=Atb)["X"]        # factor + both substitutions in one launch


def calc_kernel_matrices(coords_m, coords_n, cov_params_img, model):
    """:8-27 -> K_mm (B,m,m), K_nm (B,n,m), K_nn_diag (B,n)."""
    size = cov_params_img.shape[-2:]
    cm = normalize_coordinates(coords_m, size)
    Em = interpolate_kernel_params(cov_params_img, cm)
    cn = normalize_coordinates(coords_n, size)
    En = interpolate_kernel_params(cov_params_img, cn)
    return (model.cov_modules[-1](cm, Em), model.cross_cov_modules[-1](cn, En, cm, Em), model.diagonal_cov_modules[-1](cn, En))


def get_predictor(K_mm, K_nm, K_nn_diag):
    """:30-48 -> Knm_Kmminv (B,n,m), L_mm, 1/stdev of the conditional variance (B,n,1)."""
    f = chol_small(K_mm, want_L=True, want_inv=True)          # csrc/smallsolve.hip: L_mm and K_mm^-1 in one launch
    L_mm = f["L"]
    Kt = K_nm @ f["inv"]
    var_n = K_nn_diag - torch.sum(K_nm * Kt, dim=2)
    var_n = var_n + (torch.min(var_n) + 1e-8)
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


def distill_depth_from_scratch(coords_m, coords_n, z_obs, cov_params_img, model, distill_with_prior, min_depth, stdev_obs=None):
    """:88-118"""
    assert coords_m.shape[0] == 1
    Kt, L_mm, sinv = get_predictor(*calc_kernel_matrices(coords_m, coords_n, cov_params_img, model))
    if stdev_obs is not None:
        sinv = (1.0 / stdev_obs) * torch.ones_like(sinv)
    ok = torch.nonzero(z_obs[0, :, 0] > min_depth)[:, 0]
    return distill_depth(Kt.index_select(1, ok), z_obs.index_select(1, ok), distill_with_prior, L_mm=L_mm,
                         stdev_inv_obs=sinv.index_select(1, ok))


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


def distill_conditional_depth_from_scratch(coords_m, z_m1, coords_n, cov_params_img, z_obs, model, min_depth, stdev_obs):
    """:152-175"""
    assert coords_m.shape[0] == 1
    Kt, L_mm, sinv = get_predictor(*calc_kernel_matrices(coords_m, coords_n, cov_params_img, model))
    sinv = (1.0 / stdev_obs) * torch.ones_like(sinv)
    ok = torch.nonzero(z_obs[0, :, 0] > min_depth)[:, 0]
    return distill_conditional_depth_with_scale_prior(Kt.index_select(1, ok), z_obs.index_select(1, ok), z_m1,
                                                      sinv.index_select(1, ok))
