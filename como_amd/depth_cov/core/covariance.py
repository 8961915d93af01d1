"""DepthCov covariance modules and the GP predictor on the GPU.

Mirrors the reference's como/depth_cov/core/covariance.py (CovarianceModule / CrossCovarianceModule forward, Python-twin
kernel formula of kernels.py:22-88) and Mapping.prep_predictor (Mapping.py:430-468); the kernel-matrix assembly and the
fused K~ = K_nm K_mm^-1 run in csrc/densify.hip.  `scale` = scale_prior * exp(scale_param) (covariance.py:19-20).
"""
import torch

from como_amd import _lib
from como_amd.utils.lin_alg import chol_small
from como_amd.depth_cov.core.gaussian_kernel import interpolate_kernel_params
from como_amd.utils.coords import normalize_coordinates


def kernel_matrix(x1, E1, x2, E2, scale):
    """(B,N,2),(B,N,2,2),(B,M,2),(B,M,2,2) -> (B,N,M); CrossCovarianceModule.forward (covariance.py:33-39)."""
    _lib.require_cuda(x1, E1, x2, E2)
    dt = E1.dtype
    B, N, M = x1.shape[0], x1.shape[1], x2.shape[1]
    out = torch.empty((B, N, M), dtype=dt, device=x1.device)
    if out.numel() == 0:                                   # e.g. no sparse point of the previous keyframe reprojects into the new one
        return out
    fn = getattr(_lib.lib(), "como_kernel_matrix_" + _lib.suffix(dt))
    rc = fn(x1.to(dt).contiguous().data_ptr(), E1.contiguous().data_ptr(), x2.to(dt).contiguous().data_ptr(),
            E2.contiguous().data_ptr(), float(scale), out.data_ptr(), B, N, M, _lib.stream_ptr(x1.device))
    _lib.check(rc, "como_kernel_matrix")
    return out


def covariance(coords, E, scale):
    """CovarianceModule.forward (covariance.py:22-26)."""
    return kernel_matrix(coords, E, coords, E, scale)


class CovarianceModule:
    """covariance.py:10-26 with the scale resolved to a float: `module(coords, E)` -> (B,N,N)."""

    def __init__(self, scale):
        self.scale = float(scale)

    def get_scale(self):
        return self.scale

    def __call__(self, coords, E):
        return covariance(coords, E, self.scale)

    forward = __call__


class CrossCovarianceModule(CovarianceModule):
    """covariance.py:28-39: `module(coords_train, E_train, coords_test, E_test)` -> (B,N,M)."""

    def __call__(self, coords_train, E_train, coords_test, E_test):
        return kernel_matrix(coords_train, E_train, coords_test, E_test, self.scale)

    forward = __call__


class DiagonalCovarianceModule(CovarianceModule):
    """covariance.py:42-50 + kernels.py:69-88 (Q = 0): K_ii = scale * 2 sqrt(det E) / safe_sqrt(det 2E) * matern(0); O(n)
    elementwise torch ops."""

    def __call__(self, coords, E):
        if E.is_cuda and E.dtype in (torch.float32, torch.float64) and E.shape[-2:] == (2, 2):
            out = torch.empty(E.shape[:-2], dtype=E.dtype, device=E.device)
            rc = getattr(_lib.lib(), "como_diag_cov_" + _lib.suffix(E.dtype))(E.contiguous().data_ptr(), out.numel(), self.scale,
                                                                             out.data_ptr(), _lib.stream_ptr(E.device))
            _lib.check(rc, "como_diag_cov")
            return out
        det = E[..., 0, 0] * E[..., 1, 1] - E[..., 0, 1] * E[..., 1, 0]
        E2 = 2 * E
        det2 = E2[..., 0, 0] * E2[..., 1, 1] - E2[..., 0, 1] * E2[..., 1, 0]
        C = 2.0 * torch.sqrt(det) / torch.sqrt(det2 + 1e-8)
        q = torch.sqrt(torch.zeros_like(C) + 1e-8)                          # safe_sqrt(0)
        tmp = (3.0 ** 0.5) * q
        return C * ((1 + tmp) * torch.exp(-tmp)) * self.scale

    forward = __call__


_jitter_cache = {}


def _jitter(B, m, dev, dt):
    """diag_embed(1e-6 (float32) as dt) for (B, m): four tiny launches per keyframe for a constant -- built once per shape."""
    key = (B, m, str(dev), dt)
    J = _jitter_cache.get(key)
    if J is None:
        if len(_jitter_cache) > 64:
            _jitter_cache.clear()
        J = _jitter_cache[key] = torch.diag_embed((1e-6 * torch.ones(B, m, device=dev)).to(dt))
    return J


def prep_predictor(cov_params_img, coords_m, scale, photo_img_size=None, K_mm_inv=None, out=None, out_pix=None):
    """Mapping.prep_predictor (Mapping.py:430-468): returns (K_mm_inv (B,m,m), L_mm (B,m,m), Knm_Kmminv (B,H,W,m)).
    K_nm (H*W x m per keyframe) is never materialised.  K_mm_inv: optional precomputed inverse (then L_mm is returned as
    None and only K~ = K_nm K_mm^-1 is formed) -- K_mm is ill-conditioned (~1e8), so two float64 LAPACKs agree on its
    inverse to ~1e-8 only; parity tests of the kernels DOWNSTREAM of K~ pass the reference's own inverse."""
    _lib.require_cuda(cov_params_img, coords_m)
    B, _, Hc, Wc = cov_params_img.shape
    dt, dev = cov_params_img.dtype, cov_params_img.device
    Hp, Wp = photo_img_size or (Hc, Wc)
    m = coords_m.shape[1]
    from como_amd.depth_cov.core.gaussian_kernel import kernel_params_at
    cm, Em = kernel_params_at(cov_params_img, coords_m.to(dt))         # normalised coordinates + interpolated parameters, one launch
    if K_mm_inv is None:
        K_mm = covariance(cm, Em, scale)
        K_mm = K_mm + _jitter(B, m, dev, dt)                   # float32 jitter as Mapping.py:450 (the constant matrix is cached)
        # conditioning on the device in ONE launch (csrc/smallsolve.hip): L_mm and K_mm^-1 = L^-T L^-1
        f = chol_small(K_mm, want_L=True, want_inv=True)
        L_mm, K_mm_inv = f["L"], f["inv"].contiguous()
    else:
        L_mm = None
        K_mm_inv = K_mm_inv.to(device=dev, dtype=dt).contiguous()
    # out / out_pix: caller-owned destinations -- (B,Hp,Wp,m) in the image's dtype (e.g. the new keyframe's slot of the window's
    # predictor buffer) and, float64 only, a float32 tensor of the same shape that receives the rounded values (the per-pixel kernels'
    # mirror): the kernel writes both, no 157 MB copy + conversion pass afterwards
    if out is None:
        out = torch.empty((B, Hp, Wp, m), dtype=dt, device=dev)
    elif tuple(out.shape) != (B, Hp, Wp, m) or out.dtype != dt or not out.is_contiguous():
        raise RuntimeError("como_amd prep_predictor: `out` must be a contiguous (B,Hp,Wp,m) tensor of the covariance image's dtype")
    if out_pix is not None:
        if dt != torch.float64 or out_pix.dtype != torch.float32 or tuple(out_pix.shape) != (B, Hp, Wp, m) or not out_pix.is_contiguous():
            raise RuntimeError("como_amd prep_predictor: `out_pix` needs a float64 covariance image and a contiguous float32 (B,Hp,Wp,m) tensor")
        rc = _lib.lib().como_ktilde_mirror_f64(cov_params_img.contiguous().data_ptr(), Hc, Wc, cm.contiguous().data_ptr(),
                                               Em.contiguous().data_ptr(), K_mm_inv.data_ptr(), float(scale), B, Hp, Wp, m, out.data_ptr(),
                                               out_pix.data_ptr(), _lib.stream_ptr(dev))
    else:
        fn = getattr(_lib.lib(), "como_ktilde_" + _lib.suffix(dt))
        rc = fn(cov_params_img.contiguous().data_ptr(), Hc, Wc, cm.contiguous().data_ptr(), Em.contiguous().data_ptr(),
                K_mm_inv.data_ptr(), float(scale), B, Hp, Wp, m, out.data_ptr(), _lib.stream_ptr(dev))
    _lib.check(rc, "como_ktilde")
    return K_mm_inv, L_mm, out
