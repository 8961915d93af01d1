"""Drop-in for the reference's pybind module `como_backends` (como/backend/src/depth_cov_backends.cpp:3-6).

    import como_amd.como_backends as como_backends        # or como_amd.install_dropin()

Same two functions, argument meaning, in-place behaviour and error style (RuntimeError) as the
reference dispatcher (src/cov.cpp:5-65); GPU tensors only -- the work is done by the HIP kernels in
csrc/cov.hip through the C ABI (como_cross_covariance_*, como_chol_append_obs_info_f32).
"""
import ctypes

import torch

from como_amd import _lib


def cross_covariance(x1, E1, x2, E2, scale):
    """K12[b,i,j] = scale * C(E1_i, E2_j) * matern(Q(x1_i - x2_j, E1_i + E2_j)); returns a NEW (B,N,M) tensor with
    x1's options.  Inputs may be strided views (contiguity checks are commented out in the reference, cov.cpp:6-9)."""
    ts = (x1, E1, x2, E2)
    if not all(t.is_cuda for t in ts):
        if all(not t.is_cuda for t in ts):
            raise RuntimeError("como_amd.como_backends: CPU tensors are not supported (HIP build, no CPU fallback)")
        raise RuntimeError("All variables must be on same device.")
    dt = x1.dtype
    if any(t.dtype != dt for t in ts) or dt not in (torch.float16, torch.float32, torch.float64):
        raise RuntimeError("cross_covariance: float16 / float32 / float64 tensors of one dtype expected")
    B, N, M = x1.shape[0], x1.shape[1], x2.shape[1]
    K12 = torch.empty((B, N, M), dtype=dt, device=x1.device)
    if K12.numel() == 0:                                   # no points on one side: an empty matrix, as the torch-side callers expect
        return K12
    strides = (ctypes.c_long * 14)(*x1.stride(), *E1.stride(), *x2.stride(), *E2.stride())
    fn = getattr(_lib.lib(), "como_cross_covariance_" + _lib.suffix(dt))
    rc = fn(x1.data_ptr(), E1.data_ptr(), x2.data_ptr(), E2.data_ptr(), float(scale), K12.data_ptr(), B, N, M, strides,
            _lib.stream_ptr(x1.device))
    _lib.check(rc, "como_cross_covariance")
    return K12


def get_new_chol_obs_info(L, obs_info, var, k_ni, k_id, k_ii, N):
    """Append row N to the Cholesky factor L, write obs_info row N and downdate var, all IN PLACE
    (reference cov.cpp:34-65).  Contiguous float32 GPU tensors; returns None."""
    for name, t in (("L", L), ("obs_info", obs_info), ("var", var), ("k_ni", k_ni), ("k_id", k_id)):
        if not t.is_contiguous():
            raise RuntimeError(f"{name} must be contiguous")
    ts = (L, obs_info, k_ni, k_id)
    if not all(t.is_cuda for t in ts):
        if all(not t.is_cuda for t in ts):
            raise RuntimeError("como_amd.como_backends: CPU tensors are not supported (HIP build, no CPU fallback)")
        raise RuntimeError("All variables must be on same device.")
    if any(t.dtype != torch.float32 for t in (L, obs_info, var, k_ni, k_id)):
        raise RuntimeError("get_new_chol_obs_info: float32 only (as the reference kernels)")
    B, n, d = obs_info.shape
    rc = _lib.lib().como_chol_append_obs_info_f32(L.data_ptr(), obs_info.data_ptr(), var.data_ptr(), k_ni.data_ptr(),
                                                  k_id.data_ptr(), float(k_ii), B, n, d, int(N),
                                                  _lib.stream_ptr(L.device))
    _lib.check(rc, "como_chol_append_obs_info")
    return None
