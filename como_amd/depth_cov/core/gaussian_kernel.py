"""Covariance-image helpers (reference como/depth_cov/core/gaussian_kernel.py): elementwise torch ops on the device."""
import math

import torch


def normalize_params_cov(kernel_img, det_eps=1e-8, corr_coeff_max=0.99):
    """Network output (B,3,H,W) -> (x, z, off-diagonal): gaussian_kernel.py:6-22."""
    x = torch.exp(torch.clamp(kernel_img[:, 0], min=math.log(1e-3), max=math.log(1e4)))
    z = torch.exp(torch.clamp(kernel_img[:, 1], min=math.log(1e-3), max=math.log(1e4)))
    rho = corr_coeff_max * torch.tanh(kernel_img[:, 2])
    return torch.stack((x, z, torch.sqrt(x * z - det_eps) * rho), dim=1).float()


def kernel_params_to_covariance(kernel_img_norm):
    """(B,3,H,W) -> (B,4,H,W) = [E00, E01, E10, E11]: gaussian_kernel.py:25-49."""
    x, z, o = kernel_img_norm[:, 0], kernel_img_norm[:, 1], kernel_img_norm[:, 2]
    return torch.stack((x, o, o, z), dim=1)


def interpolate_kernel_params(kernel_img, x):
    """Bilinear, border-padded lookup at normalised (row, col) coordinates -> (B,N,2,2): gaussian_kernel.py:52-79."""
    B, N = x.shape[:2]
    grid = x.flip(-1).unsqueeze(1)
    s = torch.nn.functional.grid_sample(kernel_img, grid, mode="bilinear", padding_mode="border", align_corners=False)
    return s.squeeze(2).permute(0, 2, 1).reshape(B, N, 2, 2)


def kernel_params_at(kernel_img, coords_pix, out_dtype=None):
    """normalize_coordinates(coords_pix, image size) -> cast to `out_dtype` (default: the image's) -> interpolate_kernel_params, as
    ONE launch on the GPU (csrc/densify.hip `como_cov_params_at`: the arithmetic of the torch chain, value for value) instead of
    eight.  coords_pix (B,N,2) row/col pixels -> (normalised coords (B,N,2), E (B,N,2,2)) in `out_dtype`."""
    from como_amd.utils.coords import normalize_coordinates
    dt = out_dtype or kernel_img.dtype
    B, N = coords_pix.shape[:2]
    size = kernel_img.shape[-2:]
    ok = (kernel_img.is_cuda and kernel_img.dtype == dt and dt in (torch.float32, torch.float64) and kernel_img.shape[1] == 4 and
          coords_pix.dtype in (torch.float32, torch.float64) and not (dt == torch.float64 and coords_pix.dtype == torch.float32))
    if not ok:
        cn = normalize_coordinates(coords_pix, size).to(dt)
        return cn, interpolate_kernel_params(kernel_img, cn)
    from como_amd import _lib
    cn = torch.empty((B, N, 2), dtype=dt, device=kernel_img.device)
    E = torch.empty((B, N, 2, 2), dtype=dt, device=kernel_img.device)
    if N:
        rc = _lib.lib().como_cov_params_at(kernel_img.contiguous().data_ptr(), int(size[0]), int(size[1]), coords_pix.contiguous().data_ptr(),
                                           int(coords_pix.dtype == torch.float64), int(dt == torch.float64), N, cn.data_ptr(), E.data_ptr(),
                                           B, _lib.stream_ptr(kernel_img.device))
        _lib.check(rc, "como_cov_params_at")
    return cn, E
