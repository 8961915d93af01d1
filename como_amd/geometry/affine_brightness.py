"""Affine brightness composition (reference como/geometry/affine_brightness.py:5-16).  aff = (a, b): I' = exp(a) I + b,
stored (B,2,1).  On the GPU, for two operands of one shape and type, each function is ONE launch (csrc/kfglue.hip kg_aff_kernel: the
reference's operations, one rounding each) instead of seven on two-element tensors."""
import os

import torch

_KF_GLUE = os.environ.get("COMO_KF_GLUE", "1") != "0"


def _fused(p, q, mode):
    if not (_KF_GLUE and p.is_cuda and q.is_cuda and p.dtype == q.dtype and p.dtype in (torch.float32, torch.float64) and
            p.shape == q.shape and p.dim() == 3 and p.shape[1:] == (2, 1) and p.shape[0] > 0 and p.is_contiguous() and q.is_contiguous()
            and not (p.requires_grad or q.requires_grad)):
        return None
    from como_amd import _lib
    out = torch.empty_like(p)
    fn = getattr(_lib.lib(), "como_kf_aff_" + _lib.suffix(p.dtype))
    _lib.check(fn(p.data_ptr(), q.data_ptr(), p.shape[0], mode, out.data_ptr(), _lib.stream_ptr(p.device)), "como_kf_aff")
    return out


def get_aff_w_curr(aff_w_ref, aff_curr_ref):
    out = _fused(aff_w_ref, aff_curr_ref, 0)
    if out is not None:
        return out
    out = aff_w_ref.clone()
    out[:, 0, :] += aff_curr_ref[:, 0, :]
    out[:, 1, :] += aff_curr_ref[:, 1, :] * torch.exp(aff_curr_ref[:, 0, :])
    return out


def get_rel_aff(aff1, aff2):
    rel = _fused(aff1, aff2, 1)
    if rel is not None:
        return rel
    rel = torch.empty_like(aff1)
    rel[:, 0, :] = aff1[:, 0, :] - aff2[:, 0, :]
    rel[:, 1, :] = torch.exp(-rel[:, 0, :]) * (aff1[:, 1, :] - aff2[:, 1, :])
    return rel
