"""Affine brightness composition (reference como/geometry/affine_brightness.py:5-16).  aff = (a, b): I' = exp(a) I + b,
stored (B,2,1)."""
import torch


def get_aff_w_curr(aff_w_ref, aff_curr_ref):
    out = aff_w_ref.clone()
    out[:, 0, :] += aff_curr_ref[:, 0, :]
    out[:, 1, :] += aff_curr_ref[:, 1, :] * torch.exp(aff_curr_ref[:, 0, :])
    return out


def get_rel_aff(aff1, aff2):
    rel = torch.empty_like(aff1)
    rel[:, 0, :] = aff1[:, 0, :] - aff2[:, 0, :]
    rel[:, 1, :] = torch.exp(-rel[:, 0, :]) * (aff1[:, 1, :] - aff2[:, 1, :])
    return rel
