"""Scalar anchors (reference como/odom/factors/scalar_prior_factors.py:4-34)."""
import torch


def linearize_scalar_prior(scalar, meas, H, g, Dscalar, sigma):
    info = (1.0 / sigma) * (1.0 / sigma)
    r = (scalar - meas)[:, 0]
    g[Dscalar[0]:Dscalar[1]] += (info * (-r)).to(g.dtype)
    H[Dscalar[0]:Dscalar[1], Dscalar[0]:Dscalar[1]] += info
    return info * torch.square(r)


def linearize_multi_scalar_prior(scalar, meas, H, g, inds, sigma):
    info = (1.0 / sigma) * (1.0 / sigma)
    r = scalar - meas
    g[inds] += (info * (-r)).to(g.dtype)
    H[inds, inds] += info
    return info * torch.sum(torch.square(r))
