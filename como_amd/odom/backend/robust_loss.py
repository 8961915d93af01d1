"""Robust weights (reference como/odom/backend/robust_loss.py:4-26).  The kernels inline the Huber
weight (csrc/common.cuh `huber`); these tensor versions serve host-side callers."""
import torch


def squared_error(r):
    return torch.ones_like(r)


def huber(r, k=1.345):
    a = r.abs()
    return torch.where(a < k, torch.ones_like(a), k / a)


def tukey(r, t=4.6851):
    a = r.abs()
    u = 1 - torch.square(a / t)
    return torch.where(a < t, u * u, torch.zeros_like(a))
