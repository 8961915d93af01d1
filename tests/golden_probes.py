"""Probe vectors of the H v checksums stored in the full-size fixtures (shared by tests/golden/make_golden_r2.py, which
writes S v for the reference's H, and the tests, which recompute S v for theirs; S = D^-1/2 H D^-1/2)."""
import torch


def probes(D, seed=123):
    g = torch.Generator().manual_seed(seed)
    return torch.stack((torch.ones(D, dtype=torch.float64), torch.randn(D, generator=g, dtype=torch.float64),
                        torch.randn(D, generator=g, dtype=torch.float64)), dim=1)
