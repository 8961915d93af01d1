"""Scratch stand-in for the third-party `lietorch` package (absent from this image).

Only used by tests/golden/make_golden.py, in the build container, to let the
reference's como.geometry.lie_algebra import.  It restates the published closed
form of SE3.exp for a [tau, phi] tangent vector (translation first):
R = Rodrigues(phi), t = V(phi) tau.  lietorch itself is un-pinned by the
reference (install.sh:10-13) -> parity for SE3 exp is "unpinned".
"""
import torch


class _Mat:
    def __init__(self, T):
        self._T = T

    def matrix(self):
        return self._T


class SE3:
    @staticmethod
    def exp(xi):
        tau, phi = xi[..., :3], xi[..., 3:]
        th2 = (phi * phi).sum(-1, keepdim=True)
        th = torch.sqrt(th2)
        small = th2 < 1e-12
        ths = torch.where(small, torch.ones_like(th), th)
        A = torch.where(small, 1.0 - th2 / 6.0, torch.sin(ths) / ths)
        Bc = torch.where(small, 0.5 - th2 / 24.0, (1.0 - torch.cos(ths)) / (ths * ths))
        C = torch.where(small, 1.0 / 6.0 - th2 / 120.0, (ths - torch.sin(ths)) / (ths * ths * ths))
        z = torch.zeros_like(phi[..., 0])
        W = torch.stack(
            (z, -phi[..., 2], phi[..., 1], phi[..., 2], z, -phi[..., 0], -phi[..., 1], phi[..., 0], z),
            dim=-1,
        ).reshape(phi.shape[:-1] + (3, 3))
        W2 = W @ W
        I = torch.eye(3, dtype=xi.dtype, device=xi.device).expand_as(W)
        R = I + A[..., None] * W + Bc[..., None] * W2
        V = I + Bc[..., None] * W + C[..., None] * W2
        t = (V @ tau[..., None])[..., 0]
        T = torch.zeros(xi.shape[:-1] + (4, 4), dtype=xi.dtype, device=xi.device)
        T[..., :3, :3] = R
        T[..., :3, 3] = t
        T[..., 3, 3] = 1.0
        return _Mat(T)
