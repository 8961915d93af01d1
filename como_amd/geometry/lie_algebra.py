"""SE(3) helpers on small (B,4,4) pose tensors (device plumbing, O(B) work).

Mirrors the names of the reference's como/geometry/lie_algebra.py for the few
functions the hot path calls.  Tangent ordering is COMO's: xi = [omega(0:3), v(3:6)]
(rotation first) with a RIGHT perturbation T <- T * Exp(xi).

* se3_exp / batch_se3  -- reference lie_algebra.py:45-56 (delegates to the
  third-party lietorch `SE3.exp([v, omega]).matrix()`, un-pinned => parity
  "unpinned"; we restate the closed form: R = Rodrigues(omega), t = V(omega) v).
* invertSE3 / invertSE3_J / adjoint_matrix -- lie_algebra.py:60-95.
* skew_symmetric -- lie_algebra.py:147-157.
* SE3_logmap / SO3_logmap -- lie_algebra.py:127-176 (needed by the pose anchor
  prior; the reference's V^-1 expression is reproduced as written, including its
  `(0.5 * t) * (w_norm x t)` elementwise term).
"""
import os

import torch

_SE3_KERNEL = os.environ.get("COMO_SE3_KERNEL", "1") != "0"      # 0: invertSE3 as six torch launches (A/B)


def skew_symmetric(p):
    z = torch.zeros_like(p[..., 0])
    rows = (z, -p[..., 2], p[..., 1], p[..., 2], z, -p[..., 0], -p[..., 1], p[..., 0], z)
    return torch.stack(rows, dim=-1).reshape(p.shape[:-1] + (3, 3))


def _exp_coeffs(th2):
    th = torch.sqrt(th2)
    small = th2 < 1e-12
    ths = torch.where(small, torch.ones_like(th), th)
    a = torch.where(small, 1.0 - th2 / 6.0, torch.sin(ths) / ths)
    b = torch.where(small, 0.5 - th2 / 24.0, (1.0 - torch.cos(ths)) / (ths * ths))
    c = torch.where(small, 1.0 / 6.0 - th2 / 120.0, (ths - torch.sin(ths)) / (ths * ths * ths))
    return a, b, c


def se3_exp(delta):
    """delta (B,6) = [omega, v]  ->  (B,4,4)."""
    w, v = delta[..., :3], delta[..., 3:]
    th2 = (w * w).sum(-1, keepdim=True)
    a, b, c = _exp_coeffs(th2)
    W = skew_symmetric(w)
    W2 = W @ W
    eye = torch.eye(3, dtype=delta.dtype, device=delta.device).expand_as(W)
    R = eye + a[..., None] * W + b[..., None] * W2
    V = eye + b[..., None] * W + c[..., None] * W2
    T = torch.zeros(delta.shape[:-1] + (4, 4), dtype=delta.dtype, device=delta.device)
    T[..., :3, :3] = R
    T[..., :3, 3] = (V @ v[..., None])[..., 0]
    T[..., 3, 3] = 1.0
    return T


def batch_se3(poses, delta_T):
    return poses @ se3_exp(delta_T)


def invertSE3(T):
    if _SE3_KERNEL and T.is_cuda and T.dtype in (torch.float32, torch.float64) and T.numel() > 0:
        # one launch (csrc/window.hip se3_inverse_kernel) instead of six tiny torch ops: this runs several times per frame
        from como_amd import _lib
        Tc = T.contiguous()
        out = torch.empty_like(Tc)
        fn = getattr(_lib.lib(), "como_se3_inverse_" + _lib.suffix(T.dtype))
        _lib.check(fn(Tc.data_ptr(), out.data_ptr(), Tc.numel() // 16, _lib.stream_ptr(T.device)), "como_se3_inverse")
        return out
    Rt = T[..., :3, :3].transpose(-1, -2)
    Ti = torch.zeros_like(T)
    Ti[..., :3, :3] = Rt
    Ti[..., :3, 3:4] = -(Rt @ T[..., :3, 3:4])
    Ti[..., 3, 3] = 1.0
    return Ti


def composeSE3(A, B, mode):
    """op(A) op(B) for pose batches (n,4,4) / (1,4,4) broadcast -- mode 1: inv(A) B, mode 2: A inv(B), mode 0: A B -- as ONE launch
    (csrc/window.hip se3_compose_kernel) where the mirror's form is the inverse kernel + a 4x4 library product; the torch form for
    anything the kernel does not take (CPU tensors, other dtypes, higher-rank batches)."""
    if (_SE3_KERNEL and A.is_cuda and B.is_cuda and A.dtype == B.dtype and A.dtype in (torch.float32, torch.float64) and A.dim() == 3 and
            B.dim() == 3 and A.shape[0] > 0 and B.shape[0] > 0 and (A.shape[0] == B.shape[0] or 1 in (A.shape[0], B.shape[0]))):
        from como_amd import _lib
        n = max(A.shape[0], B.shape[0])
        Ac, Bc = A.contiguous(), B.contiguous()
        out = torch.empty((n, 4, 4), dtype=A.dtype, device=A.device)
        fn = getattr(_lib.lib(), "como_se3_compose_" + _lib.suffix(A.dtype))
        _lib.check(fn(Ac.data_ptr(), Bc.data_ptr(), out.data_ptr(), n, Ac.shape[0], Bc.shape[0], mode, _lib.stream_ptr(A.device)),
                   "como_se3_compose")
        return out
    if mode == 1:
        return invertSE3(A) @ B
    if mode == 2:
        return A @ invertSE3(B)
    return A @ B


def adjoint_matrix(T):
    R = T[:, :3, :3]
    Ad = torch.zeros((T.shape[0], 6, 6), dtype=T.dtype, device=T.device)
    Ad[:, :3, :3] = R
    Ad[:, 3:, :3] = skew_symmetric(T[:, :3, 3]) @ R
    Ad[:, 3:, 3:] = R
    return Ad


def invertSE3_J(T):
    return invertSE3(T), -adjoint_matrix(T)


def SO3_logmap(R, eps=1e-6):
    tr = R[..., 0, 0] + R[..., 1, 1] + R[..., 2, 2]
    tr3 = tr - 3.0
    theta = torch.acos(0.5 * (tr - 1))
    mag = torch.where(tr3 < -eps, theta / (2.0 * torch.sin(theta)),
                      0.5 - tr3 / 12.0 + tr3 * tr3 / 60.0)
    v = torch.stack((R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]), dim=1)
    return mag * v


def SE3_logmap(T, eps=1e-6):
    w = SO3_logmap(T[:, :3, :3])
    theta = torch.clamp(torch.linalg.norm(w, dim=1), min=eps)
    wn = w / theta
    tan = torch.tan(0.5 * theta)
    t = T[:, :3, 3]
    wxt = torch.linalg.cross(wn, t)
    Vinv_t = t - (0.5 * t) * wxt + (1.0 - theta / (2.0 * tan)) * torch.linalg.cross(wn, wxt)
    return torch.cat((w, Vinv_t), dim=-1)
