"""Pinhole camera helpers (reference como/geometry/camera.py), small torch ops with the reference's
operation order (u = fx*X/Z + cx: multiply, divide, add -- it feeds validity masks)."""
import torch


def resize_intrinsics(K, image_scale_factors):
    """camera.py:4-15 (adds the scale to the principal point, as the reference does)."""
    T = torch.tensor([[image_scale_factors[1], 0, image_scale_factors[1]], [0, image_scale_factors[0], image_scale_factors[0]],
                      [0, 0, 1]], device=K.device, dtype=K.dtype)
    return T @ K


def projection(K, P):
    """camera.py:20-37: p (…,2) and dp/dP (…,2,3)."""
    t1 = K[0, 0] * P[..., 0] / P[..., 2]
    t2 = K[1, 1] * P[..., 1] / P[..., 2]
    p = torch.stack((t1 + K[0, 2], t2 + K[1, 2]), dim=-1)
    zero = torch.zeros_like(t1)
    dp = torch.stack((K[0, 0] + zero, zero, -t1, zero, K[1, 1] + zero, -t2), dim=-1).reshape(P.shape[:-1] + (2, 3))
    return p, dp / P[..., 2, None, None]


def backprojection(K, p, z):
    """camera.py:43-54: P = z * ray, dP/dz = ray (…,3,1)."""
    ray = torch.stack(((p[..., 0] - K[0, 2]) / K[0, 0], (p[..., 1] - K[1, 2]) / K[1, 1], torch.ones_like(p[..., 0])), dim=-1)
    return z * ray, ray[..., None]


def backprojection_points(K, p, z):
    """The values of `backprojection` alone -- P = z * ray (…,3) -- as one launch on the GPU (csrc/densify.hip `como_backproject_*`,
    same operations in the same order) instead of seven; K (3,3), p (…,2) x/y, z (…,1)."""
    if not (p.is_cuda and p.dtype in (torch.float32, torch.float64) and z.dtype == p.dtype and K.dtype == p.dtype and
            z.shape[:-1] == p.shape[:-1] and z.shape[-1] == 1):
        return backprojection(K, p, z)[0]
    from como_amd import _lib
    n = p.numel() // 2
    P = torch.empty(p.shape[:-1] + (3,), dtype=p.dtype, device=p.device)
    if n:
        rc = getattr(_lib.lib(), "como_backproject_" + _lib.suffix(p.dtype))(K.contiguous().data_ptr(), p.contiguous().data_ptr(),
                                                                             z.contiguous().data_ptr(), n, P.data_ptr(), _lib.stream_ptr(p.device))
        _lib.check(rc, "como_backproject")
    return P


def transform_project(K, Tji, Pi):
    """camera.py:57-68 with sequential accumulation (K T[:3,:]) P (see csrc/track.hip for the device version)."""
    def dot3(a0, a1, a2, x, y, z):
        return (a0 * x + a1 * y) + a2 * z
    Pm = torch.stack([torch.stack([dot3(K[i, 0], K[i, 1], K[i, 2], Tji[:, 0, j], Tji[:, 1, j], Tji[:, 2, j]) for j in range(4)], -1)
                      for i in range(3)], -2)                     # (b,3,4)
    ph = torch.stack([dot3(Pm[:, None, i, 0], Pm[:, None, i, 1], Pm[:, None, i, 2], Pi[..., 0], Pi[..., 1], Pi[..., 2])
                      + Pm[:, None, i, 3] for i in range(3)], -1)
    depth = ph[..., 2:3]
    return ph[..., :2] / depth, depth
