"""Oracle geometry primitives with the reference's floating-point operation order.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The validity masks must be bit-exact,
so every value that feeds a mask comparison is computed with the same sequence of
individually-rounded multiplies/adds the reference's torch-CPU run performs
(small torch.matmul = sequential k-loop without FMA; verified bitwise in
tests/test_oracle_vs_golden.py).
"""
import torch


def dot3_seq(a0, a1, a2, x, y, z):
    """((a0*x + a1*y) + a2*z) with one rounding per op (torch CPU bmm naive kernel order)."""
    return (a0 * x + a1 * y) + a2 * z


def invert_pose(T):
    """T (b,4,4) -> T^-1, reference lie_algebra.py:83-93 (Rt, -(Rt @ t))."""
    R = T[:, :3, :3]
    t = T[:, :3, 3]
    Ti = torch.zeros_like(T)
    Ti[:, :3, :3] = R.transpose(1, 2)
    for i in range(3):
        Ti[:, i, 3] = -dot3_seq(R[:, 0, i], R[:, 1, i], R[:, 2, i], t[:, 0], t[:, 1], t[:, 2])
    Ti[:, 3, 3] = 1.0
    return Ti


def rigid_apply(T, P):
    """P' = R P + t per batch; T (b,4,4), P (b,n,3); reference transforms.py:17-23."""
    out = []
    for i in range(3):
        acc = dot3_seq(T[:, None, i, 0], T[:, None, i, 1], T[:, None, i, 2], P[..., 0], P[..., 1], P[..., 2])
        out.append(acc + T[:, None, i, 3])
    return torch.stack(out, dim=-1)


def project(K, P):
    """Pinhole projection, reference camera.py:20-26: u = fx*X/Z + cx (multiply, divide, add)."""
    u = (K[0, 0] * P[..., 0]) / P[..., 2] + K[0, 2]
    v = (K[1, 1] * P[..., 1]) / P[..., 2] + K[1, 2]
    return u, v


def in_image(u, v, H, W):
    """1 <= u < W-1 and 1 <= v < H-1, reference photo.py:15-21 / photo_utils.py:12-18."""
    return (u >= 1) & (u < W - 1) & (v >= 1) & (v < H - 1)


def skew(p):
    z = torch.zeros_like(p[..., 0])
    return torch.stack((z, -p[..., 2], p[..., 1], p[..., 2], z, -p[..., 0], -p[..., 1], p[..., 0], z),
                       dim=-1).reshape(p.shape[:-1] + (3, 3))


def adjoint(T):
    """Ad(T) = [[R, 0], [[t]x R, R]], reference lie_algebra.py:60-67."""
    R = T[:, :3, :3]
    A = torch.zeros((T.shape[0], 6, 6), dtype=T.dtype)
    A[:, :3, :3] = R
    A[:, 3:, :3] = skew(T[:, :3, 3]) @ R
    A[:, 3:, 3:] = R
    return A


def se3_exp(xi):
    """Exp of xi=[omega, v] (COMO order), closed form (reference delegates to lietorch; unpinned)."""
    w, v = xi[..., :3], xi[..., 3:]
    th2 = (w * w).sum(-1)
    th = torch.sqrt(th2)
    small = th2 < 1e-12
    ths = torch.where(small, torch.ones_like(th), th)
    a = torch.where(small, 1 - th2 / 6, torch.sin(ths) / ths)[..., None, None]
    b = torch.where(small, 0.5 - th2 / 24, (1 - torch.cos(ths)) / ths**2)[..., None, None]
    c = torch.where(small, 1.0 / 6 - th2 / 120, (ths - torch.sin(ths)) / ths**3)[..., None, None]
    W = skew(w)
    W2 = W @ W
    eye = torch.eye(3, dtype=xi.dtype).expand_as(W)
    T = torch.zeros(xi.shape[:-1] + (4, 4), dtype=xi.dtype)
    T[..., :3, :3] = eye + a * W + b * W2
    T[..., :3, 3] = ((eye + b * W + c * W2) @ v[..., None])[..., 0]
    T[..., 3, 3] = 1
    return T


def grid_position(u, size, a):
    """Sample position grid_sample actually uses after the reference's normalise step:
    x_norm = (2a) u + a - 1 (coords.py:12-20, a = 1/size as the caller built it -- float32 in the
    two-frame path, two_frame_sfm.py:187-190) then (x_norm + 1) * (size / 2) - 0.5 (ATen CPU, align_corners=False)."""
    xn = (2 * a) * u + a - 1
    return (xn + 1) * (size / 2) - 0.5


def bilinear_zeros(img, u, v, ax=None, ay=None):
    """Bilinear sample of img (C,H,W) at pixel-centre coords (u=x, v=y), zero padding.

    Equivalent to grid_sample(bilinear, zeros, align_corners=False) after the
    reference's normalize_coordinates (coords.py:12-15): x_norm = (2x+1)/W - 1 maps back
    to x exactly in real arithmetic, i.e. the sample position IS (u, v).  Here the
    taps are taken directly at floor(u), floor(v) (values agree with torch's op to
    rounding; masks never depend on it).  reference photo.py:37-43, photo_utils.py:25-27.
    """
    C, H, W = img.shape
    if ax is not None:
        u = grid_position(u, W, ax)
        v = grid_position(v, H, ay)
    x0 = torch.floor(u)
    y0 = torch.floor(v)
    wx = u - x0
    wy = v - y0
    x0 = x0.long()
    y0 = y0.long()
    flat = img.reshape(C, -1)
    out = torch.zeros((C,) + u.shape, dtype=img.dtype)
    for dy, dx, wgt in ((0, 0, (1 - wy) * (1 - wx)), (0, 1, (1 - wy) * wx),
                        (1, 0, wy * (1 - wx)), (1, 1, wy * wx)):
        xx = x0 + dx
        yy = y0 + dy
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).reshape(-1)
        tap = flat[:, idx].reshape((C,) + u.shape)
        out = out + torch.where(ok, wgt, torch.zeros_like(wgt)) * tap
    return out


def huber_weight(x, k=1.345):
    """reference robust_loss.py:9-16."""
    ax = x.abs()
    return torch.where(ax < k, torch.ones_like(ax), k / ax)
