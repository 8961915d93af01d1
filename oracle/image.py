"""Oracle: image gradients, Gaussian pyramid, intrinsics pyramid
(reference como/utils/image_processing.py, como/geometry/camera.py:4-15).
TEST INFRASTRUCTURE (see oracle/__init__.py).  Written as explicit shifted-slice stencils.
"""
import torch


def _reflect_pad(x):
    return torch.nn.functional.pad(x, (1, 1, 1, 1), mode="reflect")


def scharr(img):
    """(B,C,H,W) -> gx, gy; Scharr/32 with reflect padding, image_processing.py:8-44."""
    p = _reflect_pad(img)
    tl, tc, tr = p[..., :-2, :-2], p[..., :-2, 1:-1], p[..., :-2, 2:]
    ml, mr = p[..., 1:-1, :-2], p[..., 1:-1, 2:]
    bl, bc, br = p[..., 2:, :-2], p[..., 2:, 1:-1], p[..., 2:, 2:]
    gx = (3.0 * (tr - tl) + 10.0 * (mr - ml) + 3.0 * (br - bl)) / 32.0
    gy = (3.0 * (bl - tl) + 10.0 * (bc - tc) + 3.0 * (br - tr)) / 32.0
    return gx, gy


def blur_down(img):
    """[1 2 1]^2/16 blur (reflect) then 2x decimation, image_processing.py:47-87."""
    p = _reflect_pad(img)
    acc = torch.zeros_like(img)
    wts = ((1.0, 2.0, 1.0), (2.0, 4.0, 2.0), (1.0, 2.0, 1.0))
    H, W = img.shape[-2:]
    for dy in range(3):
        for dx in range(3):
            acc = acc + wts[dy][dx] * p[..., dy:dy + H, dx:dx + W]
    return (acc / 16.0)[..., 0::2, 0::2]


def image_pyramid(img, start_level, end_level):
    """Coarse-to-fine list, image_processing.py:68-87."""
    pyr = []
    x = img
    for i in range(end_level - 1):
        if i >= start_level:
            pyr.insert(0, x)
        x = blur_down(x)
    pyr.insert(0, x)
    return pyr


def intrinsics_pyramid(K, start_level, end_level, scale0=(1.0, 1.0)):
    """K' = [[sx,0,sx],[0,sy,sy],[0,0,1]] K  (adds s to the principal point -- as the reference does),
    camera.py:4-15, image_processing.py:109-123."""
    pyr = []
    for i in range(start_level, end_level):
        sy, sx = scale0[0] * 2.0 ** (-i), scale0[1] * 2.0 ** (-i)
        T = torch.tensor([[sx, 0, sx], [0, sy, sy], [0, 0, 1]], dtype=K.dtype)
        pyr.insert(0, T @ K)
    return pyr
