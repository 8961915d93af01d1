"""Image gradients and pyramids on the HIP kernels of csrc/image.hip.

Mirror of como/utils/image_processing.py (ImageGradientModule :8-44, GaussianBlurModule :47-65, ImagePyramidModule :68-87,
IntrinsicsPyramidModule :109-123, DepthPyramidModule :90-106 with pyr_depth of como/data/depth_resize.py): same class names, constructor arguments and call results.  GPU tensors only.
"""
import torch

from como_amd import _lib
from como_amd.geometry.camera import resize_intrinsics


def img_and_grads(img):
    """(N,C,H,W) -> (N,3C,H,W) = cat(img, gx, gy): the tensor Mapping.get_img_and_grads (Mapping.py:369-379) assembles."""
    _lib.require_cuda(img)
    img = img.contiguous()
    N, C, H, W = img.shape
    out = torch.empty((N, 3 * C, H, W), dtype=img.dtype, device=img.device)
    fn = getattr(_lib.lib(), "como_img_grads_" + _lib.suffix(img.dtype))
    _lib.check(fn(img.data_ptr(), out.data_ptr(), N, C, H, W, _lib.stream_ptr(img.device)), "como_img_grads")
    return out


def blur_down(img):
    """GaussianBlurModule followed by [:, :, 0::2, 0::2] (one pyramid step, image_processing.py:85)."""
    _lib.require_cuda(img)
    img = img.contiguous()
    N, C, H, W = img.shape
    out = torch.empty((N, C, (H + 1) // 2, (W + 1) // 2), dtype=img.dtype, device=img.device)
    fn = getattr(_lib.lib(), "como_img_blur_down_" + _lib.suffix(img.dtype))
    _lib.check(fn(img.data_ptr(), out.data_ptr(), N * C, H, W, _lib.stream_ptr(img.device)), "como_img_blur_down")
    return out


def blur(img):
    """GaussianBlurModule.forward (image_processing.py:60-65)."""
    _lib.require_cuda(img)
    img = img.contiguous()
    N, C, H, W = img.shape
    out = torch.empty_like(img)
    fn = getattr(_lib.lib(), "como_img_blur_" + _lib.suffix(img.dtype))
    _lib.check(fn(img.data_ptr(), out.data_ptr(), N * C, H, W, _lib.stream_ptr(img.device)), "como_img_blur")
    return out


_DEPTH_MODES = {"bilinear": 0, "nearest_neighbor": 1, "max": 2, "min": 3, "masked_bilinear": 4}


def pyr_depth(depth, mode, kernel_size=2):
    """como/data/depth_resize.py:6-36 (factors of 2 only, as the reference notes)."""
    if kernel_size != 2:
        raise ValueError("pyr_depth: kernel_size 2 only")
    if mode not in _DEPTH_MODES:
        raise ValueError("pyr_depth mode: " + mode + " is not implemented.")
    _lib.require_cuda(depth)
    depth = depth.contiguous()
    N, C, H, W = depth.shape
    md = _DEPTH_MODES[mode]
    Ho, Wo = ((H + 1) // 2, (W + 1) // 2) if md == 1 else (H // 2, W // 2)
    out = torch.empty((N, C, Ho, Wo), dtype=depth.dtype, device=depth.device)
    fn = getattr(_lib.lib(), "como_depth_pool2_" + _lib.suffix(depth.dtype))
    _lib.check(fn(depth.data_ptr(), out.data_ptr(), N * C, H, W, md, _lib.stream_ptr(depth.device)), "como_depth_pool2")
    return out


class GaussianBlurModule:
    def __init__(self, channels, device, dtype):
        self.channels = channels

    def __call__(self, x):
        return blur(x)

    forward = __call__


class DepthPyramidModule:
    def __init__(self, start_level, end_level, mode, device):
        self.start_level, self.end_level, self.mode = start_level, end_level, mode

    def __call__(self, x):
        pyr = []
        lvl = x
        for i in range(self.end_level - 1):
            if i >= self.start_level:
                pyr.insert(0, lvl)
            lvl = pyr_depth(lvl, self.mode, kernel_size=2)
        pyr.insert(0, lvl)
        return pyr

    forward = __call__


class ImageGradientModule:
    def __init__(self, channels, device, dtype):
        self.channels = channels

    def __call__(self, x):
        C = x.shape[1]
        s = img_and_grads(x)
        return s[:, C:2 * C], s[:, 2 * C:]

    forward = __call__


class ImagePyramidModule:
    def __init__(self, channels, start_level, end_level, device, dtype):
        self.start_level, self.end_level = start_level, end_level

    def __call__(self, x):
        pyr = []
        lvl = x
        for i in range(self.end_level - 1):
            if i >= self.start_level:
                pyr.insert(0, lvl)
            lvl = blur_down(lvl)
        pyr.insert(0, lvl)
        return pyr

    forward = __call__


class IntrinsicsPyramidModule:
    def __init__(self, start_level, end_level, device):
        self.start_level, self.end_level = start_level, end_level

    def __call__(self, K_orig, image_scale_start):
        pyr = []
        for i in range(self.start_level, self.end_level):
            pyr.insert(0, resize_intrinsics(K_orig, [image_scale_start[0] * pow(2.0, -i), image_scale_start[1] * pow(2.0, -i)]))
        return pyr

    forward = __call__


def rgb_to_grayscale(rgb):
    """ITU-R 601-2 luma of a (..,3,H,W) image, the weights torchvision's functional.rgb_to_grayscale applies (the reference
    calls it in Mapping.get_img_and_grads / Tracking.prep_tracking_img)."""
    if rgb.is_cuda and rgb.dtype in (torch.float32, torch.float64) and rgb.dim() == 4 and rgb.shape[1] == 3 and rgb.numel() > 0:
        # one launch (csrc/image.hip rgb_to_gray_kernel: the same five roundings) instead of five: this runs on every frame
        from como_amd import _lib
        x = rgb.contiguous()
        n, _, h, w = x.shape
        out = torch.empty((n, 1, h, w), dtype=x.dtype, device=x.device)
        _lib.check(getattr(_lib.lib(), "como_rgb_to_gray_" + _lib.suffix(x.dtype))(x.data_ptr(), out.data_ptr(), n, h, w, _lib.stream_ptr(x.device)),
                   "como_rgb_to_gray")
        return out
    r, g, b = rgb.unbind(dim=-3)
    return (0.2989 * r + 0.587 * g + 0.114 * b).unsqueeze(-3)
