"""Coordinate helpers (reference como/utils/coords.py)."""
import torch


def swap_coords_xy(coords):
    return coords.flip(-1)


_A_cache = {}
_KF_GLUE = __import__("os").environ.get("COMO_KF_GLUE", "1") != "0"


def _inv_dims(dims, device, dtype):
    """(A, 2 A) with A = 1 / dims on the device: built once per (dims, device, dtype) -- the keyframe path normalises coordinates
    a dozen times per keyframe, each time with a host list -> device copy and two tiny launches for the same two numbers."""
    key = (tuple(float(d) for d in dims), str(device), dtype)
    v = _A_cache.get(key)
    if v is None:
        A = 1.0 / torch.as_tensor([float(d) for d in dims], device=device, dtype=dtype)
        v = _A_cache[key] = (A, 2 * A)
    return v


def normalize_coordinates(x_pixel, dims, swap=False):
    """Pixel -> [-1,1] with pixel centres at fractional positions: x_norm = 2 A x + A - 1, A = 1/dims (coords.py:12-15).
    swap: swap_coords_xy of the result (the grid a bilinear look-up wants), in the same launch."""
    if swap:
        if (not torch.is_tensor(dims) and _KF_GLUE and x_pixel.is_cuda and x_pixel.dtype in (torch.float32, torch.float64) and
                x_pixel.dim() >= 1 and x_pixel.shape[-1] == 2 and x_pixel.is_contiguous() and x_pixel.numel() > 0 and
                not x_pixel.requires_grad):
            from como_amd import _lib
            A, A2 = _inv_dims(dims, x_pixel.device, x_pixel.dtype)
            out = torch.empty_like(x_pixel)
            fn = getattr(_lib.lib(), "como_kf_normalize_coords_swap_" + _lib.suffix(x_pixel.dtype))
            _lib.check(fn(x_pixel.data_ptr(), x_pixel.numel(), A.data_ptr(), A2.data_ptr(), out.data_ptr(),
                          _lib.stream_ptr(x_pixel.device)), "como_kf_normalize_coords_swap")
            return out
        return swap_coords_xy(normalize_coordinates(x_pixel, dims))
    if torch.is_tensor(dims) or not x_pixel.is_floating_point():
        A = 1.0 / torch.as_tensor(dims, device=x_pixel.device, dtype=x_pixel.dtype)
        return 2 * A * x_pixel + A - 1
    A, A2 = _inv_dims(dims, x_pixel.device, x_pixel.dtype)
    if (_KF_GLUE and x_pixel.is_cuda and x_pixel.dtype in (torch.float32, torch.float64) and x_pixel.dim() >= 1 and
            x_pixel.shape[-1] == 2 and x_pixel.is_contiguous() and x_pixel.numel() > 0 and not x_pixel.requires_grad):
        # one launch instead of three (csrc/kfglue.hip kg_normalize_coords_kernel: the same three roundings per element)
        from como_amd import _lib
        out = torch.empty_like(x_pixel)
        fn = getattr(_lib.lib(), "como_kf_normalize_coords_" + _lib.suffix(x_pixel.dtype))
        _lib.check(fn(x_pixel.data_ptr(), x_pixel.numel(), A.data_ptr(), A2.data_ptr(), out.data_ptr(), _lib.stream_ptr(x_pixel.device)),
                   "como_kf_normalize_coords")
        return out
    return A2 * x_pixel + A - 1


def normalize_coordinates_A(x_pixel, A):
    return 2 * A * x_pixel + A - 1


def unnormalize_coordinates(x_norm, dims):
    A = torch.as_tensor(dims, device=x_norm.device, dtype=x_norm.dtype) / 2.0
    return A * x_norm + A - 0.5


_grid_cache = {}


def get_test_coords(img_size, device, batch_size=1):
    """(batch, h*w, 2) row/col of every pixel (coords.py:31-36).  READ-ONLY: one tensor per (size, device, batch) is cached and
    shared by every caller (the tracker's pyramid and the two-frame initialiser hold the same object) -- clone it before any
    in-place use."""
    h, w = int(img_size[0]), int(img_size[1])
    key = (h, w, str(torch.device(device)), int(batch_size))
    t = _grid_cache.get(key)
    if t is None:
        r, c = torch.meshgrid(torch.arange(h, device=device), torch.arange(w, device=device), indexing="ij")
        t = _grid_cache[key] = torch.stack((r.reshape(-1), c.reshape(-1)), dim=1).repeat(batch_size, 1, 1)
    return t


def get_coord_img(img_size, device, batch_size=1):
    h, w = img_size
    r, c = torch.meshgrid(torch.arange(h, device=device), torch.arange(w, device=device), indexing="ij")
    return torch.stack((r, c), dim=-1).unsqueeze(0).repeat(batch_size, 1, 1, 1)


def fill_image(coords, vals, img_size, default_val=float("nan")):
    """Scatter vals (n,1) to the integer-truncated coords (n,2) of a (1,h,w) image pre-filled with default_val
    (coords.py:50-56).  Where several points land on one pixel the LAST one wins -- the order torch's CPU index_put
    applies them in, made explicit here so the GPU result is deterministic and equal to the reference's."""
    h, w = int(img_size[0]), int(img_size[1])
    c = coords.long()
    flat = c[..., 0] * w + c[..., 1]
    n = flat.numel()
    order = torch.full((h * w,), -1, device=coords.device, dtype=torch.long)
    order.scatter_reduce_(0, flat.reshape(-1), torch.arange(n, device=coords.device), reduce="amax", include_self=True)
    img = torch.full((h * w,), default_val, device=coords.device, dtype=vals.dtype)
    hit = order >= 0
    img[hit] = vals.reshape(-1)[order[hit]]
    return img.view(1, h, w)
