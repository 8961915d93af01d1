import enum
import torch


class InterpolationMode(enum.Enum):
    BILINEAR = "bilinear"
    NEAREST = "nearest"


def resize(img, size, interpolation=InterpolationMode.BILINEAR, antialias=True):
    return torch.nn.functional.interpolate(
        img, size=list(size), mode=interpolation.value, antialias=antialias, align_corners=False
    )


def rgb_to_grayscale(img):
    r, g, b = img.unbind(dim=-3)
    return (0.2989 * r + 0.587 * g + 0.114 * b).unsqueeze(-3)
