"""Scratch stand-in for torchvision (absent): Normalize / resize / rgb_to_grayscale only."""
from . import transforms  # noqa: F401
