import torch
from . import functional  # noqa: F401


class Normalize(torch.nn.Module):
    def __init__(self, mean, std):
        super().__init__()
        self.mean = torch.as_tensor(mean).view(1, -1, 1, 1)
        self.std = torch.as_tensor(std).view(1, -1, 1, 1)

    def forward(self, x):
        return (x - self.mean.to(x)) / self.std.to(x)
