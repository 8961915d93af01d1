"""Scratch stand-in: the reference only uses LightningModule as a checkpoint container."""
import torch.nn as nn


class LightningModule(nn.Module):
    def save_hyperparameters(self, *a, **k):
        pass
