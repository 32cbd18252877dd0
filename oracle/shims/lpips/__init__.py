import torch.nn as nn


class LPIPS(nn.Module):
    def __init__(self, *a, **k):
        raise RuntimeError("lpips shim: construct VAE(..., perceiver=None)")
