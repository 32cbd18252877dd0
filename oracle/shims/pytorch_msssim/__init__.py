import torch.nn as nn


class SSIM(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()


def ssim(*a, **k):
    raise RuntimeError("pytorch_msssim shim")
