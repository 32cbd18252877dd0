import torch.nn as nn


class _ConvFactory:
    CONV = "conv"
    CONVTRANS = "convtrans"

    def __getitem__(self, key):
        name, dim = key
        name = name.lower()
        if name == "conv":
            return (nn.Conv1d, nn.Conv2d, nn.Conv3d)[dim - 1]
        if name == "convtrans":
            return (nn.ConvTranspose1d, nn.ConvTranspose2d, nn.ConvTranspose3d)[dim - 1]
        raise KeyError(name)


class _PoolFactory:
    def __getitem__(self, key):
        name, dim = key
        name = name.lower()
        if name == "avg":
            return (nn.AvgPool1d, nn.AvgPool2d, nn.AvgPool3d)[dim - 1]
        if name == "max":
            return (nn.MaxPool1d, nn.MaxPool2d, nn.MaxPool3d)[dim - 1]
        raise KeyError(name)


Conv = _ConvFactory()
Pool = _PoolFactory()
