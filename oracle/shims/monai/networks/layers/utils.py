import torch
import torch.nn as nn


class Swish(nn.Module):
    def __init__(self, alpha=1.0):
        super().__init__()
        self.alpha = alpha

    def forward(self, x):
        return x * torch.sigmoid(self.alpha * x)


def _split(name):
    if isinstance(name, (tuple, list)):
        return name[0], dict(name[1]) if len(name) > 1 else {}
    return name, {}


def get_act_layer(name):
    n, kw = _split(name)
    n = n.lower()
    if n == "swish":
        return Swish(**kw)
    if n == "relu":
        return nn.ReLU(**kw)
    if n == "leakyrelu":
        return nn.LeakyReLU(**kw)
    if n == "gelu":
        return nn.GELU(**kw)
    raise KeyError(n)


def get_norm_layer(name, spatial_dims=1, channels=1):
    n, kw = _split(name)
    n = n.lower()
    if n == "group":
        return nn.GroupNorm(num_channels=channels, **kw)
    if n == "batch":
        return (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)[spatial_dims - 1](channels, **kw)
    if n == "instance":
        return (nn.InstanceNorm1d, nn.InstanceNorm2d, nn.InstanceNorm3d)[spatial_dims - 1](channels, **kw)
    raise KeyError(n)


def get_dropout_layer(name, dropout_dim=1):
    if isinstance(name, (int, float)):
        return (nn.Dropout, nn.Dropout2d, nn.Dropout3d)[dropout_dim - 1](p=float(name))
    n, kw = _split(name)
    return (nn.Dropout, nn.Dropout2d, nn.Dropout3d)[dropout_dim - 1](**kw)
