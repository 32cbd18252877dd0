import torch


def load(path, map_location=None):
    return torch.load(path, map_location=map_location, weights_only=False)
