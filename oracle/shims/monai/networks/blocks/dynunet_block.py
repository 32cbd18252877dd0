import numpy as np


def get_padding(kernel_size, stride):
    k = np.atleast_1d(kernel_size)
    s = np.atleast_1d(stride)
    p = (k - s + 1) / 2
    if np.min(p) < 0:
        raise AssertionError("negative padding")
    p = tuple(int(v) for v in p)
    return p if len(p) > 1 else p[0]


def get_output_padding(kernel_size, stride, padding):
    k = np.atleast_1d(kernel_size)
    s = np.atleast_1d(stride)
    p = np.atleast_1d(padding)
    o = 2 * p + s - k
    o = tuple(int(v) for v in o)
    return o if len(o) > 1 else o[0]
