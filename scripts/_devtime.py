"""Device time of a short sequence of library launches: recorded once as a command list (mf_cmdlist_*) and re-issued from C, so that the
number is the GPU's, not the Python wrapper's (a ctypes call costs more than a 10 us kernel runs)."""
import ctypes

import torch

from medfusion_amd import kernels as K
from medfusion_amd import lib as L


def device_us(fn, reps=100, warm=10):
    """fn() issues launches of this library only, into buffers that stay alive (the caller keeps what fn returns).  -> (us per call, launches)"""
    keep = fn()
    torch.cuda.synchronize()
    lib, handle = L.load(), ctypes.c_void_p()
    L.check(lib.mf_cmdlist_begin(), "mf_cmdlist_begin")
    keep = fn()
    L.check(lib.mf_cmdlist_end(ctypes.byref(handle)), "mf_cmdlist_end")
    n = lib.mf_cmdlist_count(handle)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    L.check(lib.mf_cmdlist_replay(handle, warm, K.stream()), "mf_cmdlist_replay")
    ev[0].record()
    L.check(lib.mf_cmdlist_replay(handle, reps, K.stream()), "mf_cmdlist_replay")
    ev[1].record()
    torch.cuda.synchronize()
    lib.mf_cmdlist_free(handle)
    del keep
    return ev[0].elapsed_time(ev[1]) / reps * 1e3, n
