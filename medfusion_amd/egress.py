"""Image egress for bulk generation (reference: scripts/helpers/sample_dataset.py:44-53, scripts/sample.py:49-51).

The reference copies every chunk of fp32 images to the host, clips / scales / casts them there and PNG-encodes them one by one while
the GPU idles.  Here the clip -> uint8 -> NHWC conversion runs on the device (`mf_image_egress_u8`, bit-exact with the numpy formula),
the uint8 pixels (1/4 of the bytes, already in the file layout) go to PINNED host buffers on a side stream with an asynchronous copy,
and a writer thread encodes the files -- so the next chunk's `sample()` overlaps both the copy and the encoding.
"""
from __future__ import annotations

import queue
import threading
from pathlib import Path
from typing import Callable, Optional

import numpy as np
import torch

from . import kernels as K


def save_png(arr_hwc: np.ndarray, path) -> None:
    """uint8 [H, W, C] -> `Image.fromarray(image).convert("RGB").save(path)` (sample_dataset.py:52); .npy next to it without PIL"""
    img = arr_hwc[..., 0] if arr_hwc.shape[-1] == 1 else arr_hwc
    try:
        from PIL import Image

        Image.fromarray(img).convert("RGB").save(str(path))
    except ImportError:
        np.save(str(path) + ".npy", img)


class AsyncImageWriter:
    """submit(images NCHW fp32 on the device, paths) returns at once; close() waits for everything to be on disk.

    Two pinned buffers alternate: a buffer is reused only after the writer thread has finished the files of its previous chunk."""

    def __init__(self, device, normalize_each: bool = False, sink: Optional[Callable[[np.ndarray, object], None]] = save_png, depth: int = 2, threads: int = 2):
        """threads: encoder threads the files of one chunk are spread over (PIL's PNG encoder releases the GIL inside zlib: 2 threads keep up with
        ~400 images/s of 256 x 256 RGB; the chunk-200 bulk run of one MI355X needs ~50)"""
        self.device, self.normalize_each, self.sink = device, normalize_each, sink
        self.threads = max(1, int(threads))
        self.pool = None
        if self.threads > 1 and sink is not None:
            from concurrent.futures import ThreadPoolExecutor
            self.pool = ThreadPoolExecutor(max_workers=self.threads)
        self.copy_stream = torch.cuda.Stream(device=device)
        self.slots = [dict(buf=None, free=threading.Event()) for _ in range(depth)]
        for s in self.slots:
            s["free"].set()
        self.turn = 0
        self.q: "queue.Queue" = queue.Queue()
        self.err: Optional[BaseException] = None
        self.images = 0
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def _run(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            slot, n, done, paths = item
            try:
                done.synchronize()                       # the asynchronous device -> pinned-host copy of this chunk has landed
                arr = slot["buf"][:n].numpy()
                if self.sink is not None:
                    if self.pool is not None:
                        list(self.pool.map(lambda ip: self.sink(arr[ip[0]], ip[1]), enumerate(paths)))   # (list: re-raises an encoder's exception here)
                    else:
                        for i, p in enumerate(paths):
                            self.sink(arr[i], p)
            except BaseException as e:  # noqa: BLE001  (reported by close())
                self.err = e
            finally:
                slot["free"].set()

    def submit(self, images_nchw: torch.Tensor, paths) -> None:
        if self.err is not None:
            raise RuntimeError("image writer thread failed") from self.err
        u8 = K.image_to_uint8(images_nchw, normalize_each=self.normalize_each)   # [N, H, W, C] uint8 on the device, current stream
        slot = self.slots[self.turn]
        self.turn = (self.turn + 1) % len(self.slots)
        slot["free"].wait()
        slot["free"].clear()
        if slot["buf"] is None or slot["buf"].shape[0] < u8.shape[0] or slot["buf"].shape[1:] != u8.shape[1:]:
            slot["buf"] = torch.empty(u8.shape, dtype=torch.uint8, pin_memory=True)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))
        done = torch.cuda.Event()
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(ready)
            slot["buf"][: u8.shape[0]].copy_(u8, non_blocking=True)
            done.record(self.copy_stream)
        u8.record_stream(self.copy_stream)
        self.images += u8.shape[0]
        self.q.put((slot, u8.shape[0], done, list(paths)))

    def close(self) -> int:
        self.q.put(None)
        self.thread.join()
        if self.pool is not None:
            self.pool.shutdown(wait=True)
        if self.err is not None:
            raise RuntimeError("image writer thread failed") from self.err
        return self.images
