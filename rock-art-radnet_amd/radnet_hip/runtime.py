"""Process-wide default device objects for the function-style API of the drop-in `faster_rcnn` package
(rpn.rpn_to_roi, rpn.calc_iou, utils.calc_region_props, ... take NumPy arrays and need a context to run on)."""
import ctypes as C

import numpy as np
import torch

from . import lib as L

import threading

_ctx = None
_scratch = {}
_local = threading.local()


def default_context():
    """One radnet context on the current CUDA device (created on first use; raises without an MI355X).  The main thread's
    context runs on the stream that was current when it was created; any other thread (data_feed.BackgroundFeed's worker
    resizing tiles) gets a context of its own on a stream of its own -- a context is not shared between threads, and the
    worker's device work must not queue behind the train step's."""
    global _ctx
    if threading.current_thread() is threading.main_thread():
        if _ctx is None:
            _ctx = L.Context(torch.cuda.current_device() if torch.cuda.is_available() else 0)
        return _ctx
    if getattr(_local, "ctx", None) is None:
        _local.stream = torch.cuda.Stream()
        _local.ctx = L.Context(torch.cuda.current_device(), stream_handle=_local.stream.cuda_stream)
    return _local.ctx


def thread_stream():
    """The HIP stream of this thread's default context (None on the main thread: torch's current stream)."""
    return getattr(_local, "stream", None) if threading.current_thread() is not threading.main_thread() else None


def scratch(name, nbytes):
    """Grow-only named device scratch buffers (uint8)."""
    t = _scratch.get(name)
    if t is None or t.numel() < nbytes:
        t = torch.empty(int(nbytes), dtype=torch.uint8, device="cuda")
        _scratch[name] = t
    return t


def to_dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=dtype))
    return t.cuda()


def f64_ptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))
