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
        # The current device is per thread in HIP / torch and a new thread starts on device 0: a worker of rank r's process
        # must follow the device its creator uses (bind_thread_device), or every rank's feed worker would put its stream,
        # context and resize buffers on GPU 0.
        dev = getattr(_local, "device", None)
        if dev is None:
            dev = _owner_device if _owner_device is not None else torch.cuda.current_device()
        torch.cuda.set_device(dev)
        _local.stream = torch.cuda.Stream(device=dev)
        _local.ctx = L.Context(dev, stream_handle=_local.stream.cuda_stream)
    return _local.ctx


_owner_device = None


def note_owner_device():
    """Called on the thread that CREATES a worker (data_feed.BackgroundFeed.__init__): remembers its current device."""
    global _owner_device
    if torch.cuda.is_available():
        _owner_device = torch.cuda.current_device()
    return _owner_device


def bind_thread_device(dev):
    """First thing a worker thread does: make `dev` (its creator's device) its own current device."""
    if dev is not None and torch.cuda.is_available():
        torch.cuda.set_device(dev)
        _local.device = dev


def thread_stream():
    """The HIP stream of this thread's default context (None on the main thread: torch's current stream)."""
    return getattr(_local, "stream", None) if threading.current_thread() is not threading.main_thread() else None


def scratch(name, nbytes):
    """Grow-only named device scratch buffers (uint8), per device."""
    key = (name, torch.cuda.current_device())
    t = _scratch.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(int(nbytes), dtype=torch.uint8, device="cuda")
        _scratch[key] = t
    return t


def to_dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=dtype))
    return t.cuda()


def f64_ptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))
