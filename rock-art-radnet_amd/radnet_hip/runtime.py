"""Process-wide default device objects for the function-style API of the drop-in `faster_rcnn` package
(rpn.rpn_to_roi, rpn.calc_iou, utils.calc_region_props, ... take NumPy arrays and need a context to run on)."""
import ctypes as C

import numpy as np
import torch

from . import lib as L

_ctx = None
_scratch = {}


def default_context():
    """One radnet context on the current CUDA device (created on first use; raises without an MI355X)."""
    global _ctx
    if _ctx is None:
        _ctx = L.Context(torch.cuda.current_device() if torch.cuda.is_available() else 0)
    return _ctx


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
