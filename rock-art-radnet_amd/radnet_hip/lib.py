"""ctypes binding of libradnet_hip.so (C ABI declared in include/radnet_hip.h).

Thin by design: every function takes raw device pointers (torch-ROCm `tensor.data_ptr()`), sizes and
a context; no torch types cross the boundary.  There is NO CPU fallback: if the shared library is
missing or no gfx950 device is present, loading / context creation raises.
"""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RADNET_HIP_LIBRARY", os.path.join(_HERE, "libradnet_hip.so"))    # override: A/B builds, diag twin
HEADER_PATH = os.path.abspath(os.path.join(_HERE, "..", "..", "include", "radnet_hip.h"))

c_float_p = C.c_void_p      # device pointers travel as integers
_lib = None


class RadnetError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    """Mirror of `radnet_conv_desc` (include/radnet_hip.h)."""
    _fields_ = [
        ("x", C.c_void_p), ("w", C.c_void_p), ("y", C.c_void_p),
        ("scale", C.c_void_p), ("shift", C.c_void_p), ("addend", C.c_void_p),
        ("nb", C.c_int32), ("h", C.c_int32), ("w_", C.c_int32), ("c", C.c_int32),
        ("oh", C.c_int32), ("ow", C.c_int32),
        ("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32), ("pad_t", C.c_int32), ("pad_l", C.c_int32),
        ("n", C.c_int32),
        ("ldw", C.c_int32), ("ldy", C.c_int32), ("ld_add", C.c_int32),
        ("act", C.c_int32), ("act_cols", C.c_int32),
        ("dy", C.c_void_p), ("gscale", C.c_void_p), ("dx", C.c_void_p), ("dx_add", C.c_void_p), ("dx_mask", C.c_void_p),
        ("dw", C.c_void_p),
        ("ld_dy", C.c_int32), ("ld_dx", C.c_int32), ("ld_dx_add", C.c_int32), ("ld_dx_mask", C.c_int32),
        ("dw_accumulate", C.c_int32),
        ("db", C.c_void_p),
    ]


def declared_symbols():
    """Names of every function the public header declares."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(radnet_[a-z0-9_]+)\s*\(", text)))


def load_library():
    """dlopen the in-tree library; raises RadnetError (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64; import it first so this library binds to the SAME HIP runtime as the
    # tensors whose pointers it receives (two runtimes in one process do not share devices or allocations)
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RadnetError("libradnet_hip.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(or `make -C rock-art-radnet_amd/csrc`). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    i32, i64, u64, f32, f64, vp = C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_double, C.c_void_p
    sig = {
        "radnet_create": (C.c_int, [C.c_int, vp, C.POINTER(vp)]),
        "radnet_destroy": (None, [vp]),
        "radnet_last_error": (C.c_char_p, [vp]),
        "radnet_sync": (C.c_int, [vp]),
        "radnet_set_stream": (C.c_int, [vp, vp]),
        "radnet_version": (C.c_int, []),
        "radnet_set_workspace": (C.c_int, [vp, vp, u64]),
        "radnet_set_autotune": (C.c_int, [vp, C.c_int]),
        "radnet_tuned_shapes": (C.c_int, [vp]),
        "radnet_tune_save": (C.c_int, [vp, C.c_char_p]),
        "radnet_share_tuning": (C.c_int, [vp, vp]),
        "radnet_tune_load": (C.c_int, [vp, C.c_char_p]),
        "radnet_force_config": (C.c_int, [vp, C.c_int, C.c_int, C.c_int]),
        "radnet_force_waves": (C.c_int, [vp, C.c_int]),
        "radnet_timing_enable": (C.c_int, [vp, C.c_int]),
        "radnet_timing_read": (C.c_int, [vp, C.c_int, C.POINTER(f64), C.POINTER(i64), C.POINTER(f64)]),
        "radnet_timing_reset": (C.c_int, [vp]),
        "radnet_conv_fwd": (C.c_int, [vp, C.POINTER(ConvDesc)]),
        "radnet_conv_dgrad": (C.c_int, [vp, C.POINTER(ConvDesc)]),
        "radnet_gemm_batched": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32]),
        "radnet_winograd_filter": (C.c_int, [vp, vp, i32, i32, i32, vp]),
        "radnet_winograd_input": (C.c_int, [vp, vp, i32, i32, i32, i32, vp]),
        "radnet_winograd_output": (C.c_int, [vp, vp, i32, i32, i32, i32, vp, vp, i32, vp, i32]),
        "radnet_winograd_dy": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, vp, vp]),
        "radnet_wgrad_batched": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32]),
        "radnet_winograd_filter_grad": (C.c_int, [vp, vp, i32, i32, i32, vp, i32]),
        "radnet_conv_wgrad": (C.c_int, [vp, C.POINTER(ConvDesc)]),
        "radnet_colsum": (C.c_int, [vp, vp, i32, i32, i32, vp, vp, i32]),
        "radnet_maxpool_fwd": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, i32, i32]),
        "radnet_roi_resize_fwd": (C.c_int, [vp, vp, i32, i32, i32, vp, i32, i32, vp]),
        "radnet_roi_resize_bwd": (C.c_int, [vp, vp, i32, i32, i32, vp, i32, i32, vp]),
        "radnet_avgpool_fwd": (C.c_int, [vp, vp, i32, i32, i32, vp]),
        "radnet_avgpool_bwd_relu": (C.c_int, [vp, vp, vp, i32, i32, i32, vp]),
        "radnet_dense_heads_fwd": (C.c_int, [vp, vp, i32, i32, vp, i32, vp, i32, i32, vp, vp]),
        "radnet_dense_heads_bwd": (C.c_int, [vp, vp, vp, i32, i32, vp, i32, i32, vp, vp, vp, i32]),
        "radnet_rpn_loss": (C.c_int, [vp, vp, i32, vp, vp, i32, i32, i32, vp, i32, vp, vp]),
        "radnet_det_loss": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, vp, vp]),
        "radnet_adam_step": (C.c_int, [vp, vp, vp, vp, vp, i64, i32, f32, f32, f32, f32, f32, i32]),
        "radnet_affine_vec": (C.c_int, [vp, vp, vp, vp, vp, i64]),
        "radnet_relu_mask": (C.c_int, [vp, vp, vp, i64]),
        "radnet_scatter_strided": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp]),
        "radnet_proposals_ws_bytes": (u64, [i64]),
        "radnet_rpn_to_roi": (C.c_int, [vp, vp, i32, i32, i32, i32, C.POINTER(f64), f64, i32, f64, i32, vp, vp, vp, vp]),
        "radnet_nms": (C.c_int, [vp, vp, vp, i32, f64, i32, vp, vp, vp]),
        "radnet_anchor_targets": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, C.POINTER(f64), i32, C.POINTER(f64), i32,
                                            f64, f64, vp, vp, vp, vp, vp, vp]),
        "radnet_anchor_targets_pack": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, f64, vp, vp]),
        "radnet_roi_targets": (C.c_int, [vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, f64, f64, f64, C.POINTER(f64), i32,
                                         vp, vp, vp, vp, vp, vp]),
        "radnet_roi_batch_pack": (C.c_int, [vp, vp, i32, vp, vp, vp, i32, i32, vp, vp, vp]),
        "radnet_host_choice_round": (i64, [vp, vp, vp, vp, i64, vp, i64, vp, vp]),
        "radnet_preprocess_bgr": (C.c_int, [vp, vp, i32, i32, i32, vp]),
        "radnet_resize_bicubic_u8": (C.c_int, [vp, vp, i32, i32, vp, i32, i32, i32]),
        "radnet_fill_zero": (C.c_int, [vp, vp, u64]),
        "radnet_scale": (C.c_int, [vp, vp, i64, f32]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)          # AttributeError here = header/binding drift: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _ptr(t):
    """Device pointer of a torch tensor (or None / int passthrough)."""
    if t is None:
        return None
    if isinstance(t, int):
        return t
    return t.data_ptr()


class Context:
    """One `radnet_ctx` bound to a device and a HIP stream (default: torch's current stream)."""

    def __init__(self, device_index=0, stream_handle=None):
        import torch
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise RadnetError("no MI355X visible: the RADNet HIP path has no CPU fallback")
        if stream_handle is None:
            with torch.cuda.device(device_index):
                stream_handle = torch.cuda.current_stream().cuda_stream
        h = C.c_void_p()
        rc = self.lib.radnet_create(int(device_index), C.c_void_p(stream_handle), C.byref(h))
        if rc != 0:
            raise RadnetError("radnet_create failed with code %d (needs a gfx950 device)" % rc)
        self.h = h
        self.device_index = device_index
        self.stream_handle = stream_handle
        self.timing_on = False

    def close(self):
        if getattr(self, "h", None):
            self.lib.radnet_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc, what=""):
        if rc != 0:
            msg = self.lib.radnet_last_error(self.h)
            raise RadnetError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))

    def call(self, name, *args):
        fn = getattr(self.lib, name)
        self.check(fn(self.h, *[_ptr(a) if hasattr(a, "data_ptr") else a for a in args]), name)

    def sync(self):
        self.check(self.lib.radnet_sync(self.h), "radnet_sync")

    def set_stream(self, stream_handle):
        self.check(self.lib.radnet_set_stream(self.h, C.c_void_p(stream_handle)), "radnet_set_stream")
        self.stream_handle = stream_handle

    # timing of the GEMM-class launches (HIP events on the ctx stream)
    def timing(self, enable):
        self.check(self.lib.radnet_timing_enable(self.h, 1 if enable else 0), "timing_enable")
        self.timing_on = bool(enable)

    def timing_reset(self):
        self.check(self.lib.radnet_timing_reset(self.h), "timing_reset")

    def timing_read(self, cls):
        ms, n, fl = C.c_double(), C.c_int64(), C.c_double()
        self.check(self.lib.radnet_timing_read(self.h, cls, C.byref(ms), C.byref(n), C.byref(fl)), "timing_read")
        return ms.value, n.value, fl.value
