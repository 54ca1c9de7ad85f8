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


# ---- layer programs and composed entry points (include/radnet_hip.h, csrc/program.hip) --------------------------------
OP_CONV_FWD, OP_CONV_DGRAD, OP_CONV_WGRAD, OP_MAXPOOL, OP_COLSUM, OP_WINO, OP_WINO_REUSE, OP_WINO_WGRAD, OP_SCATTER, OP_FILL0, \
    OP_RELU_MASK, OP_ROI_BWD, OP_CONV_BWD, OP_CHAIN, OP_CONV_FWD_PAIR, OP_CONV_BNECK = range(1, 17)
OP_NOP = 0


class Op(C.Structure):
    """Mirror of `radnet_op`: one launch of a layer program."""
    _fields_ = [("kind", C.c_int32), ("i", C.c_int32 * 11), ("p", C.c_void_p * 8), ("conv", ConvDesc)]


class HeadDesc(C.Structure):
    """Mirror of `radnet_head_desc`."""
    _fields_ = [("fmap", C.c_void_p), ("fh", C.c_int32), ("fw", C.c_int32), ("fc", C.c_int32),
                ("rois", C.c_void_p), ("n_rois", C.c_int32), ("pool", C.c_int32), ("pooled", C.c_void_p),
                ("fwd_ops", C.POINTER(Op)), ("n_fwd", C.c_int32),
                ("y5", C.c_void_p), ("hw", C.c_int32), ("feat_c", C.c_int32), ("feat", C.c_void_p),
                ("dense_w", C.c_void_p), ("dense_ld", C.c_int32), ("dense_b", C.c_void_p), ("nc", C.c_int32), ("nreg", C.c_int32),
                ("p_cls", C.c_void_p), ("p_regr", C.c_void_p), ("tail_scratch", C.c_void_p)]


class TileDesc(C.Structure):
    """Mirror of `radnet_tile_desc`."""
    _fields_ = [("img_u8", C.c_void_p), ("h", C.c_int32), ("w", C.c_int32), ("x", C.c_void_p),
                ("base_ops", C.POINTER(Op)), ("n_base", C.c_int32),
                ("rpn_ops", C.POINTER(Op)), ("n_rpn", C.c_int32),
                ("pred", C.c_void_p), ("ld_pred", C.c_int32), ("fh", C.c_int32), ("fw", C.c_int32), ("a", C.c_int32),
                ("anchor_wh_host", C.POINTER(C.c_double)), ("std_scaling", C.c_double), ("overlap_thresh", C.c_double), ("max_boxes", C.c_int32),
                ("R", C.c_void_p), ("Rp", C.c_void_p), ("Rn", C.c_void_p), ("prop_ws", C.c_void_p),
                ("head", C.POINTER(HeadDesc))]


class AdamDesc(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("n", C.c_int64), ("t", C.c_int32), ("lr", C.c_float)]


SUBSAMPLE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.c_int32, C.c_int32, C.c_int32)
SELECT_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int32), C.c_int32)


class HostHooks(C.Structure):
    _fields_ = [("user", C.c_void_p), ("subsample_anchors", SUBSAMPLE_FN), ("select_rois", SELECT_FN)]


class AdamWino(C.Structure):
    """Mirror of `radnet_adam_wino`: a dense [3][3][c][n] kernel inside an optimizer arena and its F(4x4,3x3) filter transform."""
    _fields_ = [("off", C.c_int64), ("c", C.c_int32), ("n", C.c_int32), ("u", C.c_void_p)]


class TrainDesc(C.Structure):
    """Mirror of `radnet_train_desc` (field order is the header's)."""
    _d, _i, _vp = C.c_double, C.c_int32, C.c_void_p
    _fields_ = [("img_u8", _vp), ("h", _i), ("w", _i), ("x", _vp),
                ("gt", _vp), ("gt_is_bg", _vp), ("gt_cls", _vp), ("g", _i), ("width", _i), ("height", _i),
                ("anchor_sizes_host", C.POINTER(_d)), ("ns", _i), ("anchor_ratios_host", C.POINTER(_d)), ("nr", _i),
                ("rpn_stride", _d), ("rpn_max_overlap", _d), ("std_scaling", _d),
                ("valid", _vp), ("overlap", _vp), ("regr", _vp), ("best_anchor", _vp), ("n_for_gt", _vp), ("at_scratch", _vp),
                ("h_valid", _vp), ("h_overlap", _vp), ("y_cls", _vp), ("y_regr", _vp),
                ("base_ops", C.POINTER(Op)), ("n_base", _i),
                ("rpn_fwd_ops", C.POINTER(Op)), ("n_rpn_fwd", _i),
                ("rpn_bwd_ops", C.POINTER(Op)), ("n_rpn_bwd", _i),
                ("rpn_refwd_ops", C.POINTER(Op)), ("n_rpn_refwd", _i),
                ("pred", _vp), ("dz", _vp), ("ld_pred", _i), ("fh", _i), ("fw", _i), ("a", _i), ("bce_mode", _i), ("loss_scratch8", _vp), ("rpn_losses", _vp),
                ("rpn_opt", AdamDesc), ("head_opt", AdamDesc), ("world", _i),
                ("wino_w", _vp), ("wino_c", _i), ("wino_n", _i), ("wino_ldw", _i), ("wino_form", _i), ("wino_u", _vp),
                ("anchor_wh_host", C.POINTER(_d)), ("overlap_thresh", _d), ("max_boxes", _i),
                ("R", _vp), ("Rp", _vp), ("Rn", _vp), ("prop_ws", _vp),
                ("rw", _i), ("rh", _i), ("min_overlap", _d), ("max_overlap", _d), ("regr_std_host4", C.POINTER(_d)), ("bg_class", _i),
                ("keep", _vp), ("roi_cls", _vp), ("roi_box", _vp), ("roi_t", _vp), ("roi_iou", _vp),
                ("h_roi_cls", _vp), ("h_n", _vp), ("sel", _vp), ("h_sel", _vp),
                ("head", C.POINTER(HeadDesc)), ("y1", _vp), ("y2", _vp),
                ("head_dz", _vp), ("det_losses", _vp), ("dense_dw", _vp), ("dense_db", _vp), ("dfeat", _vp), ("g_last", _vp),
                ("head_bwd_ops", C.POINTER(Op)), ("n_head_bwd", _i),
                ("head_shift", _vp), ("head_scale", _vp), ("head_bias", _vp), ("head_t0", _vp), ("head_bias_len", C.c_int64),
                ("tail_scratch", _vp), ("head_wino", _vp), ("n_head_wino", C.c_int32)]


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
        "radnet_set_deterministic": (C.c_int, [vp, C.c_int]),
        "radnet_get_deterministic": (C.c_int, [vp]),
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
        "radnet_winograd4_filter": (C.c_int, [vp, vp, i32, i32, i32, vp]),
        "radnet_winograd4_input": (C.c_int, [vp, vp, i32, i32, i32, i32, vp]),
        "radnet_winograd4_output": (C.c_int, [vp, vp, i32, i32, i32, i32, vp, vp, i32, vp, i32]),
        "radnet_winograd4_dy": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, vp, vp]),
        "radnet_winograd4_filter_grad": (C.c_int, [vp, vp, i32, i32, i32, vp, i32]),
        "radnet_adam_step_fused": (C.c_int, [vp, vp, vp, vp, vp, C.c_int64, i32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, i32,
                                            C.c_int64, C.c_int64, vp, vp, vp, C.POINTER(AdamWino), i32]),
        "radnet_conv_wgrad": (C.c_int, [vp, C.POINTER(ConvDesc)]),
        "radnet_conv_bwd": (C.c_int, [vp, C.POINTER(ConvDesc)]),
        "radnet_conv_fwd_pair": (C.c_int, [vp, C.POINTER(ConvDesc), C.POINTER(ConvDesc)]),
        "radnet_conv_bottleneck": (C.c_int, [vp, C.POINTER(ConvDesc), C.POINTER(ConvDesc), C.POINTER(ConvDesc)]),
        "radnet_colsum": (C.c_int, [vp, vp, i32, i32, i32, vp, vp, i32]),
        "radnet_maxpool_fwd": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, i32, i32]),
        "radnet_roi_resize_fwd": (C.c_int, [vp, vp, i32, i32, i32, vp, i32, i32, vp]),
        "radnet_roi_resize_bwd": (C.c_int, [vp, vp, i32, i32, i32, vp, i32, i32, vp]),
        "radnet_avgpool_fwd": (C.c_int, [vp, vp, i32, i32, i32, vp]),
        "radnet_avgpool_bwd_relu": (C.c_int, [vp, vp, vp, i32, i32, i32, vp]),
        "radnet_dense_heads_fwd": (C.c_int, [vp, vp, i32, i32, vp, i32, vp, i32, i32, vp, vp]),
        "radnet_dense_heads_bwd": (C.c_int, [vp, vp, vp, i32, i32, vp, i32, i32, vp, vp, vp, i32]),
        "radnet_head_tail_scratch_bytes": (u64, [i32]),
        "radnet_head_tail_fwd": (C.c_int, [vp, vp, i32, i32, i32, vp, i32, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp]),
        "radnet_rpn_loss": (C.c_int, [vp, vp, i32, vp, vp, i32, i32, i32, vp, i32, vp, vp]),
        "radnet_det_loss": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, vp, vp]),
        "radnet_adam_step": (C.c_int, [vp, vp, vp, vp, vp, i64, i32, f32, f32, f32, f32, f32, i32]),
        "radnet_adam_step_affine": (C.c_int, [vp, vp, vp, vp, vp, i64, i32, f32, f32, f32, f32, f32, i32, i64, i64, vp, vp, vp]),
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
        "radnet_warp_affine_u8": (C.c_int, [vp, vp, i32, i32, i32, vp, i32, i32, vp, vp]),
        "radnet_fill_zero": (C.c_int, [vp, vp, u64]),
        "radnet_copy_bytes": (C.c_int, [vp, vp, vp, C.c_uint64]),
        "radnet_program_run": (C.c_int, [vp, C.POINTER(Op), i32]),
        "radnet_rpn_forward": (C.c_int, [vp, C.POINTER(Op), i32, C.POINTER(Op), i32]),
        "radnet_predict_tile": (C.c_int, [vp, C.POINTER(TileDesc)]),
        "radnet_train_step": (C.c_int, [vp, C.POINTER(TrainDesc), C.POINTER(HostHooks), C.POINTER(C.c_float), C.POINTER(i32)]),
        "radnet_comm_unique_id": (C.c_int, [C.c_char_p]),
        "radnet_comm_init": (C.c_int, [vp, i32, i32, C.c_char_p]),
        "radnet_comm_destroy": (C.c_int, [vp]),
        "radnet_allreduce_grads": (C.c_int, [vp, vp, i64]),
        "radnet_comm_stats": (C.c_int, [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
        "radnet_chain_build": (C.c_int, [vp, vp, i32, i32, C.POINTER(vp)]),
        "radnet_chain_run": (C.c_int, [vp, vp]),
        "radnet_chain_error": (C.c_uint32, [vp]),
        "radnet_chain_status": (C.c_int, [vp, vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(f64), C.POINTER(f64)]),
        "radnet_chain_destroy": (None, [vp]),
        "radnet_chain_peek": (C.c_int, [vp, i32, C.POINTER(C.c_uint32), i32]),
        "radnet_chain_check": (C.c_int, [vp, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.c_char_p, i32]),
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
