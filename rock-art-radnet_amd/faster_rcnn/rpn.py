"""Region-proposal glue of the drop-in package: same function names, arguments and return conventions as the
reference's faster_rcnn/rpn.py, with the arithmetic done by hand-written gfx950 kernels (libradnet_hip.so).

    rpn_layer                   layer spec of the RPN head                  (rpn.py:12-66)
    rpn_to_roi                  decode + clip + drop degenerate + greedy NMS  (rpn.py:68-172)  -> radnet_rpn_to_roi
    calc_iou                    RoI labelling for the classifier head        (rpn.py:176-296) -> radnet_roi_targets
    apply_regr                  scalar delta decode (host, Python floats)    (rpn.py:346-378)
    non_max_suppression_fast    greedy NMS                                   (rpn.py:380-455) -> radnet_nms

Everything takes / returns host NumPy arrays exactly like the reference; device-resident fast paths live in
radnet_hip.engine / radnet_hip.trainer.
"""
import ctypes as C
import math

import numpy as np

from . import utils


class RPNSpec:
    """What rpn_layer() returns in place of Keras tensors: a description the model builder binds to an engine."""

    def __init__(self, base, num_anchors):
        self.base = base
        self.num_anchors = num_anchors
        self.layers = (("rpn_conv1", 3, 512, "relu"), ("rpn_out_class", 1, num_anchors, "sigmoid"), ("rpn_out_regress", 1, 4 * num_anchors, "linear"))


def rpn_layer(input_layer, num_anchors):
    """rpn.py:12-66: 3x3/512 ReLU conv, then 1x1 sigmoid (objectness) and 1x1 linear (4 deltas per anchor).
    Returns [x_class, x_regr, input_layer] like the reference (spec objects instead of Keras tensors)."""
    spec = RPNSpec(input_layer, num_anchors)
    return [("rpn_out_class", spec), ("rpn_out_regress", spec), input_layer]


def _anchor_wh(C_cfg):
    return np.array([[(s * r[0]) / C_cfg.rpn_stride, (s * r[1]) / C_cfg.rpn_stride] for s in C_cfg.anchor_box_scales for r in C_cfg.anchor_box_ratios],
                    dtype=np.float64)


def rpn_to_roi(rpn_layer, regr_layer, C_cfg, use_regr=True, max_boxes=300, overlap_thresh=0.9):
    """rpn.py:68-172.  rpn_layer (1,H,W,A) scores, regr_layer (1,H,W,4A) deltas (fp32 NumPy) -> (n,4) int64 boxes
    (x1,y1,x2,y2) in feature-map units, n <= max_boxes."""
    import torch
    from radnet_hip import runtime as rt
    assert rpn_layer.shape[0] == 1                                        # rpn.py:96
    rows, cols, A = rpn_layer.shape[1:4]
    ctx = rt.default_context()
    pred = np.zeros((rows * cols, 5 * A), np.float32)
    pred[:, :A] = np.asarray(rpn_layer, dtype=np.float32).reshape(-1, A)
    pred[:, A:] = np.asarray(regr_layer, dtype=np.float32).reshape(-1, 4 * A)
    pd = rt.to_dev(pred)
    R = torch.zeros(max_boxes, 4, dtype=torch.int64, device="cuda")
    Rp = torch.zeros(max_boxes, dtype=torch.float32, device="cuda")
    Rn = torch.zeros(1, dtype=torch.int32, device="cuda")
    ws = rt.scratch("proposals", ctx.lib.radnet_proposals_ws_bytes(rows * cols * A))
    awh = _anchor_wh(C_cfg)
    rc = ctx.lib.radnet_rpn_to_roi(ctx.h, pd.data_ptr(), 5 * A, rows, cols, A, rt.f64_ptr(awh), float(C_cfg.std_scaling), 1 if use_regr else 0,
                                   float(overlap_thresh), int(max_boxes), R.data_ptr(), Rp.data_ptr(), Rn.data_ptr(), ws.data_ptr())
    ctx.check(rc, "radnet_rpn_to_roi")
    n = int(Rn.cpu()[0])
    if n <= 0:
        # the reference unpacks the [] its NMS returns for "no boxes" and dies with ValueError (rpn.py:170,391-392)
        raise ValueError("not enough values to unpack (expected 2, got 0)")
    return R[:n].cpu().numpy()


def calc_iou(R, img_data, C_cfg, class_mapping):
    """rpn.py:176-296: label each proposal against the ground truth.  Returns (X (1,n,4) xywh, Y1 (1,n,nc) one-hot,
    Y2 (1,n,8(nc-1)) [labels || std-scaled targets], IoUs) or (None, None, None, None)."""
    import torch
    from radnet_hip import runtime as rt
    bboxes = img_data["bboxes"]
    width, height = img_data["width"], img_data["height"]
    rw, rh = utils.get_new_img_size(width, height, C_cfg.img_size)
    nc = len(class_mapping)
    bg = class_mapping["bg"]
    n = int(R.shape[0])
    if n == 0 or len(bboxes) == 0:
        return None, None, None, None
    ctx = rt.default_context()
    gt = rt.to_dev(np.array([[b["x1"], b["y1"], b["x2"], b["y2"]] for b in bboxes], dtype=np.float64))
    gc = rt.to_dev(np.array([class_mapping[b["class"]] for b in bboxes], dtype=np.int32))
    Rd = rt.to_dev(np.asarray(R).round().astype(np.int64))                 # rpn.py:211-214: int(round(.)) of each coordinate
    keep = torch.zeros(n, dtype=torch.uint8, device="cuda")
    cls = torch.zeros(n, dtype=torch.int32, device="cuda")
    box = torch.zeros(n, 4, dtype=torch.int32, device="cuda")
    t = torch.zeros(n, 4, dtype=torch.float64, device="cuda")
    iou = torch.zeros(n, dtype=torch.float64, device="cuda")
    std = np.array(C_cfg.classifier_regr_std, dtype=np.float64)
    rc = ctx.lib.radnet_roi_targets(ctx.h, Rd.data_ptr(), n, gt.data_ptr(), gc.data_ptr(), len(bboxes), int(width), int(height), int(rw), int(rh),
                                    float(C_cfg.rpn_stride), float(C_cfg.classifier_min_overlap), float(C_cfg.classifier_max_overlap),
                                    rt.f64_ptr(std), int(bg), keep.data_ptr(), cls.data_ptr(), box.data_ptr(), t.data_ptr(), iou.data_ptr(), None)
    ctx.check(rc, "radnet_roi_targets")
    k = keep.cpu().numpy().astype(bool)
    if not k.any():
        return None, None, None, None
    c = cls.cpu().numpy()[k]
    X = box.cpu().numpy()[k].astype(np.int64)
    tt = t.cpu().numpy()[k]
    m = len(c)
    Y1 = np.zeros((m, nc), dtype=np.int64)
    Y1[np.arange(m), c] = 1
    lab = np.zeros((m, 4 * (nc - 1)))
    coords = np.zeros((m, 4 * (nc - 1)))
    fg = np.nonzero(c != bg)[0]
    for q in range(4):
        lab[fg, 4 * c[fg] + q] = 1
        coords[fg, 4 * c[fg] + q] = tt[fg, q]
    Y2 = np.concatenate([lab, coords], axis=1)
    return np.expand_dims(X, 0), np.expand_dims(Y1, 0), np.expand_dims(Y2, 0), iou.cpu().numpy()[k].tolist()


def apply_regr(x, y, w, h, tx, ty, tw, th):
    """rpn.py:346-378: scalar decode with math.exp and Python round(); on ValueError / OverflowError the input box
    comes back unchanged."""
    try:
        cx1 = tx * w + (x + w / 2.)
        cy1 = ty * h + (y + h / 2.)
        w1 = math.exp(tw) * w
        h1 = math.exp(th) * h
        return int(round(cx1 - w1 / 2.)), int(round(cy1 - h1 / 2.)), int(round(w1)), int(round(h1))
    except (ValueError, OverflowError):
        return x, y, w, h


def non_max_suppression_fast(boxes, probs, overlap_thresh=0.9, max_boxes=300):
    """rpn.py:380-455 on the device (fp64, same suppression test).  Returns (boxes[pick].astype(int), probs[pick]);
    [] for no boxes (rpn.py:391-392); AssertionError for malformed boxes (rpn.py:400-401).
    Tie rule among equal scores: stable ascending order walked from the end (higher index first)."""
    import torch
    from radnet_hip import runtime as rt
    boxes = np.asarray(boxes)
    if len(boxes) == 0:
        return []
    probs = np.asarray(probs)
    ctx = rt.default_context()
    n = int(boxes.shape[0])
    bd = rt.to_dev(boxes[:, :4], dtype=np.float64)
    pd = rt.to_dev(probs, dtype=np.float32)
    idx = torch.zeros(max(1, min(int(max_boxes), 1024)), dtype=torch.int32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    ws = rt.scratch("nms", ctx.lib.radnet_proposals_ws_bytes(n))
    ctx.call("radnet_nms", bd, pd, n, C.c_double(float(overlap_thresh)), int(min(max_boxes, 1024)), idx, cnt, ws)
    k = int(cnt.cpu()[0])
    if k < 0:
        raise AssertionError("non_max_suppression_fast: box with x1 >= x2 or y1 >= y2")
    pick = idx.cpu().numpy()[:k]
    out_boxes = boxes[pick].astype("int")
    return out_boxes, probs[pick]
