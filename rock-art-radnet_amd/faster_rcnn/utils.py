"""Hot-path subset of the reference's faster_rcnn/utils.py behind the same names.

    get_new_img_size    utils.py:65-75
    iou / union / intersection   utils.py:77-109   (scalar host helpers; the kernels carry their own copy)
    calc_region_props   utils.py:554-822  anchor targets ("calc_rpn"): device labelling + host RNG subsampling

The CSV / image readers and generators of the reference (get_data, get_generator, get_tile_generator,
SampleSelector) are the host data feed -- out of the hot-path scope.
"""
import numpy as np


def get_new_img_size(width, height, img_min_side=300):
    if width <= height:
        f = float(img_min_side) / width
        return img_min_side, int(f * height)
    f = float(img_min_side) / height
    return int(f * width), img_min_side


def intersection(ai, bi):
    x, y = max(ai[0], bi[0]), max(ai[1], bi[1])
    w, h = min(ai[2], bi[2]) - x, min(ai[3], bi[3]) - y
    return 0 if (w < 0 or h < 0) else w * h


def union(au, bu, area_intersection):
    return (au[2] - au[0]) * (au[3] - au[1]) + (bu[2] - bu[0]) * (bu[3] - bu[1]) - area_intersection


def iou(a, b):
    """(x1,y1,x2,y2) boxes; degenerate -> 0.0; inter / (union + 1e-6)."""
    if a[0] >= a[2] or a[1] >= a[3] or b[0] >= b[2] or b[1] >= b[3]:
        return 0.0
    ai = intersection(a, b)
    return float(ai) / float(union(a, b, ai) + 1e-6)


def calc_region_props(C, img_data, width, height, width_resized, height_resized, get_feat_map_size, verbose=False):
    """utils.py:554-822.  Returns (y_rpn_cls (1,2A,H,W), y_rpn_regr (1,8A,H,W), best_anchor_for_bbox (g,4), n_pos) as
    float64 NCHW arrays, unscaled -- exactly what the reference function returns (its caller applies std_scaling and
    the NHWC transpose, utils.py:475-478).  Labelling runs on the device; the random subsampling consumes NumPy's
    global RNG on the host like the reference.  Raises KeyError where the reference does."""
    import ctypes as Ct
    import torch
    from radnet_hip import engine as E
    from radnet_hip import runtime as rt
    ctx = rt.default_context()
    fw, fh = get_feat_map_size(width_resized, height_resized)
    sizes = np.array(C.anchor_box_scales, dtype=np.float64)
    ratios = np.array(C.anchor_box_ratios, dtype=np.float64).reshape(-1, 2)
    A = len(sizes) * len(ratios)
    bboxes = img_data["bboxes"]
    g = len(bboxes)
    valid = torch.zeros(A, fh, fw, dtype=torch.uint8, device="cuda")
    overlap = torch.zeros_like(valid)
    regr = torch.zeros(fh, fw, 4 * A, dtype=torch.float64, device="cuda")
    best = torch.zeros(max(g, 1), 4, dtype=torch.int32, device="cuda")
    nfor = torch.zeros(max(g, 1), dtype=torch.int32, device="cuda")
    scratch = torch.zeros(max(g, 1), dtype=torch.int64, device="cuda")
    gt = rt.to_dev(np.array([[b["x1"], b["y1"], b["x2"], b["y2"]] for b in bboxes], dtype=np.float64).reshape(-1, 4)) if g else None
    isbg = rt.to_dev(np.array([1 if b["class"] == "bg" else 0 for b in bboxes], dtype=np.int32)) if g else None
    rc = ctx.lib.radnet_anchor_targets(ctx.h, gt.data_ptr() if g else None, isbg.data_ptr() if g else None, g, int(width), int(height),
                                       int(width_resized), int(height_resized), int(fw), int(fh), rt.f64_ptr(sizes), len(sizes), rt.f64_ptr(ratios),
                                       len(ratios), float(C.rpn_stride), float(C.rpn_max_overlap), valid.data_ptr(), overlap.data_ptr(),
                                       regr.data_ptr(), best.data_ptr(), nfor.data_ptr(), scratch.data_ptr())
    ctx.check(rc, "radnet_anchor_targets")
    v = valid.cpu().numpy()
    o = overlap.cpu().numpy()
    n_pos = E.subsample_valid(v, o)
    y_valid = v.astype(np.float64)[None]
    y_overlap = o.astype(np.float64)[None]
    y_regr = np.transpose(regr.cpu().numpy(), (2, 0, 1))[None]
    y_rpn_cls = np.concatenate([y_valid, y_overlap], axis=1)
    y_rpn_regr = np.concatenate([np.repeat(y_overlap, 4, axis=1), y_regr], axis=1)
    best_anchor = best.cpu().numpy()[:g].astype(int) if g else -1 * np.ones((0, 4)).astype(int)
    return np.copy(y_rpn_cls), np.copy(y_rpn_regr), best_anchor, n_pos
