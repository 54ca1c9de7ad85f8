#!/usr/bin/env python3
"""Generate golden input/output vectors for the NumPy half of the hot path.

Runs ONLY in the build container (needs /root/reference).  The reference's NumPy
glue (rpn.py, utils.py, RADNet.py, config.py) is imported *from where it lies* --
nothing is copied -- with the absent third-party modules (cv2, skimage, keras)
satisfied by empty stubs.  Outputs are small .npz / .json fixtures under
tests/golden/; they hold data only (inputs + the reference's outputs).

    python tools/gen_golden.py            # rewrites tests/golden/*.npz

numpy version is recorded in tests/golden/MANIFEST.json because two behaviours of
the reference depend on it (NEP-50 fp32 comparison in calc_region_props, argsort
tie order in NMS) -- see SURVEY.md 8c.
"""
import json
import os
import sys
import types

import numpy as np

REF = os.environ.get("RADNET_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
sys.dont_write_bytecode = True


def _stub_modules():
    cv2 = types.ModuleType("cv2")
    cv2.INTER_CUBIC = 2

    def _resize(img, size, interpolation=None):
        # OpenCV is absent.  Identity resizes (tile already at the target size) copy; the full-image cases
        # (C.include_full_img: RADNet.py:606-665, utils.py:484-549), where the whole panel is scaled to the
        # network size, are bound to THIS repo's restatement of cv2's 8-bit INTER_CUBIC (oracle/resize.py,
        # parity vs cv2 unpinned): what those goldens pin is the reference's code around the resize.
        w, h = size
        if img.shape[0] == h and img.shape[1] == w:
            return img.copy()
        assert interpolation == cv2.INTER_CUBIC
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
        from oracle import resize as oresize
        return oresize.resize_bicubic_u8(np.ascontiguousarray(img), int(w), int(h))

    cv2.resize = _resize
    sys.modules["cv2"] = cv2
    for name in ("skimage", "skimage.util", "skimage.transform", "skimage.exposure"):
        sys.modules[name] = types.ModuleType(name)
    keras = types.ModuleType("keras")
    kl = types.ModuleType("keras.layers")
    kl.Conv2D = object
    kl.Input = object
    km = types.ModuleType("keras.models")
    km.Model = object
    keras.layers = kl
    keras.models = km
    sys.modules["keras"] = keras
    sys.modules["keras.layers"] = kl
    sys.modules["keras.models"] = km


def _import_reference():
    _stub_modules()
    sys.path.insert(0, REF)
    import faster_rcnn.config as rconfig
    import faster_rcnn.rpn as rrpn
    import faster_rcnn.utils as rutils
    import faster_rcnn.RADNet as rradnet
    assert os.path.realpath(rrpn.__file__).startswith(os.path.realpath(REF))
    return rconfig, rrpn, rutils, rradnet


def resnet_out_len(L):
    L += 6
    for f in (7, 3, 1, 1):
        L = (L - f + 2) // 2
    return L


def feat_size(w, h):
    return resnet_out_len(w), resnet_out_len(h)


def synth_gt(rs, n, W, H, classes, smin=64, smax=400):
    out = []
    for i in range(n):
        bw = int(rs.randint(smin, smax))
        bh = int(rs.randint(smin, smax))
        x1 = int(rs.randint(0, max(1, W - bw)))
        y1 = int(rs.randint(0, max(1, H - bh)))
        out.append({"class": classes[i % len(classes)], "x1": x1, "x2": x1 + bw, "y1": y1, "y2": y1 + bh})
    return out


def gt_to_arrays(bboxes, class_mapping):
    box = np.array([[b["x1"], b["y1"], b["x2"], b["y2"]] for b in bboxes], dtype=np.float64).reshape(-1, 4)
    cls = np.array([class_mapping[b["class"]] for b in bboxes], dtype=np.int64)
    return box, cls


def main():
    rconfig, rrpn, rutils, rradnet = _import_reference()
    os.makedirs(OUT, exist_ok=True)
    manifest = {"numpy": np.__version__, "python": sys.version.split()[0], "files": {}}

    def save(name, **arrs):
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **arrs)
        manifest["files"][name] = sorted(arrs.keys())

    C = rconfig.Config()
    fg = [k for k in C.class_mapping if k != "bg"]

    # ---- Config attribute dump (surface check) -----------------------------------
    with open(os.path.join(OUT, "config_attrs.json"), "w") as f:
        json.dump(C.__dict__, f, indent=1, sort_keys=True)

    # ---- iou / get_new_img_size ---------------------------------------------------
    rs = np.random.RandomState(11)
    a = rs.uniform(0, 100, (400, 4))
    b = rs.uniform(0, 100, (400, 4))
    a[:, 2:] += a[:, :2] * 0.2
    b[:, 2:] += b[:, :2] * 0.2
    a[::17, 2] = a[::17, 0]          # degenerate
    b[::23, 3] = b[::23, 1] - 1.0    # inverted
    r = np.array([rutils.iou(list(x), list(y)) for x, y in zip(a, b)])
    ai = rs.randint(0, 40, (400, 4)); ai[:, 2:] += ai[:, :2] // 2
    bi = rs.randint(0, 40, (400, 4)); bi[:, 2:] += bi[:, :2] // 2
    ri = np.array([rutils.iou([int(v) for v in x], [int(v) for v in y]) for x, y in zip(ai, bi)])
    sizes = [(2000, 1200), (1200, 2000), (600, 600), (2048, 2048), (1333, 777), (640, 481), (3000, 999), (999, 3000)]
    mins = [600, 1000, 300]
    gs = np.array([[w, h, m, *rutils.get_new_img_size(w, h, m)] for (w, h) in sizes for m in mins])
    save("iou", a=a, b=b, iou=r, ai=ai, bi=bi, iou_int=ri, new_img_size=gs)

    # ---- apply_regr_np / apply_regr -------------------------------------------------
    rs = np.random.RandomState(12)
    X = np.stack([rs.uniform(-10, 60, (9, 13)), rs.uniform(-10, 40, (9, 13)),
                  rs.choice([4., 8., 16., 32., 5.5], (9, 13)), rs.choice([4., 8., 16., 32., 7.25], (9, 13))])
    T = (rs.standard_normal((4, 9, 13)) * 0.5).astype(np.float32)
    Y = rrpn.apply_regr_np(X.copy(), T)
    sc_in = np.concatenate([rs.randint(0, 60, (64, 4)).astype(np.float64), rs.standard_normal((64, 4)) * 0.7], axis=1)
    sc_in[:, 2:4] += 1
    sc_in[5, 6] = 800.0   # OverflowError path -> returns inputs
    sc_out = np.array([rrpn.apply_regr(*[float(v) for v in row]) for row in sc_in], dtype=np.float64)
    save("apply_regr", X=X, T=T, Y=Y, scalar_in=sc_in, scalar_out=sc_out)

    # ---- non_max_suppression_fast ---------------------------------------------------
    cases = {}
    for ci, (n, thr, mb, seed, span) in enumerate([(500, 0.7, 300, 21, 60), (2000, 0.7, 300, 22, 62), (300, 0.2, 300, 23, 900),
                                                  (120, 0.4, 300, 24, 2000), (1, 0.7, 300, 25, 60), (4000, 0.9, 50, 26, 62)]):
        rs = np.random.RandomState(seed)
        x1 = rs.randint(0, span - 2, n); y1 = rs.randint(0, span - 2, n)
        w = rs.randint(1, span // 2, n); h = rs.randint(1, span // 2, n)
        boxes = np.stack([x1, y1, np.minimum(x1 + w, span), np.minimum(y1 + h, span)], 1).astype(np.float64)
        probs = rs.permutation(n).astype(np.float32) / np.float32(n) * np.float32(0.999) + np.float32(0.0005)  # tie-free
        ob, op = rrpn.non_max_suppression_fast(boxes.copy(), probs.copy(), overlap_thresh=thr, max_boxes=mb)
        cases.update({f"c{ci}_boxes": boxes, f"c{ci}_probs": probs, f"c{ci}_thr": np.float64(thr), f"c{ci}_max": np.int64(mb),
                      f"c{ci}_out_boxes": ob, f"c{ci}_out_probs": op})
    # integer dtype input (as final per-class NMS gets python ints)
    rs = np.random.RandomState(27)
    x1 = rs.randint(0, 900, 80); y1 = rs.randint(0, 900, 80)
    boxes = np.stack([x1, y1, x1 + rs.randint(16, 300, 80), y1 + rs.randint(16, 300, 80)], 1)
    probs = (rs.permutation(80) / 80.0 * 0.3 + 0.7)
    ob, op = rrpn.non_max_suppression_fast(boxes.copy(), probs.copy(), overlap_thresh=0.2)
    cases.update({"c6_boxes": boxes, "c6_probs": probs, "c6_thr": np.float64(0.2), "c6_max": np.int64(300),
                  "c6_out_boxes": ob, "c6_out_probs": op})
    cases["n_cases"] = np.int64(7)
    save("nms", **cases)

    # ---- rpn_to_roi -------------------------------------------------------------------
    cases = {}
    specs = [  # rows, cols, scales, seed, regr_sigma, thr, max_boxes
        (38, 63, [64, 128, 256, 512], 31, 0.5, 0.7, 300),
        (38, 50, [64, 128, 256, 512], 32, 0.3, 0.7, 300),
        (38, 38, [64, 128, 256, 512], 33, 1.0, 0.7, 300),
        (37, 62, [128, 256, 512], 34, 0.5, 0.7, 300),
        (10, 12, [64, 128, 256, 512], 35, 0.5, 0.9, 300),
        (63, 63, [64, 128, 256, 512], 36, 0.2, 0.7, 300),
        (38, 63, [64, 128, 256, 512], 37, 3.0, 0.7, 300),   # wild regressions: many degenerate boxes
    ]
    for ci, (rows, cols, scales, seed, sig, thr, mb) in enumerate(specs):
        Cx = rconfig.Config()
        Cx.anchor_box_scales = scales
        A = len(scales) * len(Cx.anchor_box_ratios)
        rs = np.random.RandomState(seed)
        n = rows * cols * A
        cls = (rs.permutation(n).astype(np.float32) / np.float32(n)).reshape(1, rows, cols, A)  # tie-free
        regr = (rs.standard_normal((1, rows, cols, 4 * A)) * sig * Cx.std_scaling).astype(np.float32)
        R = rrpn.rpn_to_roi(cls.copy(), regr.copy(), Cx, use_regr=True, max_boxes=mb, overlap_thresh=thr)
        cases.update({f"c{ci}_cls": cls, f"c{ci}_regr": regr, f"c{ci}_scales": np.array(scales, dtype=np.int64),
                      f"c{ci}_thr": np.float64(thr), f"c{ci}_max": np.int64(mb), f"c{ci}_R": R})
    cases["n_cases"] = np.int64(len(specs))
    save("rpn_to_roi", **cases)

    # ---- calc_iou (RoI labelling) ---------------------------------------------------------
    cases = {}
    for ci, (seed, ngt, W, H) in enumerate([(41, 8, 2000, 1200), (42, 40, 2000, 1200), (43, 3, 1200, 2000), (44, 1, 2048, 2048)]):
        rs = np.random.RandomState(seed)
        bboxes = synth_gt(rs, ngt, W, H, fg)
        img_data = {"bboxes": bboxes, "width": W, "height": H}
        rw, rh = rutils.get_new_img_size(W, H, C.img_size)
        fw, fh = feat_size(rw, rh)
        n = 300
        x1 = rs.randint(0, fw - 2, n); y1 = rs.randint(0, fh - 2, n)
        R = np.stack([x1, y1, np.minimum(x1 + rs.randint(1, 30, n), fw - 1), np.minimum(y1 + rs.randint(1, 30, n), fh - 1)], 1).astype(np.int64)
        # plant RoIs that match GT well so the positive branch is exercised
        for k, bb in enumerate(bboxes[: min(ngt, 40)]):
            sx, sy = rw / W / 16.0, rh / H / 16.0
            R[k] = [int(round(bb["x1"] * sx)), int(round(bb["y1"] * sy)), int(round(bb["x2"] * sx)), int(round(bb["y2"] * sy))]
            if k + 60 < n:
                R[k + 60] = R[k] + np.array([1, 0, 1, 1])
        X2, Y1, Y2, ious = rrpn.calc_iou(R, img_data, C, C.class_mapping)
        gb, gc = gt_to_arrays(bboxes, C.class_mapping)
        cases.update({f"c{ci}_R": R, f"c{ci}_gt_boxes": gb, f"c{ci}_gt_cls": gc, f"c{ci}_wh": np.array([W, H], dtype=np.int64),
                      f"c{ci}_X": X2, f"c{ci}_Y1": Y1, f"c{ci}_Y2": Y2, f"c{ci}_ious": np.array(ious)})
    # nothing kept -> 4x None
    img_data = {"bboxes": [{"class": "boat", "x1": 1900, "x2": 1990, "y1": 1100, "y2": 1190}], "width": 2000, "height": 1200}
    R = np.array([[0, 0, 2, 2], [1, 1, 3, 3]], dtype=np.int64)
    res = rrpn.calc_iou(R, img_data, C, C.class_mapping)
    assert res[0] is None
    cases["n_cases"] = np.int64(4)
    save("calc_iou", **cases)

    # ---- calc_region_props (anchor targets; "calc_rpn") --------------------------------------
    cases = {}
    specs = [  # seed, ngt, W, H, img_size, smin, smax, rng_seed
        (51, 2, 2000, 1200, 600, 64, 400, 64),
        (52, 8, 2000, 1200, 600, 64, 400, 64),
        (53, 15, 2000, 1200, 600, 100, 700, 7),
        (54, 40, 2000, 1200, 600, 64, 400, 64),
        (55, 3, 1200, 2000, 600, 64, 500, 1),
        (56, 5, 2048, 2048, 600, 200, 900, 2),
        (57, 0, 2000, 1200, 600, 64, 400, 64),      # no GT: nothing written
        (58, 60, 2000, 1200, 600, 200, 420, 3),      # many positives (n_pos > 128 path if reachable)
        (59, 6, 800, 600, 600, 30, 200, 5),          # 600x800 (cfg 1 size)
    ]
    for ci, (seed, ngt, W, H, isz, smin, smax, rseed) in enumerate(specs):
        rs = np.random.RandomState(seed)
        bboxes = synth_gt(rs, ngt, W, H, fg, smin, smax)
        if ci == 3:
            bboxes[5]["class"] = "bg"   # 'bg'-class GT is skipped by the labeller
        img_data = {"bboxes": bboxes, "width": W, "height": H}
        Cx = rconfig.Config(); Cx.img_size = isz
        rw, rh = rutils.get_new_img_size(W, H, isz)
        np.random.seed(rseed)
        try:
            ycls, yregr, best_anchor, n_pos = rutils.calc_region_props(Cx, img_data, W, H, rw, rh, feat_size)
            raised = 0
        except KeyError:
            raised = 1
            fw, fh = feat_size(rw, rh)
            ycls = np.zeros((1, 24, fh, fw)); yregr = np.zeros((1, 96, fh, fw)); best_anchor = np.zeros((ngt, 4), dtype=int); n_pos = -1
        rng_after = np.random.randint(0, 2 ** 31 - 1)   # pins how much of the global stream was consumed
        gb = np.array([[b["x1"], b["y1"], b["x2"], b["y2"]] for b in bboxes], dtype=np.float64).reshape(-1, 4)
        gbg = np.array([1 if b["class"] == "bg" else 0 for b in bboxes], dtype=np.int64)
        cases.update({f"c{ci}_gt_boxes": gb, f"c{ci}_gt_is_bg": gbg, f"c{ci}_wh": np.array([W, H, rw, rh, isz, rseed], dtype=np.int64),
                      f"c{ci}_y_rpn_cls": ycls.astype(np.float64), f"c{ci}_y_rpn_regr": yregr.astype(np.float64),
                      f"c{ci}_best_anchor": np.asarray(best_anchor).astype(np.int64), f"c{ci}_n_pos": np.int64(n_pos),
                      f"c{ci}_raised": np.int64(raised), f"c{ci}_rng_after": np.int64(rng_after)})
    cases["n_cases"] = np.int64(len(specs))
    save("calc_region_props", **cases)

    # ---- RADNet: apply_spatial_pyramid_pooling, final_nms, get_real_coordinates, predict ---------
    class FakeDetector:
        """Deterministic stand-in for model_detector: outputs are a pure function of the RoIs."""
        def __init__(self, nc, seed):
            self.nc = nc; self.seed = seed; self.calls = []
        def predict(self, inputs):
            F, rois = inputs
            self.calls.append(np.array(rois))
            r = np.asarray(rois)[0].astype(np.float64)
            key = (r * np.array([3.0, 5.0, 7.0, 11.0])).sum(1) + self.seed
            logits = np.stack([np.sin(key * (k + 1) * 0.37) * 9.0 for k in range(self.nc)], 1)
            e = np.exp(logits - logits.max(1, keepdims=True)); p = (e / e.sum(1, keepdims=True)).astype(np.float32)
            regr = np.stack([np.cos(key * (k + 1) * 0.11) * 2.0 for k in range(4 * (self.nc - 1))], 1).astype(np.float32)
            return [p[None], regr[None]]

    class FakeRPN:
        def __init__(self, A, seed):
            self.A = A; self.seed = seed
        def predict(self, X):
            h = resnet_out_len(X.shape[1]); w = resnet_out_len(X.shape[2])
            rs = np.random.RandomState(self.seed + int(abs(float(X.sum()))) % 1000)
            n = h * w * self.A
            cls = (rs.permutation(n).astype(np.float32) / np.float32(n)).reshape(1, h, w, self.A)
            regr = (rs.standard_normal((1, h, w, 4 * self.A)) * 2.0).astype(np.float32)
            F = rs.standard_normal((1, h, w, 8)).astype(np.float32)
            return [cls, regr, F]

    nc = len(C.class_mapping)
    det = FakeDetector(nc, 5)
    net = rradnet.RADNet(C, FakeRPN(12, 3), det, lambda x: x)
    rs = np.random.RandomState(61)
    n = 47   # not a multiple of n_rois -> last chunk is padded
    R = np.stack([rs.randint(0, 50, n), rs.randint(0, 30, n), rs.randint(1, 12, n), rs.randint(1, 8, n)], 1).astype(np.int64)
    bb, pp = net.apply_spatial_pyramid_pooling(R, np.zeros((1, 38, 63, 8), np.float32))
    spp = {"R": R, "classes": np.array(sorted(bb.keys()))}
    for k in bb:
        spp[f"boxes_{k}"] = np.array(bb[k], dtype=np.int64); spp[f"probs_{k}"] = np.array(pp[k], dtype=np.float64)
    spp["detector_calls"] = np.stack(det.calls)
    # exact multiple of n_rois (the reference then makes one more empty iteration and breaks)
    det2 = FakeDetector(nc, 9); net2 = rradnet.RADNet(C, None, det2, None)
    R40 = R[:40]
    bb2, pp2 = net2.apply_spatial_pyramid_pooling(R40, None)
    spp["n_calls_40"] = np.int64(len(det2.calls))
    for k in bb2:
        spp[f"m40_boxes_{k}"] = np.array(bb2[k], dtype=np.int64); spp[f"m40_probs_{k}"] = np.array(pp2[k], dtype=np.float64)
    spp["m40_classes"] = np.array(sorted(bb2.keys()))
    save("spp", **spp)

    fn = {}
    for ci, (seed, n) in enumerate([(71, 60), (72, 7), (73, 200)]):
        rs = np.random.RandomState(seed)
        x1 = rs.randint(0, 1500, n); y1 = rs.randint(0, 1500, n)
        boxes = np.stack([x1, y1, x1 + rs.randint(20, 400, n), y1 + rs.randint(20, 400, n)], 1)
        probs = 0.7 + 0.3 * rs.permutation(n) / n
        nb, npb = net.final_nms(boxes.copy(), probs.copy(), obj_avg_threshold=0.2, obj_confidence_threshold=0.8, n_obj_avg=5)
        fn.update({f"c{ci}_boxes": boxes, f"c{ci}_probs": probs, f"c{ci}_out_boxes": nb, f"c{ci}_out_probs": npb})
    fn["n_cases"] = np.int64(3)
    ratios = [0.3, 0.6, 1.0, 600 / 2048.0, 0.2929]
    coords = np.random.RandomState(74).randint(0, 1000, (20, 4))
    fn["grc_in"] = coords; fn["grc_ratios"] = np.array(ratios)
    fn["grc_out"] = np.array([[net.get_real_coordinates(r, *[int(v) for v in c]) for c in coords] for r in ratios])
    save("final_nms", **fn)

    # whole predict() with fake models; tile already at the network size so cv2.resize == identity
    Cp = rconfig.Config(); Cp.tile_size = 600; Cp.tile_overlap = 300; Cp.img_size = 600
    detp = FakeDetector(nc, 2)
    netp = rradnet.RADNet(Cp, FakeRPN(12, 8), detp, lambda x: x - np.float32(100.0))
    img = np.random.RandomState(81).randint(0, 256, (600, 900, 3)).astype(np.uint8)
    import io, contextlib
    with contextlib.redirect_stderr(io.StringIO()):
        dets = netp.predict([img])
    pd_ = {"img": img, "n": np.int64(len(dets)),
           "classes": np.array([d["class"] for d in dets]),
           "probs": np.array([d["prob"] for d in dets], dtype=np.float64),
           "boxes": np.array([[d["x1"], d["y1"], d["x2"], d["y2"]] for d in dets], dtype=np.int64).reshape(-1, 4)}
    # the same panel with the full-image pass on top of the tiles (C.include_full_img, RADNet.py:606-665): the whole image
    # at the network size is one more source of detections, without a tile offset
    class FakeDetectorTieFree(FakeDetector):
        """As FakeDetector, with irrational RoI weights and a softer softmax: distinct RoIs get distinct probabilities, so the
        result does not hang on np.argsort's order among EQUAL scores (SURVEY.md A.4 rule 6) -- with two panels and the
        cross-image NMS that order changes which boxes survive."""
        def predict(self, inputs):
            F, rois = inputs
            r = np.asarray(rois)[0].astype(np.float64)
            # + a term from the feature map: the same RoI proposed on two panels must not score the same on both
            key = (r * np.array([3.0 ** 0.5, 5.0 ** 0.5, 7.0 ** 0.5, 11.0 ** 0.5])).sum(1) + self.seed + float(np.abs(np.asarray(F, dtype=np.float64)).sum()) % 7.0
            logits = np.stack([np.sin(key * (k + 1) * 0.37) * 4.0 for k in range(self.nc)], 1)
            e = np.exp(logits - logits.max(1, keepdims=True)); p = (e / e.sum(1, keepdims=True)).astype(np.float32)
            regr = np.stack([np.cos(key * (k + 1) * 0.11) * 2.0 for k in range(4 * (self.nc - 1))], 1).astype(np.float32)
            return [p[None], regr[None]]

    def run_predict(Cq, rpn_seed, det_seed, images, tie_free=False):
        netq = rradnet.RADNet(Cq, FakeRPN(12, rpn_seed), (FakeDetectorTieFree if tie_free else FakeDetector)(nc, det_seed), lambda x: x - np.float32(100.0))
        with contextlib.redirect_stderr(io.StringIO()):
            d = netq.predict(images)
        return {"n": np.int64(len(d)), "classes": np.array([q["class"] for q in d]),
                "probs": np.array([q["prob"] for q in d], dtype=np.float64),
                "boxes": np.array([[q["x1"], q["y1"], q["x2"], q["y2"]] for q in d], dtype=np.int64).reshape(-1, 4)}
    Cf = rconfig.Config(); Cf.tile_size = 600; Cf.tile_overlap = 300; Cf.img_size = 600; Cf.include_full_img = True
    for k, v in run_predict(Cf, 8, 2, [img]).items():
        pd_["full_" + k] = v
    # full image only (no tiling: max_n_tiles_train = 0), two panels of different sizes, scaled 620x400 -> 465x300 and
    # 350x500 -> 300x428 by the stubbed cv2.resize (oracle/resize.py): format_img's ratio and get_real_coordinates at work
    Co = rconfig.Config(); Co.img_size = 300; Co.include_full_img = True; Co.max_n_tiles_train = 0
    rs2 = np.random.RandomState(82)
    img_a = rs2.randint(0, 256, (400, 620, 3)).astype(np.uint8)
    img_b = rs2.randint(0, 256, (500, 350, 3)).astype(np.uint8)
    pd_["only_img_a"], pd_["only_img_b"] = img_a, img_b
    for k, v in run_predict(Co, 4, 6, [img_a, img_b], tie_free=True).items():
        pd_["only_" + k] = v
    assert len(set(pd_["only_probs"].tolist())) == len(pd_["only_probs"]), "tie-free detector produced equal scores"
    save("predict_fake", **pd_)

    mpath = os.path.join(OUT, "MANIFEST.json")
    if os.path.exists(mpath):          # entries of the other generators (tools/gen_golden_feed.py, ...) stay
        with open(mpath) as f:
            prev = json.load(f)
        for k, v in prev.items():
            manifest.setdefault(k, v)
        for k, v in prev.get("files", {}).items():
            manifest["files"].setdefault(k, v)
    with open(mpath, "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("wrote", sorted(manifest["files"]))


if __name__ == "__main__":
    main()
