"""Oracle (oracle/glue.py) vs. vectors produced by the reference itself (tools/gen_golden.py).
Integer / index / fp64 outputs are compared bit-for-bit."""
import json
import os
import pickle

import numpy as np
import pytest

from conftest import GOLDEN, load_golden
from oracle import glue
from faster_rcnn.config import Config


def test_config_surface_matches_reference_dump():
    ref = json.load(open(os.path.join(GOLDEN, "config_attrs.json")))
    assert Config().__dict__ == ref
    c2 = pickle.loads(pickle.dumps(Config()))
    assert c2.__dict__ == ref and type(c2).__module__ == "faster_rcnn.config"


def test_iou_and_img_size():
    g = load_golden("iou")
    assert np.array_equal(glue.iou_pairs(g["a"], g["b"]), g["iou"])
    assert np.array_equal(glue.iou_pairs(g["ai"], g["bi"]), g["iou_int"])
    for w, h, m, rw, rh in g["new_img_size"]:
        assert glue.new_img_size(int(w), int(h), int(m)) == (rw, rh)


def test_decode():
    g = load_golden("apply_regr")
    assert np.array_equal(glue.decode_deltas_np(g["X"], g["T"]), g["Y"])
    out = np.array([glue.decode_delta_scalar(*[float(v) for v in r]) for r in g["scalar_in"]], dtype=np.float64)
    assert np.array_equal(out, g["scalar_out"])
    assert np.array_equal(out[5], g["scalar_in"][5, :4])      # overflow -> input returned


def test_nms():
    g = load_golden("nms")
    for i in range(int(g["n_cases"])):
        b, p = glue.greedy_nms(g[f"c{i}_boxes"], g[f"c{i}_probs"], float(g[f"c{i}_thr"]), int(g[f"c{i}_max"]))
        assert np.array_equal(b, g[f"c{i}_out_boxes"]), i
        assert np.array_equal(p, g[f"c{i}_out_probs"]), i
    assert glue.greedy_nms(np.zeros((0, 4)), np.zeros(0)) == []
    with pytest.raises(AssertionError):
        glue.greedy_nms(np.array([[3., 3., 3., 5.]]), np.array([0.5]))


def test_rpn_to_roi():
    g = load_golden("rpn_to_roi")
    for i in range(int(g["n_cases"])):
        C = Config()
        C.anchor_box_scales = [int(v) for v in g[f"c{i}_scales"]]
        R = glue.rpn_to_roi(g[f"c{i}_cls"], g[f"c{i}_regr"], C, True, int(g[f"c{i}_max"]), float(g[f"c{i}_thr"]))
        assert R.dtype == g[f"c{i}_R"].dtype
        assert np.array_equal(R, g[f"c{i}_R"]), i


def test_roi_targets():
    g = load_golden("calc_iou")
    C = Config()
    for i in range(int(g["n_cases"])):
        W, H = (int(v) for v in g[f"c{i}_wh"])
        X, Y1, Y2, ious = glue.roi_targets(g[f"c{i}_R"], g[f"c{i}_gt_boxes"], g[f"c{i}_gt_cls"], W, H, C)
        assert np.array_equal(X, g[f"c{i}_X"]) and X.dtype == g[f"c{i}_X"].dtype
        assert np.array_equal(Y1, g[f"c{i}_Y1"])
        assert np.array_equal(Y2, g[f"c{i}_Y2"])
        assert np.array_equal(np.array(ious), g[f"c{i}_ious"])
    res = glue.roi_targets(np.array([[0, 0, 2, 2], [1, 1, 3, 3]]), np.array([[1900., 1100., 1990., 1190.]]),
                           np.array([0]), 2000, 1200, C)
    assert res == (None, None, None, None)


def test_anchor_targets():
    g = load_golden("calc_region_props")
    fs = lambda w, h: (glue.resnet50_feat_len(w), glue.resnet50_feat_len(h))
    for i in range(int(g["n_cases"])):
        W, H, rw, rh, isz, rseed = (int(v) for v in g[f"c{i}_wh"])
        C = Config()
        C.img_size = isz
        np.random.seed(rseed)
        ycls, yregr, best, n_pos = glue.anchor_targets(C, g[f"c{i}_gt_boxes"], g[f"c{i}_gt_is_bg"], W, H, rw, rh, fs)
        assert n_pos == int(g[f"c{i}_n_pos"]), i
        assert np.array_equal(best, g[f"c{i}_best_anchor"]), i
        assert np.array_equal(ycls, g[f"c{i}_y_rpn_cls"]), i
        assert np.array_equal(yregr, g[f"c{i}_y_rpn_regr"]), i
        assert np.random.randint(0, 2 ** 31 - 1) == int(g[f"c{i}_rng_after"]), i   # same RNG consumption


def test_select_samples_contract():
    # contract checks; the reference's own outputs: tests/test_script_goldens.py
    C = Config()
    np.random.seed(5)
    Y1 = np.zeros((1, 50, 7)); Y1[0, :, -1] = 1; Y1[0, :4, -1] = 0; Y1[0, :4, 0] = 1
    sel, npos = glue.select_samples(Y1, C.n_rois)
    assert len(sel) == C.n_rois and npos == 4 and sel[:4] == [0, 1, 2, 3]
    Y1 = np.zeros((1, 8, 7)); Y1[0, :, 2] = 1           # no negatives at all
    sel, npos = glue.select_samples(Y1, C.n_rois)
    assert len(sel) == C.n_rois and npos == 8 and sorted(sel[:8]) == list(range(8))
    Y1 = np.zeros((1, 30, 7)); Y1[0, :, 2] = 1          # > n_rois positives, no negatives:
    with pytest.raises(ValueError):                     # the reference asks choice() for a negative count
        glue.select_samples(Y1, C.n_rois)
    Y1 = np.zeros((1, 12, 7)); Y1[0, :, -1] = 1; Y1[0, 0, -1] = 0   # too few negatives: with replacement
    sel, npos = glue.select_samples(Y1, C.n_rois)
    assert len(sel) == C.n_rois and npos == 1


def fake_detector(nc, seed, calls):
    """Same closed-form stand-in detector the golden generator used (pure function of the RoIs)."""
    def predict(rois):
        calls.append(np.array(rois))
        r = np.asarray(rois)[0].astype(np.float64)
        key = (r * np.array([3.0, 5.0, 7.0, 11.0])).sum(1) + seed
        logits = np.stack([np.sin(key * (k + 1) * 0.37) * 9.0 for k in range(nc)], 1)
        e = np.exp(logits - logits.max(1, keepdims=True))
        p = (e / e.sum(1, keepdims=True)).astype(np.float32)
        regr = np.stack([np.cos(key * (k + 1) * 0.11) * 2.0 for k in range(4 * (nc - 1))], 1).astype(np.float32)
        return [p[None], regr[None]]
    return predict


def fake_detector_tie_free(nc, seed, calls):
    """tools/gen_golden.py: FakeDetectorTieFree (distinct RoIs, and the same RoI on distinct feature maps -> distinct
    probabilities)."""
    def predict(rois, F):
        calls.append(np.array(rois))
        r = np.asarray(rois)[0].astype(np.float64)
        key = (r * np.array([3.0 ** 0.5, 5.0 ** 0.5, 7.0 ** 0.5, 11.0 ** 0.5])).sum(1) + seed + float(np.abs(np.asarray(F, dtype=np.float64)).sum()) % 7.0
        logits = np.stack([np.sin(key * (k + 1) * 0.37) * 4.0 for k in range(nc)], 1)
        e = np.exp(logits - logits.max(1, keepdims=True))
        p = (e / e.sum(1, keepdims=True)).astype(np.float32)
        regr = np.stack([np.cos(key * (k + 1) * 0.11) * 2.0 for k in range(4 * (nc - 1))], 1).astype(np.float32)
        return [p[None], regr[None]]
    return predict


def test_spp_decode():
    g = load_golden("spp")
    C = Config()
    calls = []
    bb, pp = glue.spp_decode(g["R"], fake_detector(7, 5, calls), C)
    assert sorted(bb) == list(g["classes"])
    assert np.array_equal(np.stack(calls), g["detector_calls"])
    for k in bb:
        assert np.array_equal(np.array(bb[k], dtype=np.int64), g[f"boxes_{k}"])
        assert np.array_equal(np.array(pp[k], dtype=np.float64), g[f"probs_{k}"])
    calls = []
    bb, pp = glue.spp_decode(g["R"][:40], fake_detector(7, 9, calls), C)
    assert len(calls) == int(g["n_calls_40"])
    assert sorted(bb) == list(g["m40_classes"])
    for k in bb:
        assert np.array_equal(np.array(bb[k], dtype=np.int64), g[f"m40_boxes_{k}"])


def test_merge_nms_and_real_coords():
    g = load_golden("final_nms")
    for i in range(int(g["n_cases"])):
        b, p = glue.merge_nms(g[f"c{i}_boxes"], g[f"c{i}_probs"])
        assert np.array_equal(b, g[f"c{i}_out_boxes"]) and np.array_equal(p, g[f"c{i}_out_probs"])
    out = np.array([[glue.real_coords(r, *[int(v) for v in c]) for c in g["grc_in"]] for r in g["grc_ratios"]])
    assert np.array_equal(out, g["grc_out"])


def test_resize_oracle_vectorised_equals_pixel_loops():
    """oracle/resize.py: the two-pass vectorised restatement of the 8-bit INTER_CUBIC definition == the plain integer loops
    (parity vs cv2 itself unpinned: OpenCV absent); identity size reproduces the input; constant images stay constant."""
    from oracle import resize as R
    rs = np.random.RandomState(5)
    for (h, w, nh, nw, c) in [(7, 9, 5, 4, 3), (3, 2, 7, 9, 3), (1, 1, 4, 4, 1), (5, 5, 5, 5, 3), (2, 3, 1, 1, 4), (12, 17, 30, 23, 3)]:
        img = rs.randint(0, 256, (h, w, c)).astype(np.uint8)
        a = R.resize_bicubic_u8(img, nw, nh)
        assert np.array_equal(a, R.resize_bicubic_u8_loops(img, nw, nh))
        if (h, w) == (nh, nw):
            assert np.array_equal(a, img)
    flat = np.full((9, 11, 3), 143, np.uint8)
    assert (R.resize_bicubic_u8(flat, 31, 17) == 143).all()
    idx, wi = R._axis_tables(2048, 600)
    assert (wi.sum(1) >= 2047).all() and (wi.sum(1) <= 2049).all() and idx.min() == 0 and idx.max() == 2047
