"""Evaluation tail (test.py:48-173, 242-264): the product's vectorised implementation against the loop
restatement in oracle/evaluate.py and hand-computed cases.  The reference's own outputs are checked in tests/test_script_goldens.py."""
import copy

import numpy as np
import pytest

from faster_rcnn import evaluate as ev
from oracle import evaluate as oev

CLASSES = ("boat", "human", "other", "animal", "circle", "wheel")


def random_case(seed, n_gt, n_pred, tie_probs=False):
    rs = np.random.RandomState(seed)
    gt, pred = [], []
    for i in range(n_gt):
        x1, y1 = int(rs.randint(0, 900)), int(rs.randint(0, 500))
        w, h = int(rs.randint(20, 200)), int(rs.randint(20, 200))
        gt.append({"class": CLASSES[rs.randint(len(CLASSES))], "x1": x1, "y1": y1, "x2": x1 + w, "y2": y1 + h})
    for i in range(n_pred):
        if gt and rs.rand() < 0.6:                        # jittered copy of a ground-truth box
            g = gt[rs.randint(len(gt))]
            j = rs.randint(-25, 26, 4)
            box = (g["x1"] + j[0], g["y1"] + j[1], g["x2"] + j[2], g["y2"] + j[3])
            cls = g["class"] if rs.rand() < 0.8 else CLASSES[rs.randint(len(CLASSES))]
        else:
            x1, y1 = int(rs.randint(0, 900)), int(rs.randint(0, 500))
            box = (x1, y1, x1 + int(rs.randint(-5, 200)), y1 + int(rs.randint(-5, 200)))      # may be degenerate
            cls = CLASSES[rs.randint(len(CLASSES))]
        prob = float(rs.choice([0.81, 0.9, 0.95])) if tie_probs else float(rs.uniform(0.8, 1.0))
        pred.append({"class": cls, "x1": int(box[0]), "y1": int(box[1]), "x2": int(box[2]), "y2": int(box[3]), "prob": prob})
    return pred, gt


@pytest.mark.parametrize("seed,n_gt,n_pred,ties", [(0, 12, 40, False), (1, 30, 25, True), (2, 0, 10, False), (3, 9, 0, False),
                                                  (4, 60, 300, True), (5, 1, 1, False)])
def test_matches_loop_restatement(seed, n_gt, n_pred, ties):
    pred, gt = random_case(seed, n_gt, n_pred, ties)
    gt_a, gt_b = copy.deepcopy(gt), copy.deepcopy(gt)
    T, P = ev.get_objects(pred, gt_a, 0.5)
    To, Po = oev.get_objects(pred, gt_b, 0.5)
    assert list(T.keys()) == list(To.keys()) and T == To and P == Po
    assert [g["bbox_matched"] for g in gt_a] == [g["bbox_matched"] for g in gt_b]
    for key in T:
        a, b = ev.calc_class_ap(T[key], P[key]), oev.calc_class_ap(To[key], Po[key])
        assert a[0] == b[0]                                   # bit-identical AP (same sequential sum)
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[3] == b[3] and a[4] == b[4]
    if T:
        acc, acco = ev.mean_average_precision(pred, copy.deepcopy(gt)), oev.mean_average_precision(pred, copy.deepcopy(gt))
        assert list(acc.keys()) == list(acco.keys()) and all(acc[k] == acco[k] for k in acc)


def test_hand_computed_curve():
    # scores descending: TP, FP, TP, then one unmatched ground-truth box (score 0)
    ap, prec, rec, ip, ir = ev.calc_class_ap([1, 0, 1, 1], [0.9, 0.8, 0.7, 0])
    assert np.allclose(prec, [1.0, 0.5, 2 / 3, 2 / 3]) and np.allclose(rec, [1 / 3, 1 / 3, 2 / 3, 2 / 3])
    assert np.allclose(ip, [1.0, 2 / 3, 2 / 3, 2 / 3])
    # the first recall step (0 -> 1/3, precision 1) is NOT counted: only 2/3 * (2/3 - 1/3)
    assert ap == pytest.approx(2 / 9)
    # a single prediction has no second curve point: AP 0 even when it is a hit (reference quirk)
    assert ev.calc_class_ap([1], [0.99])[0] == 0
    # no ground truth at all: recall identically 0
    ap, prec, rec, _, _ = ev.calc_class_ap([0, 0], [0.9, 0.8])
    assert ap == 0 and np.array_equal(rec, [0.0, 0.0]) and np.array_equal(prec, [0.0, 0.0])


def test_greedy_matching_rules():
    gt = [{"class": "boat", "x1": 0, "y1": 0, "x2": 100, "y2": 100}, {"class": "boat", "x1": 10, "y1": 0, "x2": 110, "y2": 100},
          {"class": "human", "x1": 0, "y1": 0, "x2": 100, "y2": 100}]
    pred = [{"class": "boat", "x1": 0, "y1": 0, "x2": 100, "y2": 100, "prob": 0.9},      # index 0
            {"class": "boat", "x1": 0, "y1": 0, "x2": 100, "y2": 100, "prob": 0.9},      # index 1: tie -> visited FIRST
            {"class": "boat", "x1": 0, "y1": 0, "x2": 100, "y2": 100, "prob": 0.85},     # both boats taken -> miss
            {"class": "wheel", "x1": 0, "y1": 0, "x2": 100, "y2": 100, "prob": 0.95}]    # class without ground truth
    T, P = ev.get_objects(pred, gt, 0.5)
    assert list(T.keys()) == ["wheel", "boat", "human"]
    assert T["wheel"] == [0] and P["wheel"] == [0.95]
    assert T["boat"] == [1, 1, 0] and P["boat"] == [0.9, 0.9, 0.85]     # second tie-visited box takes the shifted boat (IoU 0.82)
    assert T["human"] == [1] and P["human"] == [0]                      # unmatched ground truth: (1, score 0)
    assert [g["bbox_matched"] for g in gt] == [True, True, False]
