"""Evaluation tail of the reference's test.py: detections + ground truth -> PASCAL-VOC-style AP per class
and mAP, with the reference's conventions (test.py:48-173, 242-264):

  * detections and ground-truth boxes of ALL images are pooled, without image identity, before matching (test.py:225-226);
  * predictions are visited in descending probability, ties resolving to the later index (argsort()[::-1]); each takes the
    first still-unmatched ground-truth box of its class with IoU >= threshold;
  * unmatched ground truth enters the curve as (true = 1, score = 0) and moves neither tp nor fp;
  * AP sums interpolated precision over recall steps starting at the SECOND curve point (the first step 0 -> r[0] is not
    counted).

KB-scale host work after `RADNet.predict`; vectorised NumPy (IoU matrices per class, cumulative sums), no GPU kernel.
Parity unpinned against the reference itself (test.py needs cv2 / TensorFlow to import); tests compare with the loop
restatement in oracle/evaluate.py and hand-computed cases.
"""
import numpy as np

GT_IOU_THRESHOLD = 0.5       # test.py:44


def _iou_matrix(p, g):
    """IoU of every prediction box p[i] with every ground-truth box g[j] (utils.py:77-109 semantics: degenerate box -> 0,
    disjoint -> 0, inter / (union + 1e-6))."""
    p = np.asarray(p, dtype=np.float64).reshape(-1, 1, 4)
    g = np.asarray(g, dtype=np.float64).reshape(1, -1, 4)
    bad = (p[..., 0] >= p[..., 2]) | (p[..., 1] >= p[..., 3]) | (g[..., 0] >= g[..., 2]) | (g[..., 1] >= g[..., 3])
    w = np.minimum(p[..., 2], g[..., 2]) - np.maximum(p[..., 0], g[..., 0])
    h = np.minimum(p[..., 3], g[..., 3]) - np.maximum(p[..., 1], g[..., 1])
    inter = np.where((w < 0) | (h < 0), 0.0, w * h)
    union = (p[..., 2] - p[..., 0]) * (p[..., 3] - p[..., 1]) + (g[..., 2] - g[..., 0]) * (g[..., 3] - g[..., 1]) - inter
    return np.where(bad, 0.0, inter / (union + 1e-6))


def get_objects(pred, gt, treshold):
    """test.py:48-115 (argument spelling as in the reference).  Returns (T, P): per class, 1/0 per prediction (+ 1 per
    unmatched ground-truth box) and the matching scores (+ 0 per unmatched box).  Marks gt[i]['bbox_matched'] like the
    reference does.  Dict insertion order = order of first appearance, as in the reference's loop."""
    T, P = {}, {}
    for g in gt:
        g["bbox_matched"] = False
    order = np.argsort(np.array([p["prob"] for p in pred]))[::-1] if len(pred) else np.zeros(0, np.int64)
    gt_cls = np.array([g["class"] for g in gt], dtype=object)
    gt_box = np.array([[g["x1"], g["y1"], g["x2"], g["y2"]] for g in gt], dtype=np.float64).reshape(-1, 4)
    matched = np.zeros(len(gt), dtype=bool)
    by_class = {}
    for k in order:                                      # predictions of a class, in visiting order
        by_class.setdefault(pred[k]["class"], []).append(int(k))
    hit_of = {}
    for c, ks in by_class.items():
        gi = np.flatnonzero(gt_cls == c)
        if len(gi) == 0:
            for k in ks:
                hit_of[k] = False
            continue
        ok = _iou_matrix([[pred[k]["x1"], pred[k]["y1"], pred[k]["x2"], pred[k]["y2"]] for k in ks], gt_box[gi]) >= treshold
        free = np.ones(len(gi), dtype=bool)
        for r, k in enumerate(ks):                       # greedy and order-dependent by definition
            cand = np.flatnonzero(ok[r] & free)
            hit_of[k] = len(cand) > 0
            if len(cand):
                free[cand[0]] = False
        matched[gi[~free]] = True
    for k in order:                                      # emit in the reference's order (class keys by first appearance)
        c = pred[k]["class"]
        P.setdefault(c, []).append(pred[k]["prob"])
        T.setdefault(c, []).append(int(hit_of[int(k)]))
    for i, g in enumerate(gt):
        g["bbox_matched"] = bool(matched[i])
        if not matched[i]:
            T.setdefault(g["class"], []).append(1)
            P.setdefault(g["class"], []).append(0)
    return T, P


def calc_class_ap(y_true, y_pred):
    """test.py:119-173 -> (ap, precision, recall, interpolated_precision, interpolated_recall)."""
    y_true, y_pred = np.array(y_true), np.array(y_pred)
    n = len(y_pred)
    if n == 0:
        return 0, np.array([]), np.array([]), [], []
    n_gt = np.sum(y_true)
    idx = np.flip(np.argsort(y_pred))
    scored = y_pred[idx] > 0.0
    tp = np.cumsum((y_true[idx] > 0) & scored)
    fp = np.cumsum((y_true[idx] == 0) & scored)
    seen = tp + fp
    precision = np.where(seen == 0, 0.0, tp / np.maximum(seen, 1))
    recall = tp / n_gt if n_gt != 0 else np.zeros(n)
    interp = np.maximum(np.maximum.accumulate(precision[::-1])[::-1], 0.0)
    # the reference adds the terms one by one from an integer 0: a sequential sum (cumsum), not a pairwise one
    terms = interp[1:] * np.diff(recall)
    ap = np.cumsum(terms)[-1] if n > 1 else 0
    return ap, precision, recall, list(interp), list(recall)


def mean_average_precision(all_dets, all_gt, treshold=GT_IOU_THRESHOLD):
    """test.py:242-264: {class: AP ..., 'mAP': plain mean over the classes that occur}, classes in sorted order."""
    T, P = get_objects(all_dets, all_gt, treshold)
    accuracy = {}
    for key in sorted(T.keys()):
        accuracy[key] = calc_class_ap(T[key], P[key])[0]
    accuracy["mAP"] = np.mean(np.array([accuracy[k] for k in sorted(T.keys())]))
    return accuracy
