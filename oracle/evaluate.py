"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's evaluation tail (test.py:48-173, 242-264), plain loops.

Parity PINNED for get_objects / calc_class_ap: tools/gen_golden_scripts.py imports the reference's test.py (empty stubs for
cv2 / tensorflow / keras / matplotlib.pyplot, none of which these functions touch) and records their outputs in
tests/golden/voc_ap.json; tests/test_script_goldens.py checks this restatement and the product (faster_rcnn/evaluate.py,
vectorised) against them.  The flip / 90-degree augmentation loops below stay unpinned (cv2.flip cannot run here).  Nothing under
rock-art-radnet_amd/ imports this file.
"""
import numpy as np

from oracle.glue import iou_pairs


def _iou(a, b):
    return float(iou_pairs(np.asarray(a, np.float64)[None], np.asarray(b, np.float64)[None])[0])      # utils.py:77-109


def get_objects(pred, gt, threshold):
    """test.py:48-115.  pred: [{class,x1,y1,x2,y2,prob}], gt: [{class,x1,y1,x2,y2}] (gains 'bbox_matched').
    Predictions in descending probability (argsort()[::-1]: ties resolve to the LATER index first, test.py:57) each take
    the first still-unmatched ground-truth box of their class with IoU >= threshold (test.py:77-101); T gets 1/0 per
    prediction, P its probability; every ground-truth box left unmatched adds (T=1, P=0) (test.py:105-113).
    Detections and boxes are pooled over all images by the caller -- there is no image identity (test.py:225-226)."""
    T, P = {}, {}
    for g in gt:
        g["bbox_matched"] = False
    probs = np.array([p["prob"] for p in pred])
    for k in np.argsort(probs)[::-1]:
        p = pred[k]
        c = p["class"]
        T.setdefault(c, [])
        P.setdefault(c, []).append(p["prob"])
        hit = False
        for g in gt:
            if g["class"] != c or g["bbox_matched"]:
                continue
            if _iou((p["x1"], p["y1"], p["x2"], p["y2"]), (g["x1"], g["y1"], g["x2"], g["y2"])) >= threshold:
                g["bbox_matched"] = True
                hit = True
                break
        T[c].append(int(hit))
    for g in gt:
        if not g["bbox_matched"]:
            T.setdefault(g["class"], []).append(1)
            P.setdefault(g["class"], []).append(0)
    return T, P


def calc_class_ap(y_true, y_pred):
    """test.py:119-173.  Walk the scores downwards (flip(argsort): ties -> later index first); entries with score 0 (the
    unmatched ground truth) move neither tp nor fp; precision 0 while tp+fp == 0; recall tp / sum(y_true) (0 if that is 0);
    interpolated precision = running maximum from the right; AP = sum_{i>=0} ip[i+1] * (r[i+1] - r[i]) -- the first
    recall step (0 -> r[0]) is NOT counted (test.py:168-170)."""
    y_true, y_pred = np.array(y_true), np.array(y_pred)
    n_gt = np.sum(y_true)
    tp = fp = 0
    prec, rec = [], []
    for i in np.flip(np.argsort(y_pred)):
        if y_true[i] > 0 and y_pred[i] > 0.0:
            tp += 1
        elif y_true[i] == 0 and y_pred[i] > 0.0:
            fp += 1
        prec.append(0.0 if tp + fp == 0 else tp / (tp + fp))
        rec.append(tp / n_gt if n_gt != 0 else 0.0)
    best = 0.0
    ip = [0.0] * len(prec)
    for i in range(len(prec) - 1, -1, -1):
        best = max(best, prec[i])
        ip[i] = best
    ap = 0
    for i in range(len(ip) - 1):
        ap += ip[i + 1] * (rec[i + 1] - rec[i])
    return ap, np.array(prec), np.array(rec), ip, list(rec)


def mean_average_precision(all_dets, all_gt, threshold=0.5):
    """test.py:242-264: per-class AP over sorted class names, 'mAP' = their plain mean."""
    T, P = get_objects(all_dets, all_gt, threshold)
    acc = {}
    for key in sorted(T.keys()):
        acc[key] = calc_class_ap(T[key], P[key])[0]
    acc["mAP"] = np.mean(np.array([acc[k] for k in sorted(T.keys())]))
    return acc


def augment_geometric_loops(boxes, img, flips, angle):
    """augmentation.py:85-159 restated with explicit pixel loops (test infrastructure): flips = (horizontal, vertical)
    booleans, angle in (None, 90, 180, 270), applied in the reference's order.  Returns (boxes, img)."""
    boxes = [dict(b) for b in boxes]
    if flips[0]:
        rows, cols = img.shape[:2]
        out = np.empty_like(img)
        for x in range(cols):
            out[:, x] = img[:, cols - 1 - x]                    # cv2.flip(img, 1)
        img = out
        for b in boxes:
            x1, x2 = b["x1"], b["x2"]
            b["x2"], b["x1"] = cols - x1, cols - x2
    if flips[1]:
        rows, cols = img.shape[:2]
        out = np.empty_like(img)
        for y in range(rows):
            out[y] = img[rows - 1 - y]                          # cv2.flip(img, 0)
        img = out
        for b in boxes:
            y1, y2 = b["y1"], b["y2"]
            b["y2"], b["y1"] = rows - y1, rows - y2
    if angle is not None:
        rows, cols = img.shape[:2]
        if angle == 180:
            out = np.empty_like(img)
            for y in range(rows):
                for x in range(cols):
                    out[y, x] = img[rows - 1 - y, cols - 1 - x]  # cv2.flip(img, -1)
        else:
            out = np.empty((cols, rows) + img.shape[2:], img.dtype)
            for y in range(cols):
                for x in range(rows):
                    # transpose, then flip rows (270) or columns (90)
                    out[y, x] = img[x, cols - 1 - y] if angle == 270 else img[rows - 1 - x, y]
        img = out
        for b in boxes:
            x1, x2, y1, y2 = b["x1"], b["x2"], b["y1"], b["y2"]
            if angle == 270:
                b["x1"], b["x2"], b["y1"], b["y2"] = y1, y2, cols - x2, cols - x1
            elif angle == 180:
                b["x2"], b["x1"], b["y2"], b["y1"] = cols - x1, cols - x2, rows - y1, rows - y2
            else:
                b["x1"], b["x2"], b["y1"], b["y2"] = rows - y2, rows - y1, x1, x2
    return boxes, img
