"""ORACLE (test infrastructure, not product): NumPy restatement of the reference's
host-side detection glue.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the shipped path is the HIP library.

Parity status: PINNED -- every function here is checked bit-for-bit against vectors
produced by importing the reference itself (tools/gen_golden.py -> tests/golden/*.npz),
including `select_samples` (train.py:93-129): train.py is imported with empty stubs for its absent
third-party imports by tools/gen_golden_scripts.py -> tests/golden/selected_samples.json.

All citations are relative to the reference tree (faster_rcnn/...).
The arithmetic is IEEE double unless stated; where the reference mixes fp32 in
(numpy-2 NEP-50 comparisons, fp32 score arrays) the same mixing is reproduced.
"""
import math

import numpy as np


# ----------------------------------------------------------------------------------
# boxes
# ----------------------------------------------------------------------------------
def iou_pairs(a, b):
    """Element-wise IoU of boxes a[i] and b[i], each (x1,y1,x2,y2).

    utils.py:77-109: degenerate box -> 0; negative overlap extent -> 0;
    inter / (area_a + area_b - inter + 1e-6)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    ax1, ay1, ax2, ay2 = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bx1, by1, bx2, by2 = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    bad = (ax1 >= ax2) | (ay1 >= ay2) | (bx1 >= bx2) | (by1 >= by2)
    w = np.minimum(ax2, bx2) - np.maximum(ax1, bx1)
    h = np.minimum(ay2, by2) - np.maximum(ay1, by1)
    inter = np.where((w < 0) | (h < 0), 0.0, w * h)
    union = (ax2 - ax1) * (ay2 - ay1) + (bx2 - bx1) * (by2 - by1) - inter
    out = inter / (union + 1e-6)
    return np.where(bad, 0.0, out)


def new_img_size(width, height, min_side=300):
    """utils.py:65-75: short side -> min_side, long side truncated by int()."""
    if width <= height:
        f = float(min_side) / width
        return min_side, int(f * height)
    f = float(min_side) / height
    return int(f * width), min_side


def resnet50_feat_len(n):
    """base_models/resnet50.py:19-35 (pad 6, then k=7,3,1,1 at stride 2)."""
    n = n + 6
    for k in (7, 3, 1, 1):
        n = (n - k + 2) // 2
    return n


def vgg16_feat_len(n):
    """base_models/vgg16.py:18-23."""
    return n // 16


# ----------------------------------------------------------------------------------
# proposal decode (G1, G2) and greedy NMS (G3)
# ----------------------------------------------------------------------------------
def decode_deltas_np(X, T):
    """rpn.py:299-344.  X (4,H,W) f64 = x,y,w,h ; T (4,H,W) f32 = tx,ty,tw,th.
    exp() is taken on the f64 cast of the f32 delta; results rounded half-to-even."""
    x, y, w, h = X[0], X[1], X[2], X[3]
    cx = x + w / 2.0
    cy = y + h / 2.0
    cx1 = T[0] * w + cx
    cy1 = T[1] * h + cy
    w1 = np.exp(T[2].astype(np.float64)) * w
    h1 = np.exp(T[3].astype(np.float64)) * h
    return np.stack([np.round(cx1 - w1 / 2.0), np.round(cy1 - h1 / 2.0), np.round(w1), np.round(h1)])


def decode_delta_scalar(x, y, w, h, tx, ty, tw, th):
    """rpn.py:346-378: scalar decode with math.exp and Python round(); falls back to
    the input box on ValueError / OverflowError."""
    try:
        cx1 = tx * w + (x + w / 2.0)
        cy1 = ty * h + (y + h / 2.0)
        w1 = math.exp(tw) * w
        h1 = math.exp(th) * h
        return int(round(cx1 - w1 / 2.0)), int(round(cy1 - h1 / 2.0)), int(round(w1)), int(round(h1))
    except (ValueError, OverflowError):
        return x, y, w, h


def greedy_nms(boxes, probs, overlap_thresh=0.9, max_boxes=300):
    """rpn.py:380-455.  Returns (boxes[pick] as int64, probs[pick]) or [] for no boxes.

    Order rule: ascending sort of probs, walk from the end.  Among *equal* scores the
    reference's order is implementation-defined (default np.argsort); this restatement
    fixes "stable ascending, take from the end" = among equals the higher index first.
    Suppression test: inter / (area_i + area_j - inter + 1e-6) > thresh, areas without +1."""
    boxes = np.asarray(boxes)
    if len(boxes) == 0:
        return []
    if not (np.all(boxes[:, 0] < boxes[:, 2]) and np.all(boxes[:, 1] < boxes[:, 3])):
        raise AssertionError("nms: malformed box (x1>=x2 or y1>=y2)")   # rpn.py:400-401
    b = boxes.astype(np.float64)
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    area = (x2 - x1) * (y2 - y1)
    order = np.argsort(probs, kind="stable")[::-1]
    alive = np.ones(len(order), dtype=bool)      # indexed by rank in `order`
    pick = []
    for r in range(len(order)):
        if not alive[r]:
            continue
        i = order[r]
        pick.append(i)
        if len(pick) >= max_boxes:
            break
        rest = order[r + 1:]
        ww = np.maximum(0, np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]))
        hh = np.maximum(0, np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]))
        inter = ww * hh
        ov = inter / (area[i] + area[rest] - inter + 1e-6)
        alive[r + 1:] &= ~(ov > overlap_thresh)
    pick = np.asarray(pick, dtype=np.int64)
    return b[pick].astype("int"), np.asarray(probs)[pick]


def anchor_shapes(C):
    """Anchor (w,h) in feature-map units, anchor index a = ratio + n_ratios*size
    (rpn.py:108-117)."""
    out = []
    for s in C.anchor_box_scales:
        for r in C.anchor_box_ratios:
            out.append(((s * r[0]) / C.rpn_stride, (s * r[1]) / C.rpn_stride))
    return out


def decode_all_anchors(cls, regr, C, use_regr=True):
    """rpn.py:91-166 up to (and including) the removal of degenerate boxes.
    Returns (boxes (n,4) f64 xyxy in fmap units, probs (n,) f32, keep_index (n,) into the
    anchor-major flat order a*H*W + row*W + col)."""
    assert cls.shape[0] == 1
    regr = regr / C.std_scaling                      # stays f32 (rpn.py:91)
    rows, cols = cls.shape[1:3]
    shapes = anchor_shapes(C)
    A = len(shapes)
    gx, gy = np.meshgrid(np.arange(cols), np.arange(rows))
    out = np.zeros((A, 4, rows, cols))
    for a, (aw, ah) in enumerate(shapes):
        box = np.stack([gx - aw / 2, gy - ah / 2, np.full((rows, cols), aw), np.full((rows, cols), ah)]).astype(np.float64)
        if use_regr:
            box = decode_deltas_np(box, np.transpose(regr[0, :, :, 4 * a:4 * a + 4], (2, 0, 1)))
        bw = np.maximum(1, box[2])
        bh = np.maximum(1, box[3])
        x2 = bw + box[0]
        y2 = bh + box[1]
        out[a, 0] = np.maximum(0, box[0])
        out[a, 1] = np.maximum(0, box[1])
        out[a, 2] = np.minimum(cols - 1, x2)
        out[a, 3] = np.minimum(rows - 1, y2)
    boxes = out.transpose(0, 2, 3, 1).reshape(-1, 4)
    probs = cls.transpose(0, 3, 1, 2).reshape(-1)
    keep = np.nonzero(~((boxes[:, 0] - boxes[:, 2] >= 0) | (boxes[:, 1] - boxes[:, 3] >= 0)))[0]
    return boxes[keep], probs[keep], keep


def rpn_to_roi(cls, regr, C, use_regr=True, max_boxes=300, overlap_thresh=0.9):
    """rpn.py:68-172: decode every anchor, clip, drop degenerate, greedy NMS; boxes only."""
    boxes, probs, _ = decode_all_anchors(cls, regr, C, use_regr)
    res = greedy_nms(boxes, probs, overlap_thresh=overlap_thresh, max_boxes=max_boxes)
    if len(res) == 0:
        raise ValueError("rpn_to_roi: no valid proposal (reference fails unpacking [] at rpn.py:170)")
    return res[0]


# ----------------------------------------------------------------------------------
# RoI labelling (G6) and sampling (G7)
# ----------------------------------------------------------------------------------
def roi_targets(R, gt_boxes, gt_cls, width, height, C):
    """rpn.py:176-296.  R (n,4) xyxy fmap units; gt_boxes (g,4) [x1,y1,x2,y2] in source-image
    pixels; gt_cls (g,) class indices.  Returns (X (1,m,4) xywh, Y1 (1,m,nc), Y2 (1,m,8(nc-1)),
    ious list) or (None,)*4 if no RoI reaches classifier_min_overlap."""
    nc = len(C.class_mapping)
    bg = C.class_mapping["bg"]
    rw, rh = new_img_size(width, height, C.img_size)
    g = np.zeros((len(gt_boxes), 4))       # columns x1, x2, y1, y2 as in the reference
    for k, bb in enumerate(gt_boxes):
        g[k, 0] = int(round(bb[0] * (rw / float(width)) / C.rpn_stride))
        g[k, 1] = int(round(bb[2] * (rw / float(width)) / C.rpn_stride))
        g[k, 2] = int(round(bb[1] * (rh / float(height)) / C.rpn_stride))
        g[k, 3] = int(round(bb[3] * (rh / float(height)) / C.rpn_stride))
    xs, y1s, y2s, ious = [], [], [], []
    sx, sy, sw, sh = C.classifier_regr_std
    for row in np.asarray(R):
        x1, y1, x2, y2 = (int(round(v)) for v in row)
        best, best_k = 0.0, -1
        for k in range(len(g)):
            v = float(iou_pairs([g[k, 0], g[k, 2], g[k, 1], g[k, 3]], [x1, y1, x2, y2]))
            if v > best:
                best, best_k = v, k
        if best < C.classifier_min_overlap:
            continue
        w, h = x2 - x1, y2 - y1
        onehot = [0] * nc
        labels = [0] * (4 * (nc - 1))
        coords = [0] * (4 * (nc - 1))
        if best < C.classifier_max_overlap:
            onehot[bg] = 1
        else:
            c = int(gt_cls[best_k])
            onehot[c] = 1
            tx = ((g[best_k, 0] + g[best_k, 1]) / 2.0 - (x1 + w / 2.0)) / float(w)
            ty = ((g[best_k, 2] + g[best_k, 3]) / 2.0 - (y1 + h / 2.0)) / float(h)
            tw = np.log((g[best_k, 1] - g[best_k, 0]) / float(w))
            th = np.log((g[best_k, 3] - g[best_k, 2]) / float(h))
            if c != bg:
                coords[4 * c:4 * c + 4] = [sx * tx, sy * ty, sw * tw, sh * th]
                labels[4 * c:4 * c + 4] = [1, 1, 1, 1]
        xs.append([x1, y1, w, h])
        y1s.append(onehot)
        y2s.append(labels + coords)
        ious.append(best)
    if not xs:
        return None, None, None, None
    return np.array(xs)[None], np.array(y1s)[None], np.array(y2s)[None], ious


def select_samples(Y1, n_rois):
    """train.py:93-129 (pinned: tests/golden/selected_samples.json, tests/test_script_goldens.py).
    Consumes the global NumPy RNG exactly as the reference does."""
    neg = np.where(Y1[0, :, -1] == 1)[0]
    pos = np.where(Y1[0, :, -1] == 0)[0]
    if len(pos) < n_rois // 2:
        sel_pos = pos.tolist()
    else:
        sel_pos = np.random.choice(pos, n_rois // 2, replace=False).tolist()
    if len(neg) > 0:
        need = n_rois - len(sel_pos)
        try:
            sel_neg = np.random.choice(neg, need, replace=False).tolist()
        except Exception:
            sel_neg = np.random.choice(neg, need, replace=True).tolist()
        return sel_pos + sel_neg, len(pos)
    sel_pos = np.random.choice(pos, len(pos), replace=False).tolist()
    sel_pos += np.random.choice(pos, n_rois - len(sel_pos), replace=True).tolist()
    return sel_pos, len(pos)


# ----------------------------------------------------------------------------------
# anchor targets (G4, "calc_rpn")
# ----------------------------------------------------------------------------------
def anchor_targets_dense(C, gt_boxes, gt_is_bg, width, height, rw, rh, fw, fh):
    """utils.py:585-766 (everything before the random subsampling), vectorised but with the
    reference's loop-order semantics (size -> ratio -> ix -> jy -> gt; strict '>' so the first
    maximum wins; fp32 best-IoU bookkeeping compared in fp32 as numpy-2 does).

    Returns dict(valid (fh,fw,A), overlap (fh,fw,A), regr (fh,fw,4A), best_anchor (g,4) int,
    n_for_gt (g,))."""
    ratios = C.anchor_box_ratios
    nr = len(ratios)
    A = len(C.anchor_box_scales) * nr
    ds = float(C.rpn_stride)
    ng = len(gt_boxes)
    valid = np.zeros((fh, fw, A))
    overlap = np.zeros((fh, fw, A))
    regr = np.zeros((fh, fw, 4 * A))
    best_anchor = -1 * np.ones((ng, 4), dtype=int)
    n_for_gt = np.zeros(ng, dtype=int)
    if ng == 0:
        return dict(valid=valid, overlap=overlap, regr=regr, best_anchor=best_anchor, n_for_gt=n_for_gt)
    gt = np.zeros((ng, 4))      # x1, x2, y1, y2 in resized-image pixels (utils.py:608-613)
    gt_boxes = np.asarray(gt_boxes, dtype=np.float64)
    gt[:, 0] = gt_boxes[:, 0] * (rw / float(width))
    gt[:, 1] = gt_boxes[:, 2] * (rw / float(width))
    gt[:, 2] = gt_boxes[:, 1] * (rh / float(height))
    gt[:, 3] = gt_boxes[:, 3] * (rh / float(height))
    fgmask = np.asarray(gt_is_bg) == 0

    # enumerate anchors in the reference's visiting order
    recs = []
    for si, s in enumerate(C.anchor_box_scales):
        for ri, r in enumerate(ratios):
            aw, ah = s * r[0], s * r[1]
            ix = np.arange(fw)
            jy = np.arange(fh)
            x1 = ds * (ix + 0.5) - aw / 2
            x2 = ds * (ix + 0.5) + aw / 2
            y1 = ds * (jy + 0.5) - ah / 2
            y2 = ds * (jy + 0.5) + ah / 2
            okx = ~((x1 < 0) | (x2 > rw))
            oky = ~((y1 < 0) | (y2 > rh))
            IX, JY = np.meshgrid(ix[okx], jy[oky], indexing="ij")     # ix outer, jy inner
            n = IX.size
            if n == 0:
                continue
            X1, Y1 = np.meshgrid(x1[okx], y1[oky], indexing="ij")
            X2, Y2 = np.meshgrid(x2[okx], y2[oky], indexing="ij")
            recs.append(np.stack([IX.ravel(), JY.ravel(), np.full(n, ri), np.full(n, si),
                                  X1.ravel(), Y1.ravel(), X2.ravel(), Y2.ravel()], 1))
    if not recs:
        return dict(valid=valid, overlap=overlap, regr=regr, best_anchor=best_anchor, n_for_gt=n_for_gt)
    rec = np.concatenate(recs)
    aix = rec[:, 0].astype(int); ajy = rec[:, 1].astype(int)
    ari = rec[:, 2].astype(int); asi = rec[:, 3].astype(int)
    ax1, ay1, ax2, ay2 = rec[:, 4], rec[:, 5], rec[:, 6], rec[:, 7]
    ach = ari + nr * asi
    n = len(rec)

    abox = np.stack([ax1, ay1, ax2, ay2], 1)[:, None, :]                       # (n,1,4)
    gbox = np.stack([gt[:, 0], gt[:, 2], gt[:, 1], gt[:, 3]], 1)[None, :, :]  # (1,g,4) as x1,y1,x2,y2
    iou = iou_pairs(np.broadcast_to(gbox, (n, ng, 4)), np.broadcast_to(abox, (n, ng, 4)))   # (n,g) f64

    # regression targets of every (anchor, gt) pair (utils.py:669-687)
    cx = (gt[:, 0] + gt[:, 1]) / 2.0
    cy = (gt[:, 2] + gt[:, 3]) / 2.0
    cxa = (ax1 + ax2) / 2.0
    cya = (ay1 + ay2) / 2.0
    aw_ = ax2 - ax1
    ah_ = ay2 - ay1
    tx = (cx[None, :] - cxa[:, None]) / aw_[:, None]
    ty = (cy[None, :] - cya[:, None]) / ah_[:, None]
    with np.errstate(divide="ignore", invalid="ignore"):
        tw = np.log((gt[:, 1] - gt[:, 0])[None, :] / aw_[:, None])
        th = np.log((gt[:, 3] - gt[:, 2])[None, :] / ah_[:, None])

    # every visited anchor is written as 'neg' first (the 'neutral' branch never fires: utils.py:720)
    valid[ajy, aix, ach] = 1
    posmat = (iou > C.rpn_max_overlap) & fgmask[None, :]
    is_pos = posmat.any(1)
    n_for_gt[:] = posmat.sum(0)
    if is_pos.any():
        masked = np.where(posmat, iou, -1.0)
        kbest = masked.argmax(1)                      # first maximum in gt order (strict '>')
        p = np.nonzero(is_pos)[0]
        overlap[ajy[p], aix[p], ach[p]] = 1
        for q, col in enumerate((tx, ty, tw, th)):
            regr[ajy[p], aix[p], 4 * ach[p] + q] = col[p, kbest[p]]

    # per-GT best anchor: first anchor in visiting order with the largest fp32 IoU (> 0)
    iou32 = iou.astype(np.float32)
    for k in range(ng):
        if not fgmask[k]:
            continue
        m = iou32[:, k].max()
        if not (m > np.float32(0.0)):
            continue
        j = int(np.argmax(iou32[:, k]))
        best_anchor[k] = [ajy[j], aix[j], ari[j], asi[j]]
        if n_for_gt[k] == 0:
            # fallback positive (utils.py:741-766); its deltas were stored as fp32 (utils.py:605,700)
            ch = ari[j] + nr * asi[j]
            valid[ajy[j], aix[j], ch] = 1
            overlap[ajy[j], aix[j], ch] = 1
            regr[ajy[j], aix[j], 4 * ch:4 * ch + 4] = np.array([tx[j, k], ty[j, k], tw[j, k], th[j, k]], dtype=np.float32)
    return dict(valid=valid, overlap=overlap, regr=regr, best_anchor=best_anchor, n_for_gt=n_for_gt)


def subsample_anchor_targets(valid_chw, overlap_chw, max_regions=256):
    """utils.py:777-813 on (1,A,H,W) arrays, in place; consumes the global NumPy RNG.
    Raises KeyError exactly where the reference does (probability table built from the
    negatives' channel histogram, utils.py:789)."""
    pos = np.where((overlap_chw[0] == 1) & (valid_chw[0] == 1))
    neg = np.where((overlap_chw[0] == 0) & (valid_chw[0] == 1))
    n_pos, n_neg = len(pos[0]), len(neg[0])
    if n_pos > max_regions / 2:
        chans = set(np.unique(neg[0]).tolist())
        for l in pos[0]:
            if int(l) not in chans:
                raise KeyError(l)
        # the reference computes (count/n_pos)/count per element (= 1/n_pos up to rounding)
        cu, cc = np.unique(neg[0], return_counts=True)
        table = dict(zip(cu.tolist(), cc.tolist()))
        p = [((table[int(l)] / n_pos) / table[int(l)]) for l in pos[0]]
        off = np.random.choice(n_pos, n_pos - int(max_regions / 2), replace=False, p=p)
        valid_chw[0, pos[0][off], pos[1][off], pos[2][off]] = 0
        n_pos = int(max_regions / 2)
    if n_neg + n_pos > max_regions:
        cu, cc = np.unique(neg[0], return_counts=True)
        table = dict(zip(cu.tolist(), cc.tolist()))
        p = [((table[int(l)] / n_neg) / table[int(l)]) for l in neg[0]]
        off = np.random.choice(n_neg, n_neg - n_pos, replace=False, p=p)
        valid_chw[0, neg[0][off], neg[1][off], neg[2][off]] = 0
    return n_pos


def anchor_targets(C, gt_boxes, gt_is_bg, width, height, rw, rh, feat_size_fn):
    """utils.py:554-822 complete: returns (y_rpn_cls (1,2A,H,W), y_rpn_regr (1,8A,H,W),
    best_anchor (g,4), n_pos) -- NCHW and *unscaled*, as the reference function returns them."""
    fw, fh = feat_size_fn(rw, rh)
    d = anchor_targets_dense(C, gt_boxes, gt_is_bg, width, height, rw, rh, fw, fh)
    ov = np.transpose(d["overlap"], (2, 0, 1))[None]
    va = np.transpose(d["valid"], (2, 0, 1))[None]
    rg = np.transpose(d["regr"], (2, 0, 1))[None]
    n_pos = subsample_anchor_targets(va, ov)
    y_cls = np.concatenate([va, ov], axis=1)
    y_regr = np.concatenate([np.repeat(ov, 4, axis=1), rg], axis=1)
    return y_cls, y_regr, d["best_anchor"], n_pos


def to_train_layout(y_cls, y_regr, std_scaling):
    """utils.py:475-478: scale the regression half by std_scaling, NCHW -> NHWC."""
    y_regr = y_regr.copy()
    y_regr[:, y_regr.shape[1] // 2:] *= std_scaling
    return np.transpose(y_cls, (0, 2, 3, 1)), np.transpose(y_regr, (0, 2, 3, 1))


# ----------------------------------------------------------------------------------
# detector post-processing (G8, N3)
# ----------------------------------------------------------------------------------
def real_coords(ratio, x1, y1, x2, y2):
    """RADNet.py:44-51: floor-divide by the resize ratio, then round."""
    return tuple(int(round(v // ratio)) for v in (x1, y1, x2, y2))


def spp_decode(R, detector_predict, C, bbox_threshold=0.7):
    """RADNet.py:104-154.  R (n,4) xywh fmap units; detector_predict(rois (1,n_rois,4)) ->
    [P_cls (1,n_rois,nc), P_regr (1,n_rois,4(nc-1))].  Returns ({class: boxes}, {class: probs})
    with boxes in resized-image pixels (x rpn_stride)."""
    inv = {v: k for k, v in C.class_mapping.items()}
    k = C.n_rois
    boxes, probs = {}, {}
    n = R.shape[0]
    for c0 in range(0, n, k):
        chunk = R[c0:c0 + k]
        if chunk.shape[0] < k:
            pad = np.zeros((k, 4), dtype=R.dtype)
            pad[:chunk.shape[0]] = chunk
            pad[chunk.shape[0]:] = chunk[0]
            chunk = pad
        P_cls, P_regr = detector_predict(chunk[None])
        for ii in range(k):
            c = int(np.argmax(P_cls[0, ii]))
            pmax = np.max(P_cls[0, ii])
            if pmax < bbox_threshold or c == P_cls.shape[2] - 1:
                continue
            name = inv[c]
            x, y, w, h = chunk[ii]
            tx, ty, tw, th = P_regr[0, ii, 4 * c:4 * c + 4]
            s = C.classifier_regr_std
            x, y, w, h = decode_delta_scalar(x, y, w, h, tx / s[0], ty / s[1], tw / s[2], th / s[3])
            st = C.rpn_stride
            boxes.setdefault(name, []).append([st * x, st * y, st * (x + w), st * (y + h)])
            probs.setdefault(name, []).append(pmax)
    return boxes, probs


def merge_nms(boxes, probs, avg_thr=0.2, conf_thr=0.8, n_avg=5):
    """RADNet.py:156-240 ("final_nms"): greedy clustering by IoU against the current best box;
    each cluster is replaced by the mean of its confident members (prob > conf_thr), or of its
    n_avg best members when none is confident."""
    boxes = np.asarray(boxes)
    if len(boxes) == 0:
        return []
    assert np.all(boxes[:, 0] < boxes[:, 2]) and np.all(boxes[:, 1] < boxes[:, 3])
    b = boxes.astype(np.float64)
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    area = (x2 - x1) * (y2 - y1)
    idxs = np.argsort(probs, kind="stable")
    groups = []
    while len(idxs) > 0:
        last = len(idxs) - 1
        i = idxs[last]
        rest = idxs[:last]
        ww = np.maximum(0, np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]))
        hh = np.maximum(0, np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]))
        inter = ww * hh
        ov = inter / (area[i] + area[rest] - inter + 1e-6)
        member = np.concatenate((np.where(ov > avg_thr)[0], [last]))
        mp = probs[idxs[member]]
        if mp.max() < conf_thr:
            chosen = idxs[member][-n_avg:]
        else:
            chosen = idxs[member][np.nonzero(mp > conf_thr)[0]]
        groups.append(chosen)
        idxs = np.delete(idxs, member)
    nb = [np.rint(b[g].mean(axis=0)).astype("int") for g in groups]
    npb = [probs[g].mean() for g in groups]
    return np.array(nb), np.array(npb)
