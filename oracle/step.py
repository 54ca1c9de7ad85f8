"""ORACLE (test infrastructure, not product): one training iteration of the reference (train.py:278-402)
and the RPN-only / predict-tile paths, composed from oracle/glue.py (pinned by goldens) and oracle/dense.py
(parity unpinned, torch-cross-checked).  Used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg only -- as the checker / the CPU baseline, never as the thing shipped.
"""
import numpy as np

from . import dense, glue


def feat_size(w, h):
    return glue.resnet50_feat_len(w), glue.resnet50_feat_len(h)


class OracleTrainer:
    """State of the reference's training script for ResNet50 in train.py mode (base frozen): weights plus
    the two independent Adam instances (train.py:236-252)."""

    def __init__(self, C, P, lr=5e-5, keras2_bce=True):
        self.C, self.P = C, P
        self.A = len(C.anchor_box_scales) * len(C.anchor_box_ratios)
        self.nc = len(C.class_mapping)
        self.keras2_bce = keras2_bce
        self.opt_rpn = dense.AdamState(P, dense.RPN_TRAINABLE, lr)
        self.opt_head = dense.AdamState(P, dense.head_trainable(self.nc), lr)

    def targets(self, sample):
        """Data-generator half (utils.py:438-478): anchor targets in the NHWC / scaled layout the model consumes."""
        C = self.C
        H, W = sample["img"].shape[:2]
        gt = np.array([[b["x1"], b["y1"], b["x2"], b["y2"]] for b in sample["bboxes"]], dtype=np.float64).reshape(-1, 4)
        isbg = np.array([1 if b["class"] == "bg" else 0 for b in sample["bboxes"]])
        ycls, yregr, _, _ = glue.anchor_targets(C, gt, isbg, sample["width"], sample["height"], W, H, feat_size)
        return glue.to_train_layout(ycls, yregr, C.std_scaling)

    def step(self, sample, detail=None, override_R=None):
        """One iteration on one image.  Returns [rpn_cls, rpn_regr, det_cls, det_regr, det_acc] or the first two
        and None when calc_iou keeps nothing (train.py:378-380).  `detail` (dict) receives intermediates.
        `override_R`: proposals to label instead of this trainer's own (stage-wise parity checks feed the
        device's proposals here, because a 1e-7 score difference may legitimately reorder near-tied boxes)."""
        C, P = self.C, self.P
        y_cls, y_regr = self.targets(sample)
        x = dense.preprocess_caffe_bgr(sample["img"])
        F = dense.base_forward(P, x)                                        # frozen: identical in all three passes
        l_rpn, g_rpn = dense.rpn_losses_and_grads(P, F, y_cls.astype(np.float32), y_regr.astype(np.float32), self.A, self.keras2_bce)
        self.opt_rpn.apply(P, g_rpn)                                        # train.py:288
        p, r, _ = dense.rpn_forward(P, F)                                   # train.py:291 (post-update weights)
        R = glue.rpn_to_roi(p, r, C, use_regr=True, overlap_thresh=0.7, max_boxes=300)
        if detail is not None:
            detail["R_own"] = R
        if override_R is not None:
            R = override_R
        gt = np.array([[b["x1"], b["y1"], b["x2"], b["y2"]] for b in sample["bboxes"]], dtype=np.float64).reshape(-1, 4)
        gcls = np.array([C.class_mapping[b["class"]] for b in sample["bboxes"]])
        X2, Y1, Y2, _ = glue.roi_targets(R, gt, gcls, sample["width"], sample["height"], C)
        if detail is not None:
            detail.update(F=F, g_rpn=g_rpn, p=p, r=r, R=R, X2=X2, Y1=Y1, Y2=Y2)
        if X2 is None:
            return [l_rpn[1], l_rpn[2], None, None, None]
        sel, _ = glue.select_samples(Y1, C.n_rois)
        l_det, g_head = dense.head_losses_and_grads(P, F, X2[0, sel].astype(np.float32), Y1[:, sel].astype(np.float32),
                                                    Y2[:, sel].astype(np.float32), self.nc)
        self.opt_head.apply(P, g_head)                                      # train.py:393
        if detail is not None:
            detail.update(sel=sel, g_head=g_head)
        return [l_rpn[1], l_rpn[2], l_det[1], l_det[2], l_det[3]]


def oracle_validate(trainer, samples, override_R=None):
    """The reference's validation loop (train.py:478-561) on an OracleTrainer / OracleTrainerVGG: per sample test_on_batch of
    the RPN model, proposals from predict_on_batch, calc_iou, get_selected_samples, test_on_batch of the classifier (inference
    arithmetic: no dropout); a sample without an overlapping RoI is skipped whole (`continue` before the appends).  Weights are
    not touched.  Returns the list of per-sample records [rpn_cls, rpn_regr, det_cls, det_regr, det_acc, n_pos] of the samples
    used.  override_R: per sample, proposals to label instead of the oracle's own (the device's: a 1e-7 score difference may
    legitimately reorder near-tied boxes)."""
    C, P = trainer.C, trainer.P
    vgg = getattr(trainer, "vgg", None)
    out = []
    for k, s in enumerate(samples):
        y_cls, y_regr = trainer.targets(s)
        x = dense.preprocess_caffe_bgr(s["img"])
        F = vgg.base_forward(P, x) if vgg is not None else dense.base_forward(P, x)
        p, r, _ = dense.rpn_forward(P, F)
        l_cls, _ = dense.rpn_loss_cls(y_cls.astype(np.float32), p, trainer.A, trainer.keras2_bce)
        l_regr, _ = dense.smooth_l1_masked(y_regr.astype(np.float32), r, 4 * trainer.A)
        R = glue.rpn_to_roi(p, r, C, use_regr=True, overlap_thresh=0.7, max_boxes=300)
        if override_R is not None:
            R = override_R[k]
        gt = np.array([[b["x1"], b["y1"], b["x2"], b["y2"]] for b in s["bboxes"]], dtype=np.float64).reshape(-1, 4)
        gcls = np.array([C.class_mapping[b["class"]] for b in s["bboxes"]])
        X2, Y1, Y2, _ = glue.roi_targets(R, gt, gcls, s["width"], s["height"], C)
        if X2 is None:
            continue
        sel, n_pos = glue.select_samples(Y1, C.n_rois)
        rois = X2[0, sel].astype(np.float32)
        if vgg is not None:
            pc, pr, _ = vgg.head_forward(P, F, rois, trainer.nc, None)
        else:
            pc, pr, _ = dense.head_forward(P, F, rois, trainer.nc)
        lc, _ = dense.class_loss_cls(Y1[:, sel].astype(np.float32), pc)
        lr_, _ = dense.smooth_l1_masked(Y2[:, sel].astype(np.float32), pr, 4 * (trainer.nc - 1))
        out.append([l_cls, l_regr, lc, lr_, dense.categorical_accuracy(Y1[:, sel].astype(np.float32), pc), n_pos, R])
    return out


class OracleTrainerCont(OracleTrainer):
    """cont_train.py trainability (SURVEY.md 8d cfg 2, secondary): ResNet50 stages 3-4 train in BOTH models, each model's
    Adam keeps its own moments for those shared weights (cont_train.py:169-185), lr 2e-5.  Per iteration: RPN loss ->
    gradients of the RPN convs and (through rpn_conv1) of stages 3-4 -> Adam #1; the base is then re-run (its weights
    moved) for the re-prediction and the classifier phase; classifier loss -> gradients of the head and (through the
    RoI crop-resize) of stages 3-4 -> Adam #2."""

    def __init__(self, C, P, lr=2e-5, keras2_bce=True):
        super().__init__(C, P, lr, keras2_bce)
        self.shared = dense.s34_trainable()
        self.opt_rpn = dense.AdamState(P, list(dense.RPN_TRAINABLE) + self.shared, lr)
        self.opt_head = dense.AdamState(P, dense.head_trainable(self.nc) + self.shared, lr)

    def step(self, sample, detail=None, override_R=None):
        C, P = self.C, self.P
        y_cls, y_regr = self.targets(sample)
        x = dense.preprocess_caffe_bgr(sample["img"])
        F, caches = dense.base_forward(P, x, want_cache=True)
        p, r, rc = dense.rpn_forward(P, F)
        l_cls, dp = dense.rpn_loss_cls(y_cls.astype(np.float32), p, self.A, self.keras2_bce)
        l_regr, dr = dense.smooth_l1_masked(y_regr.astype(np.float32), r, 4 * self.A)
        g_rpn, dF = dense.rpn_backward(P, rc, dp, dr, need_dF=True)
        g_rpn.update(dense.base_backward(P, caches, dF))
        if detail is not None:
            detail.update(g_rpn=g_rpn, dF_rpn=dF, F0=F)
        self.opt_rpn.apply(P, g_rpn)
        F, caches = dense.base_forward(P, x, want_cache=True)               # stages 3-4 moved: new feature map
        p, r, _ = dense.rpn_forward(P, F)
        R = glue.rpn_to_roi(p, r, C, use_regr=True, overlap_thresh=0.7, max_boxes=300)
        if detail is not None:
            detail["R_own"] = R
        if override_R is not None:
            R = override_R
        gt = np.array([[b["x1"], b["y1"], b["x2"], b["y2"]] for b in sample["bboxes"]], dtype=np.float64).reshape(-1, 4)
        gcls = np.array([C.class_mapping[b["class"]] for b in sample["bboxes"]])
        X2, Y1, Y2, _ = glue.roi_targets(R, gt, gcls, sample["width"], sample["height"], C)
        if X2 is None:
            return [l_cls, l_regr, None, None, None]
        sel, _ = glue.select_samples(Y1, C.n_rois)
        rois = X2[0, sel].astype(np.float32)
        pc, pr, hc = dense.head_forward(P, F, rois, self.nc)
        lc, dpc = dense.class_loss_cls(Y1[:, sel].astype(np.float32), pc)
        lr_, dpr = dense.smooth_l1_masked(Y2[:, sel].astype(np.float32), pr, 4 * (self.nc - 1))
        g_head, dpooled = dense.head_backward(P, hc, dpc, dpr, need_dpooled=True)
        dF2 = dense.roi_crop_resize_bwd(F.shape, rois, 14, dpooled)
        g_head.update(dense.base_backward(P, caches, dF2))
        if detail is not None:
            detail.update(F=F, R=R, sel=sel, g_head=g_head, dF_head=dF2, dpooled=dpooled)
        self.opt_head.apply(P, g_head)
        return [l_cls, l_regr, lc, lr_, dense.categorical_accuracy(Y1[:, sel].astype(np.float32), pc)]


class OracleTrainerVGG(OracleTrainer):
    """The same iteration with the VGG16 base model (BASELINE cfg 5; vgg16.py:29-124): block1..5 frozen as train.py freezes
    every base model, rpn_layer on the 512-channel map, classifier = RoI crop-resize 7x7 -> fc1 -> Dropout -> fc2 -> Dropout
    -> two dense heads.  TF's dropout RNG cannot be matched, so the two keep-masks of a step (already x2) are handed in
    (`masks`), as oracle/vgg.py documents; a test gives the device path the same masks."""

    def __init__(self, C, P, lr=5e-5, keras2_bce=True):
        from . import vgg
        self.vgg = vgg
        self.C, self.P = C, P
        self.A = len(C.anchor_box_scales) * len(C.anchor_box_ratios)
        self.nc = len(C.class_mapping)
        self.keras2_bce = keras2_bce
        self.opt_rpn = dense.AdamState(P, dense.RPN_TRAINABLE, lr)
        self.opt_head = dense.AdamState(P, list(vgg.HEAD_TRAINABLE) + ["dense_class_%d" % self.nc, "dense_regress_%d" % self.nc], lr)

    def targets(self, sample):
        C = self.C
        H, W = sample["img"].shape[:2]
        gt = np.array([[b["x1"], b["y1"], b["x2"], b["y2"]] for b in sample["bboxes"]], dtype=np.float64).reshape(-1, 4)
        isbg = np.array([1 if b["class"] == "bg" else 0 for b in sample["bboxes"]])
        fs = lambda w, h: (glue.vgg16_feat_len(w), glue.vgg16_feat_len(h))
        ycls, yregr, _, _ = glue.anchor_targets(C, gt, isbg, sample["width"], sample["height"], W, H, fs)
        return glue.to_train_layout(ycls, yregr, C.std_scaling)

    def step(self, sample, detail=None, override_R=None, masks_fn=None):
        """masks_fn(R) -> (m1, m2) keep-masks [R][4096] x2 for the classifier pass (None: no dropout, i.e. inference arithmetic)."""
        C, P, vgg = self.C, self.P, self.vgg
        y_cls, y_regr = self.targets(sample)
        F = vgg.base_forward(P, dense.preprocess_caffe_bgr(sample["img"]))
        l_rpn, g_rpn = dense.rpn_losses_and_grads(P, F, y_cls.astype(np.float32), y_regr.astype(np.float32), self.A, self.keras2_bce)
        self.opt_rpn.apply(P, g_rpn)
        p, r, _ = dense.rpn_forward(P, F)
        R = glue.rpn_to_roi(p, r, C, use_regr=True, overlap_thresh=0.7, max_boxes=300)
        if detail is not None:
            detail["R_own"] = R
        if override_R is not None:
            R = override_R
        gt = np.array([[b["x1"], b["y1"], b["x2"], b["y2"]] for b in sample["bboxes"]], dtype=np.float64).reshape(-1, 4)
        gcls = np.array([C.class_mapping[b["class"]] for b in sample["bboxes"]])
        X2, Y1, Y2, _ = glue.roi_targets(R, gt, gcls, sample["width"], sample["height"], C)
        if detail is not None:
            detail.update(F=F, g_rpn=g_rpn, p=p, r=r, R=R, X2=X2, Y1=Y1, Y2=Y2)
        if X2 is None:
            return [l_rpn[1], l_rpn[2], None, None, None]
        sel, _ = glue.select_samples(Y1, C.n_rois)
        masks = masks_fn(len(sel)) if masks_fn is not None else None
        l_det, g_head = vgg.head_losses_and_grads(P, F, X2[0, sel].astype(np.float32), Y1[:, sel].astype(np.float32),
                                                  Y2[:, sel].astype(np.float32), self.nc, masks)
        self.opt_head.apply(P, g_head)
        if detail is not None:
            detail.update(sel=sel, g_head=g_head)
        return [l_rpn[1], l_rpn[2], l_det[1], l_det[2], l_det[3]]


def _acc(total, g):
    if total is None:
        return {n: {k: v.astype(np.float64).copy() for k, v in d.items()} for n, d in g.items()}
    for n, d in g.items():
        for k, v in d.items():
            total[n][k] += v
    return total


def _scaled(total, f):
    return {n: {k: v * f for k, v in d.items()} for n, d in total.items()}


def step_batch(trainer, samples, details=None, override_R=None):
    """BASELINE cfg 4 semantics (the reference is batch-1 only, the build defines the mini-batch): every
    image is an independent reference iteration on the SAME weights, each optimizer applies once with the MEAN
    gradient over all B images (an image whose classifier phase is skipped contributes zero to that mean).
    NumPy-RNG order: anchor subsampling of image 0..B-1, then RoI sample selection of image 0..B-1.
    Returns one loss list per image, as OracleTrainer.step does."""
    t = trainer
    C, P, B = t.C, t.P, len(samples)
    tg = [t.targets(s) for s in samples]
    Fs = [dense.base_forward(P, dense.preprocess_caffe_bgr(s["img"])) for s in samples]
    out, g_sum = [], None
    for F, (yc, yr) in zip(Fs, tg):
        l, g = dense.rpn_losses_and_grads(P, F, yc.astype(np.float32), yr.astype(np.float32), t.A, t.keras2_bce)
        out.append([l[1], l[2], None, None, None])
        g_sum = _acc(g_sum, g)
    t.opt_rpn.apply(P, _scaled(g_sum, 1.0 / B))
    h_sum = None
    for i, (s, F) in enumerate(zip(samples, Fs)):
        p, r, _ = dense.rpn_forward(P, F)
        R = glue.rpn_to_roi(p, r, C, use_regr=True, overlap_thresh=0.7, max_boxes=300) if override_R is None else override_R[i]
        gt = np.array([[b["x1"], b["y1"], b["x2"], b["y2"]] for b in s["bboxes"]], dtype=np.float64).reshape(-1, 4)
        gcls = np.array([C.class_mapping[b["class"]] for b in s["bboxes"]])
        X2, Y1, Y2, _ = glue.roi_targets(R, gt, gcls, s["width"], s["height"], C)
        if X2 is None:
            continue
        sel, _ = glue.select_samples(Y1, C.n_rois)
        l, g = dense.head_losses_and_grads(P, F, X2[0, sel].astype(np.float32), Y1[:, sel].astype(np.float32), Y2[:, sel].astype(np.float32), t.nc)
        out[i][2:] = [l[1], l[2], l[3]]
        h_sum = _acc(h_sum, g)
        if details is not None:
            details.append(dict(sel=sel, Y1=Y1))
    if h_sum is not None:
        t.opt_head.apply(P, _scaled(h_sum, 1.0 / B))
    return out


def rpn_only_forward(P, img_bgr_u8):
    """cfg 1: model_rpn.predict on one image -> (cls, regr, F)."""
    F = dense.base_forward(P, dense.preprocess_caffe_bgr(img_bgr_u8))
    p, r, _ = dense.rpn_forward(P, F)
    return p, r, F
