"""The reference's 4-phase training iteration (train.py:278-402) on one MI355X, data-parallel over images.

    phase A  anchor targets (utils.calc_region_props)           device kernels + host RNG subsampling
    phase B  base forward ONCE per image (train.py mode: the whole base is frozen, so the three base passes
             the reference runs at train.py:288,291,393 are bit-identical -- SURVEY.md 3.1)
    phase C  model_rpn.train_on_batch: RPN fwd + losses + bwd, grads summed over the local images,
             [RCCL all-reduce], Adam #1
    phase D  model_rpn.predict_on_batch with the UPDATED weights -> rpn_to_roi -> calc_iou ->
             get_selected_samples (host RNG) -> model_classifier.train_on_batch: head fwd + losses + bwd,
             [RCCL all-reduce], Adam #2
Batch semantics (the reference is batch-1 only): every image is an independent reference step and the
gradient is the mean over all images of all ranks (SURVEY.md 8d cfg 4).
"""
import numpy as np
import torch

from . import engine as E


def new_img_size(width, height, min_side):
    """utils.get_new_img_size (utils.py:65-75)."""
    if width <= height:
        f = float(min_side) / width
        return min_side, int(f * height)
    f = float(min_side) / height
    return int(f * width), min_side


class TrainStep:

    def __init__(self, eng, dist_group=None, world_size=1):
        self.eng = eng
        self.world = world_size
        self.group = dist_group
        self.skipped_head_steps = 0
        self.last = None
        self.capture = None         # set to [] to record per-image intermediates (tests: stage-wise parity)

    def _allreduce(self, arena):
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(arena.g, op=dist.ReduceOp.SUM, group=self.group)

    def step(self, batch):
        """batch: list of dicts {img: uint8 BGR HWC (already at network size), bboxes: [{class,x1,x2,y1,y2}],
        width, height: source-frame size the boxes refer to}.  Returns dict of the five Keras-order losses
        averaged over the local images: rpn_cls, rpn_regr, det_cls, det_regr, det_acc (+ n_head)."""
        eng, C = eng_C(self.eng)
        nloc = len(batch)
        ntot = nloc * self.world
        cm = C.class_mapping
        # ---- phase A+B: targets and base features
        plans, tgts = [], []
        for i, s in enumerate(batch):
            H, W = s["img"].shape[:2]
            gt = np.array([[b["x1"], b["y1"], b["x2"], b["y2"]] for b in s["bboxes"]], dtype=np.float64).reshape(-1, 4)
            isbg = np.array([1 if b["class"] == "bg" else 0 for b in s["bboxes"]], dtype=np.int32)
            ycls, yregr, _, _ = eng.anchor_targets(gt, isbg, s["width"], s["height"], W, H, slot=i)
            bp = eng.upload_image(s["img"], slot=i)
            eng.base_forward(bp)
            plans.append(bp)
            tgts.append((ycls, yregr, gt))
        # ---- phase C: RPN train
        rpn_l = torch.zeros(2, dtype=torch.float32, device=eng.dev)
        rps = []
        for i, bp in enumerate(plans):
            rp = eng.rpn_forward(bp)
            eng.set_accumulate(rp["bwd"], i > 0)
            eng.rpn_backward(rp, tgts[i][0], tgts[i][1])
            rpn_l += eng.rpn_losses            # tiny device add (logging only)
            rps.append(rp)
        self._allreduce(eng.rpn_arena)
        eng.adam(eng.rpn_arena, grad_scale=1.0 / ntot)
        # ---- phase D: re-predict with updated weights, propose, label, sample, head train
        det_l = torch.zeros(3, dtype=torch.float32, device=eng.dev)
        n_head = 0
        first = True
        for i, bp in enumerate(plans):
            s = batch[i]
            H, W = s["img"].shape[:2]
            rp = rps[i]
            eng._run(rp["fwd"])
            R, Rn = eng.proposals(rp, overlap_thresh=0.7, max_boxes=300)
            gt_cls = np.array([cm[b["class"]] for b in s["bboxes"]], dtype=np.int32)
            n = int(Rn.cpu()[0])                                   # sync #1 of this image
            if n <= 0:
                self.skipped_head_steps += 1
                continue
            rw, rh = new_img_size(s["width"], s["height"], C.img_size)       # rpn.py:189 recomputes it from the config
            P = eng.roi_targets(R, n, tgts[i][2], gt_cls, s["width"], s["height"], rw, rh)
            keep = P["keep"][:n].cpu().numpy().astype(bool)        # sync #2
            cls = P["cls"][:n].cpu().numpy()
            kept = np.nonzero(keep)[0]
            if len(kept) == 0:                                     # calc_iou -> None: the reference skips the head step
                self.skipped_head_steps += 1
                continue
            sel_k, _ = E.select_samples(cls[kept], eng.bg, C.n_rois)
            sel = kept[np.asarray(sel_k, dtype=np.int64)]
            if self.capture is not None:
                self.capture.append(dict(pred=rp["pred"].cpu().numpy().copy(), R=R[:n].cpu().numpy().copy(), keep=keep.copy(),
                                         cls=cls.copy(), sel_kept=list(sel_k)))
            hp = eng._plan_head(C.n_rois, bp["fh"], bp["fw"], bp["F"])
            eng.pack_roi_batch(P, sel, hp)
            eng.head_forward(hp)
            eng.set_accumulate(hp["bwd"], not first, dense=hp)
            eng.head_backward(hp, accumulate=not first)
            first = False
            det_l += eng.det_losses
            n_head += 1
        if n_head == 0 and self.world > 1:
            eng.zero_grads(eng.head_arena)
        if n_head > 0 or self.world > 1:
            self._allreduce(eng.head_arena)
            eng.adam(eng.head_arena, grad_scale=1.0 / ntot)
            eng.refresh_head_shift()
        self.last = (rpn_l, det_l, nloc, n_head)
        return self

    def losses(self):
        """Host copy of the last step's mean losses (one device sync)."""
        rpn_l, det_l, nloc, n_head = self.last
        r = (rpn_l / nloc).cpu().numpy()
        d = (det_l / max(n_head, 1)).cpu().numpy()
        return {"rpn_cls": float(r[0]), "rpn_regr": float(r[1]), "det_cls": float(d[0]), "det_regr": float(d[1]), "det_acc": float(d[2]),
                "n_head": n_head}


def eng_C(eng):
    return eng, eng.C
