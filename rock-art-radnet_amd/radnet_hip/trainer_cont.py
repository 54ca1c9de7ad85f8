"""The reference's training iteration with cont_train.py trainability (stages 3-4 of the base unfrozen) on one MI355X.

Differences from trainer.TrainStep (train.py mode, whole base frozen):
  * the RPN loss also drives stages 3-4 (through rpn_conv1), so Adam #1 moves the base: stages 3-4 are re-run before the
    re-prediction -- only the stem (conv1 .. stage 2) is computed once per image;
  * the classifier loss also drives stages 3-4 (through res5a, the RoI crop-resize and the feature map), Adam #2 applies
    to the head and, with its OWN moments, to the same stage-3/4 weights (cont_train.py:169-185).
Mini-batch semantics as TrainStep: every image is an independent reference iteration on the same weights, each optimizer
applies once with the mean gradient over all images of all ranks.  Data parallel: the stage-3/4 gradient arena joins both
exchanges (19.0 + 33.2 MB before Adam #1, 60.1 + 33.2 MB before Adam #2), synchronously -- the base moves
with both optimizers, so neither update can be deferred across the next step's base forward as in train.py mode.
"""
import numpy as np
import torch

from . import engine as E
from .trainer import allreduce_grad_arena, new_img_size


class ContTrainStep:

    def __init__(self, eng, dist_group=None, world_size=1):
        self.eng = eng
        self._pre = None                 # next batch's labelling kernels, upload and frozen stem, enqueued on the side lane
        self._slot = 0
        self.world = world_size
        self.group = dist_group
        self.skipped_head_steps = 0
        self.dropped_images = 0
        self.last = None
        self.capture = None
        dev = eng.dev
        self._rpn_l = torch.zeros(64, 2, dtype=torch.float32, device=dev)
        self._det_l = torch.zeros(64, 3, dtype=torch.float32, device=dev)

    def _gt(self, s):
        if "_gt_dev" not in s:
            cm = self.eng.C.class_mapping
            boxes = np.array([[b["x1"], b["y1"], b["x2"], b["y2"]] for b in s["bboxes"]], dtype=np.float64).reshape(-1, 4)
            isbg = np.array([1 if b["class"] == "bg" else 0 for b in s["bboxes"]], dtype=np.int32)
            cls = np.array([cm[b["class"]] for b in s["bboxes"]], dtype=np.int32)
            s["_gt_dev"] = self.eng.upload_gt(boxes, isbg, cls)
        return s["_gt_dev"]

    def _launch_stem(self, batch, slot):
        """What does not depend on any trainable weight: labelling kernels, upload, conv1 .. stage 2 (frozen)."""
        eng = self.eng
        nloc = len(batch)
        tp, plans = [], []
        for i, s in enumerate(batch):
            H, W = s["img"].shape[:2]
            tp.append(eng.anchor_targets_launch(self._gt(s), s["width"], s["height"], W, H, slot=slot * nloc + i))
        for i, s in enumerate(batch):
            bp = eng.upload_image(s["img"], slot=slot * nloc + i)
            eng.stem_forward(bp)                       # frozen: once per image
            plans.append(bp)
        return dict(batch=batch, tp=tp, plans=plans)

    def step(self, batch, upcoming=None):
        """upcoming (optional): the next call's batch -- its labelling kernels, upload and frozen stem run on the engine's
        side lane beside this step (everything else of cont_train.py's step depends on weights both optimizers move)."""
        eng = self.eng
        C = eng.C
        nloc = len(batch)
        ntot = nloc * self.world
        st = self._pre if self._pre is not None and self._pre["batch"] is batch else None
        self._pre = None
        if st is None:
            st = self._launch_stem(batch, self._slot)
        else:
            eng.after(st["done"])
        self._slot ^= 1
        if upcoming and hasattr(eng, "lane") and not eng.ctx.timing_on:
            ev = eng.mark()                            # the other buffer set was last read by the previous step (main lane)
            with eng.lane("side"):
                eng.after(ev)
                self._pre = self._launch_stem(upcoming[0], self._slot)
                self._pre["done"] = eng.mark()
        tp, plans, rps = st["tp"], st["plans"], []
        for bp in plans:
            eng.s34_forward(bp)
            rps.append(eng.rpn_forward(bp))
        # ---- RPN model: loss, gradients of the RPN convs and (via dL/dF) of stages 3-4, Adam #1 over both
        dead = [False] * nloc          # labeller failures: the reference's generator skips such a sample (utils.py:461-465)
        n_live = 0
        for i in range(nloc):
            try:
                ycls, yregr, _ = eng.anchor_targets_finish(tp[i])
            except KeyError as e:
                import sys
                sys.stderr.write("radnet: anchor labelling failed (KeyError: %s); sample skipped as the reference's generator does\n" % (e,))
                dead[i] = True
                self.dropped_images += 1
                continue
            eng.set_accumulate(rps[i]["bwd"], n_live > 0, prezeroed=True)
            eng.set_accumulate(plans[i]["bwd34"], n_live > 0, prezeroed=True)
            eng.rpn_backward(rps[i], ycls, yregr, self._rpn_l[i])      # ends with rpn_conv1's dgrad into plan['dF']
            eng.s34_backward(plans[i])
            n_live += 1
        if n_live == 0 and self.world == 1:
            self.last = (nloc, 0, dead)
            return self                                # whole batch dropped: no optimizer step
        allreduce_grad_arena(eng.rpn_arena.g, self.world, self.group)
        allreduce_grad_arena(eng.s34_arena.g, self.world, self.group)
        eng.adam(eng.rpn_arena, grad_scale=1.0 / ntot)
        eng.adam_s34(0, grad_scale=1.0 / ntot)
        # ---- classifier model on the moved base
        n_head = 0
        for i, bp in enumerate(plans):
            if dead[i]:
                continue
            s = batch[i]
            rp = rps[i]
            eng.s34_forward(bp)
            eng._run(rp["fwd"])
            R, Rn = eng.proposals(rp, overlap_thresh=0.7, max_boxes=300)
            rw, rh = new_img_size(s["width"], s["height"], C.img_size)
            P, cls, n = eng.roi_targets(R, Rn, self._gt(s), s["width"], s["height"], rw, rh)
            kept = np.nonzero(cls >= 0)[0]
            if n <= 0 or len(kept) == 0:
                self.skipped_head_steps += 1
                continue
            sel_k, _ = E.select_samples(cls[kept], eng.bg, C.n_rois)
            sel = kept[np.asarray(sel_k, dtype=np.int64)]
            if self.capture is not None:
                self.capture.append(dict(pred=rp["pred"].cpu().numpy().copy(), R=R[:n].cpu().numpy().copy(), keep=(cls >= 0).copy(),
                                         cls=cls.copy(), sel_kept=list(sel_k)))
            hp = eng._plan_head(C.n_rois, bp["fh"], bp["fw"], bp["F"])
            eng.pack_roi_batch(P, sel, hp)
            eng.head_forward(hp, training=True)
            eng.set_accumulate(hp["bwd"], n_head > 0, prezeroed=True)
            eng.set_accumulate(bp["bwd34"], n_head > 0, prezeroed=True)
            eng.head_backward(hp, accumulate=True, loss_out=self._det_l[n_head])   # ... -> RoI crop-resize -> plan['dF']
            eng.s34_backward(bp)
            n_head += 1
        if n_head > 0 or self.world > 1:          # a rank whose images all skipped still joins the exchange (zeros)
            allreduce_grad_arena(eng.head_arena.g, self.world, self.group)
            allreduce_grad_arena(eng.s34_arena.g, self.world, self.group)
            eng.adam(eng.head_arena, grad_scale=1.0 / ntot)
            eng.refresh_head_shift()
            eng.adam_s34(1, grad_scale=1.0 / ntot)
        self.last = (nloc, n_head, dead)
        return self

    def losses(self):
        nloc, n_head, dead = self.last
        live = [i for i in range(nloc) if not dead[i]]
        r = self._rpn_l[:nloc].cpu().numpy()[live].mean(0) if live else np.full(2, np.nan, np.float32)
        d = self._det_l[:max(n_head, 1)].cpu().numpy().mean(0) if n_head else np.zeros(3, np.float32)
        return {"rpn_cls": float(r[0]), "rpn_regr": float(r[1]), "det_cls": float(d[0]), "det_regr": float(d[1]), "det_acc": float(d[2]),
                "n_head": n_head, "dropped": nloc - len(live)}
