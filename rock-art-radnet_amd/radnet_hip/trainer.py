"""The reference's 4-phase training iteration (train.py:278-402) on one MI355X, data-parallel over images.

    phase A  anchor targets (utils.calc_region_props): device kernels, async copy of the 2 x 28 KB label maps to
             pinned host memory; the RNG-driven subsampling runs on the host WHILE the GPU does phase B
    phase B  base forward ONCE per image (train.py mode: the whole base is frozen, so the three base passes
             the reference runs at train.py:288,291,393 are bit-identical -- SURVEY.md 3.1) + RPN forward
    phase C  model_rpn.train_on_batch: RPN losses + bwd, grads summed over the local images,
             [RCCL all-reduce], Adam #1
    phase D  model_rpn.predict_on_batch with the UPDATED weights -> rpn_to_roi -> calc_iou ->
             get_selected_samples (host RNG) -> model_classifier.train_on_batch: head fwd + losses + bwd,
             [RCCL all-reduce], Adam #2
Batch semantics (the reference is batch-1 only): every image is an independent reference step and the
gradient is the mean over all images of all ranks (SURVEY.md 8d cfg 4).  The order of draws from NumPy's
global RNG is the reference's: subsampling of image 0..B-1, then sample selection of image 0..B-1.
"""
import os

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


def allreduce_grad_arena(flat, world, group=None):
    """Data-parallel exchange of one optimizer's flat gradient arena: SUM over ranks (RCCL over xGMI on the GPUs,
    gloo in the CPU tests).  Returns the factor Adam applies to the summed gradient so that the update uses the MEAN
    over all `world * images_per_rank` images (the caller divides by images_per_rank as well)."""
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / world


def allreduce_grad_arena_start(flat, world, group=None):
    """The same exchange, asynchronous: returns a handle whose wait() orders the CURRENT stream after the reduction
    (None on a single rank).  The collective runs on the backend's own stream, beside whatever is enqueued next."""
    if world > 1:
        import torch.distributed as dist
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)
    return None


class TrainStep:

    def __init__(self, eng, dist_group=None, world_size=1, defer_head_update=None):
        """defer_head_update (default: on when world_size > 1): the all-reduce of the head gradients (60 MB, the larger
        of the two exchanges) is started asynchronously after the head backward and Adam #2 is applied just before
        the NEXT step's head forward -- the first point that reads head weights (the base is frozen and the RPN has
        its own optimizer), so the update order the reference defines is unchanged while the exchange hides under
        the next image's anchor labelling, base forward and RPN phases.  Call flush() after the last
        step."""
        self.eng = eng
        self.world = world_size
        self.group = dist_group
        self.defer_head_update = (world_size > 1) if defer_head_update is None else bool(defer_head_update)
        self._head_pending = None        # (work handle, images in the global batch) of the exchange in flight
        self._pre = None                 # phases A/B of the next batch, enqueued ahead (step(next_batch=...))
        self._parity = 0                 # buffer set the next _launch_ab uses
        # the prefetched phases A/B run on the engine's side stream, concurrently with this step's head phase
        self.side_prefetch = os.environ.get("RADNET_SIDE_PREFETCH", "1") == "1" and hasattr(eng, "on_side_stream")
        self.skipped_head_steps = 0
        self.last = None
        self.capture = None         # set to [] to record per-image intermediates (tests: stage-wise parity)
        self.host_marks = None      # set to [] to record (label, perf_counter) at the host-side phase boundaries
        dev = eng.dev
        self._rpn_l = torch.zeros(64, 2, dtype=torch.float32, device=dev)     # per-image loss slots (logging)
        self._det_l = torch.zeros(64, 3, dtype=torch.float32, device=dev)

    def _allreduce(self, arena):
        allreduce_grad_arena(arena.g, self.world, self.group)

    def _finish_head_update(self):
        if self._head_pending is None:
            return
        work, ntot = self._head_pending
        self._head_pending = None
        if work is not None:
            work.wait()
        self.eng.adam(self.eng.head_arena, grad_scale=1.0 / ntot)
        self.eng.refresh_head_shift()

    def flush(self):
        """Apply a head update still in flight (deferred mode); no-op otherwise."""
        self._finish_head_update()

    def _gt(self, s):
        """Device copy of a sample's ground truth (cached on the sample: uploaded once)."""
        if "_gt_dev" not in s:
            cm = self.eng.C.class_mapping
            boxes = np.array([[b["x1"], b["y1"], b["x2"], b["y2"]] for b in s["bboxes"]], dtype=np.float64).reshape(-1, 4)
            isbg = np.array([1 if b["class"] == "bg" else 0 for b in s["bboxes"]], dtype=np.int32)
            cls = np.array([cm[b["class"]] for b in s["bboxes"]], dtype=np.int32)
            s["_gt_dev"] = self.eng.upload_gt(boxes, isbg, cls)
        return s["_gt_dev"]

    def _launch_a(self, batch, parity):
        """Device half of phase A (anchor labelling + async copy of the label maps) and the weight-independent part of
        phase B (upload, frozen base forward) of every image of `batch`, into the buffer set `parity`."""
        eng = self.eng
        nloc = len(batch)
        tp, plans = [], []
        for i, s in enumerate(batch):
            H, W = s["img"].shape[:2]
            tp.append(eng.anchor_targets_launch(self._gt(s), s["width"], s["height"], W, H, slot=parity * nloc + i))
        for i, s in enumerate(batch):
            bp = eng.upload_image(s["img"], slot=parity * nloc + i)
            eng.base_forward(bp)
            plans.append(bp)
        return dict(batch=batch, tp=tp, plans=plans, rps=None, parity=parity)

    def _launch_b(self, st):
        """RPN forward of every image of a batch whose base forward is enqueued: reads the RPN weights."""
        st["rps"] = [self.eng.rpn_forward(bp) for bp in st["plans"]]
        return st

    def _launch_ab(self, batch, parity):
        return self._launch_b(self._launch_a(batch, parity))

    def step(self, batch, next_batch=None):
        """batch: list of dicts {img: uint8 BGR HWC (already at network size), bboxes: [{class,x1,x2,y1,y2}],
        width, height: source-frame size the boxes refer to}.  Losses of the step: self.losses().

        next_batch (optional, what a prefetching data loader knows): its phases A/B are ENQUEUED while this step waits
        for the RoI class codes -- the one point where the host must drain the GPU (sample selection runs on NumPy's
        RNG).  They read only the frozen base and the RPN weights, which nothing after this step's Adam #1 changes, so
        the arithmetic and the order of RNG draws are exactly those of back-to-back steps; the GPU just never idles
        while the host selects samples and builds the head batch.  The next call must pass that same object."""
        eng = self.eng
        C = eng.C
        nloc = len(batch)
        ntot = nloc * self.world
        marks = self.host_marks

        def mark(label):
            if marks is not None:
                import time
                marks.append((label, time.perf_counter()))

        mark("start")
        # ---- phase A (device half) + phase B, all asynchronous -- unless the previous step already enqueued them
        if self._pre is not None and self._pre["batch"] is batch:
            st = self._pre
        else:
            st = self._launch_ab(batch, self._parity)
        self._pre = None
        if self.side_prefetch:
            eng.join_side()
        self._parity = st["parity"] ^ 1
        tp, plans, rps = st["tp"], st["plans"], st["rps"]
        # side-stream prefetch: the next batch's labelling kernels and frozen base forward need nothing from this step,
        # so they start now, beside this step's RPN backward / proposal / head phases; its RPN forward follows Adam #1
        early = self.side_prefetch and next_batch is not None and not eng.ctx.timing_on
        if early:
            with eng.on_side_stream():
                self._pre = self._launch_a(next_batch, self._parity)
        mark("B: upload + base + rpn forward enqueued")
        # ---- phase A (host half, overlapped with B) + phase C
        # gradient arenas are zero here: allocated zeroed, and every Adam pass clears what it consumed
        for i in range(nloc):
            ycls, yregr, _ = eng.anchor_targets_finish(tp[i])
            mark("A: label maps on host, subsampled, packed")
            eng.set_accumulate(rps[i]["bwd"], i > 0, prezeroed=True)
            eng.rpn_backward(rps[i], ycls, yregr, self._rpn_l[i])
        self._allreduce(eng.rpn_arena)
        eng.adam(eng.rpn_arena, grad_scale=1.0 / ntot)
        if early:
            with eng.on_side_stream():
                self._launch_b(self._pre)
        mark("C: rpn backward + adam enqueued")
        # ---- phase D: re-predict with updated weights, propose, label, sample, head train
        n_head = 0
        for i, bp in enumerate(plans):
            s = batch[i]
            rp = rps[i]
            eng._run(rp["fwd"])
            R, Rn = eng.proposals(rp, overlap_thresh=0.7, max_boxes=300)
            rw, rh = new_img_size(s["width"], s["height"], C.img_size)       # rpn.py:189 recomputes it from the config
            P = eng.roi_targets_launch(R, Rn, self._gt(s), s["width"], s["height"], rw, rh)
            if next_batch is not None and i == nloc - 1 and not early:
                self._pre = self._launch_ab(next_batch, self._parity)      # keeps the GPU busy across the sync below
            mark("D: rpn re-predict + proposals + roi targets enqueued")
            P, cls, n = eng.roi_targets_finish(P)                             # the step's one host sync in this phase
            mark("D: roi classes on host")
            kept = np.nonzero(cls >= 0)[0]
            if n <= 0 or len(kept) == 0:                           # calc_iou -> None: the reference skips the head step
                self.skipped_head_steps += 1
                continue
            sel_k, _ = E.select_samples(cls[kept], eng.bg, C.n_rois)
            sel = kept[np.asarray(sel_k, dtype=np.int64)]
            mark("D: samples selected")
            if self.capture is not None:
                self.capture.append(dict(pred=rp["pred"].cpu().numpy().copy(), R=R[:n].cpu().numpy().copy(), keep=(cls >= 0).copy(),
                                         cls=cls.copy(), sel_kept=list(sel_k)))
            hp = eng._plan_head(C.n_rois, bp["fh"], bp["fw"], bp["F"])
            eng.pack_roi_batch(P, sel, hp)
            self._finish_head_update()               # deferred Adam #2 of the previous step: head weights are read next
            eng.head_forward(hp, training=True)
            eng.set_accumulate(hp["bwd"], n_head > 0, prezeroed=True)
            eng.head_backward(hp, accumulate=True, loss_out=self._det_l[n_head])
            n_head += 1
        if n_head > 0 or self.world > 1:
            self._finish_head_update()               # only still pending when every local image skipped its head phase
            if self.defer_head_update:
                self._head_pending = (allreduce_grad_arena_start(eng.head_arena.g, self.world, self.group), ntot)
            else:
                self._allreduce(eng.head_arena)
                eng.adam(eng.head_arena, grad_scale=1.0 / ntot)
                eng.refresh_head_shift()
        mark("D: head forward + backward + adam enqueued")
        self.last = (nloc, n_head)
        return self

    def losses(self):
        """Host copy of the last step's mean losses (one device sync)."""
        nloc, n_head = self.last
        r = self._rpn_l[:nloc].cpu().numpy().mean(0)
        d = self._det_l[:max(n_head, 1)].cpu().numpy().mean(0) if n_head else np.zeros(3, np.float32)
        return {"rpn_cls": float(r[0]), "rpn_regr": float(r[1]), "det_cls": float(d[0]), "det_regr": float(d[1]), "det_acc": float(d[2]),
                "n_head": n_head}
