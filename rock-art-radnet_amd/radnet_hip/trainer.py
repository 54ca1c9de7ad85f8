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
import collections
import contextlib
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


FORCE_COLLECTIVES = False      # rehearsal hook (bench.py RADNET_BENCH_REHEARSAL=nccl1): issue the exchanges on a 1-rank group too


def allreduce_grad_arena(flat, world, group=None):
    """Data-parallel exchange of one optimizer's flat gradient arena: SUM over ranks (RCCL over xGMI on the GPUs,
    gloo in the CPU tests).  Returns the factor Adam applies to the summed gradient so that the update uses the MEAN
    over all `world * images_per_rank` images (the caller divides by images_per_rank as well)."""
    if world > 1 or FORCE_COLLECTIVES:
        import torch.distributed as dist
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / world


def allreduce_grad_arena_start(flat, world, group=None):
    """The same exchange, asynchronous: returns a handle whose wait() orders the CURRENT stream after the reduction
    (None on a single rank).  The collective runs on the backend's own stream, beside whatever is enqueued next."""
    if world > 1 or FORCE_COLLECTIVES:
        import torch.distributed as dist
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)
    return None


class _EventWork:
    """Handle of an exchange issued on the native communicator's stream: wait() orders the CURRENT lane after it (an event wait,
    no host synchronisation) -- the surface of torch.distributed's async work handles, which the deferred update uses."""

    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)


class TrainStep:
    NBUF = 8        # buffer sets in rotation: this batch, up to four announced ones, whose head phase may still run; even,
                    # so that the sets alternate between the two prefetch lanes (and pairs of sets stay pairs)
    LOOKAHEAD = 4   # announced batches used (step(upcoming=[...]))

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
        # The two exchanges run on different lanes and belong to different batches (AR#1 of batch i+1 is issued before
        # AR#2 of batch i): on ONE communicator they would queue on its single internal stream, AR#1 of the next batch
        # behind an AR#2 that waits for a head backward -- the RPN chain would inherit the head lane's latency.  The head
        # exchange therefore gets its own communicator (created by all ranks here, in the same order).
        self.group_head = dist_group
        if world_size > 1 or FORCE_COLLECTIVES:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                self.group_head = dist.new_group(ranks=None if dist_group is None else dist.get_process_group_ranks(dist_group))
        self.defer_head_update = (world_size > 1) if defer_head_update is None else bool(defer_head_update)
        self._init_native_comm()
        self._head_pending = None        # (work handle, images in the global batch) of the exchange in flight
        self._head_done = {}             # buffer set -> event: end of the head phase that last read its feature map
        self._head_last = None           # event: end of the last head phase enqueued on the head lane
        self._lane_mode = False          # True between the first pipelined call and flush()
        self._queue = collections.deque()    # states of the announced batches, enqueued ahead (step(next_batch=...))
        self._slots = 0                  # batches started: buffer set = count % NBUF
        # pipelined mode (step(next_batch=...)): the engine's lanes run three batches' phases side by side
        self.side_prefetch = os.environ.get("RADNET_SIDE_PREFETCH", "1") == "1" and hasattr(eng, "lane")
        self.crop_ahead = os.environ.get("RADNET_NO_CROP_AHEAD", "0") != "1"       # pipelined step: RoI packing + crop-resize on the RPN lane
        # per-GPU mini-batch (BASELINE cfg 4) as ONE layer program: base / RPN / stage-5 GEMMs run once with the images
        # stacked along M (RADNET_BATCHED=0: image by image, the round-1 path)
        self.batched = os.environ.get("RADNET_BATCHED", "1") == "1" and getattr(eng, "supports_batched", False)
        # Pipelined mode, one image per batch: the FROZEN base forward of two consecutive announced batches runs as one
        # nb = 2 program (the base does not depend on anything the steps in between update), so its GEMMs see M doubled --
        # the same lever as the per-GPU mini-batch, applied across steps.  The feature maps are the same function of the same
        # weights; only the GEMM partitioning (and with it fp32 summation order) differs from the nb = 1 program.
        # Measured on MI355X (bench.py, 300 steps): 479.9 images/s with it, 481.7 without -- the pipelined step is bound by the
        # aggregate rate at which the chip retires GEMM work with four lanes resident, and there a launch with M doubled is
        # no faster per image (the per-launch overheads it removes were already hidden by the other lanes).  Hence OFF by
        # default (every announced batch gets its own nb = 1 base forward: bit-identical to back-to-back steps);
        # RADNET_STACK_BASE=1 switches it on.
        self.stack_base = os.environ.get("RADNET_STACK_BASE", "0") == "1" and getattr(eng, "supports_batched", False)
        self.skipped_head_steps = 0
        # per image of the last step: positive RoIs among the sampled ones (get_selected_samples' second value, what train.py:385-386
        # appends to rpn_accuracy_for_epoch), 0 for an image whose classifier step was skipped (train.py:378-380), None for an image
        # the labeller dropped (the reference's generator never yields it)
        self.last_n_pos = []
        self.last_took_head = []    # per image of the last step: its classifier step ran (n_pos may be 0 for one on background RoIs only)
        self._loss_log = None       # start_loss_log(): device rows [rpn_cls, rpn_regr, det_cls, det_regr, det_acc] per classifier step
        self._loss_log_n = 0
        self.dropped_images = 0     # images whose anchor labelling raised (reference: sample skipped, utils.py:461-465)
        self.on_drop = None         # optional callback(sample, exception); default: one line on stderr, like the reference's print
        self.last = None
        self.capture = None         # set to [] to record per-image intermediates (tests: stage-wise parity)
        self.host_marks = None      # set to [] to record (label, perf_counter) at the host-side phase boundaries
        dev = eng.dev
        self._rpn_l = torch.zeros(self.NBUF, 64, 2, dtype=torch.float32, device=dev)     # per-image loss slots (logging)
        self._det_l = torch.zeros(self.NBUF, 64, 3, dtype=torch.float32, device=dev)

    def _init_native_comm(self):
        """Round 4: both gradient exchanges through the library's own RCCL binding (radnet_allreduce_grads: one ncclAllReduce on a
        context's stream, ~5 us of host time) instead of torch.distributed (~45 us per call, its own streams and the event hops to
        and from them).  AR#1 (RPN arena, 19 MB) is issued IN LINE on the main lane -- the stream its producer (RPN backward) and
        consumer (Adam #1) run on.  AR#2 (head arena, 60 MB) gets a communicator on a stream of its own: one call after the head
        backward, ordered by two events, finished (event wait) just before the next step's head forward -- the deferred update of
        the torch path, without cutting the backward program into per-block buckets: the exchange has the whole RPN phase of the
        next step to hide under.  torch.distributed stays for the rendezvous (128-byte ids) and the run's bookkeeping collectives.
        Falls back to the torch path when the engine has no native library (recording engines of the CPU tests), the process
        group is not RCCL (gloo rehearsals: several ranks on one GPU cannot share an RCCL communicator) or RADNET_NATIVE_COMM=0."""
        self.native_comm = False
        if not (self.world > 1 or FORCE_COLLECTIVES) or os.environ.get("RADNET_NATIVE_COMM", "1") != "1":
            return
        eng = self.eng
        if not (hasattr(eng, "lib") and hasattr(eng, "ctx") and hasattr(eng.lib, "radnet_comm_init")) or not torch.cuda.is_available():
            return
        import torch.distributed as dist
        ready = dist.is_available() and dist.is_initialized()
        if self.world > 1 and (not ready or dist.get_backend(self.group) != "nccl"):
            return
        from . import lib as L
        from . import native as N
        try:
            rank = dist.get_rank(self.group) if ready else 0
            self._comm_stream = torch.cuda.Stream(device=eng.dev)
            self._comm_ctx = L.Context(eng.dev.index, stream_handle=self._comm_stream.cuda_stream)
            self._main_ctx = eng.ctx
            N.comm_init(eng, self.world, rank, dist if ready else None, self.group, ctx=self._main_ctx)
            N.comm_init(eng, self.world, rank, dist if ready else None, self.group, ctx=self._comm_ctx)
            self.native_comm = True
        except (L.RadnetError, RuntimeError) as e:      # (RuntimeError: the rendezvous broadcast itself failed)
            import sys
            sys.stderr.write("radnet: native RCCL exchange unavailable (%s); gradients travel through torch.distributed\n" % (e,))
        if ready and self.world > 1:                    # one path for the whole job: every rank uses the native exchange, or none does
            flag = torch.tensor([1 if self.native_comm else 0], dtype=torch.int32, device=eng.dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            self.native_comm = bool(int(flag.item()))

    def _allreduce(self, arena):
        if self.native_comm and arena is self.eng.rpn_arena:
            from . import native as N
            N.allreduce(self.eng, arena.g, ctx=self._main_ctx)      # in line on the main lane: after the RPN backward, before Adam #1
            return
        allreduce_grad_arena(arena.g, self.world, self.group_head if arena is self.eng.head_arena else self.group)

    def _native_head_exchange(self):
        """AR#2 on the head communicator's stream, after everything enqueued so far on the current lane (the head backward)."""
        from . import native as N
        ready = self.eng.mark()
        with torch.cuda.stream(self._comm_stream):
            self.eng.after(ready)
            N.allreduce(self.eng, self.eng.head_arena.g, ctx=self._comm_ctx)
            done = self.eng.mark()
        return _EventWork(done)

    def _finish_head_update(self):
        if self._head_pending is None:
            return
        works, ntot = self._head_pending
        self._head_pending = None
        for work in works if isinstance(works, (list, tuple)) else [works]:
            if work is not None:
                work.wait()
        self.eng.adam(self.eng.head_arena, grad_scale=1.0 / ntot)
        self.eng.refresh_head_shift()

    def start_loss_log(self, capacity):
        """Keep the five losses of every classifier step from here on in a device buffer (one row per image that took a head step,
        in order): what train.py:405-417 writes into losses[iter_num] -- read back ONCE per epoch (read_loss_log) instead of with a
        host synchronisation per iteration, so the pipelined step keeps running ahead of the host."""
        self._loss_log = torch.zeros(max(int(capacity), 1), 5, dtype=torch.float32, device=self.eng.dev)
        self._loss_log_n = 0

    def read_loss_log(self):
        """(n, 5) host array of the rows logged since start_loss_log (one device sync)."""
        if self._head_last is not None:
            self._head_last.synchronize()
        else:
            torch.cuda.synchronize(self.eng.dev)
        return self._loss_log[:self._loss_log_n].cpu().numpy() if self._loss_log is not None else np.zeros((0, 5), np.float32)

    def _log_losses(self, rpn_row, det_row):
        if self._loss_log is None or self._loss_log_n >= self._loss_log.shape[0]:
            return
        row = self._loss_log[self._loss_log_n]
        self.eng._copy(row[:2], rpn_row[:2])
        self.eng._copy(row[2:5], det_row[:3])
        self._loss_log_n += 1

    def flush(self):
        """End of a pipelined run: the main stream waits for the head lane, and a head update still in flight
        (deferred mode) is applied."""
        if self._head_last is not None:
            self.eng.after(self._head_last)
            self._head_last = None
        self._lane_mode = False
        self._finish_head_update()
        if getattr(self.eng, "_chain_plans", None):
            self.eng.check_chains()

    VAL_SLOT = 97          # buffer set of the validation pass (plans are keyed by slot: never one of the training step's)

    def validate(self, samples):
        """The reference's validation loop (train.py:478-561) over an iterable of samples, device-resident and forward only:
        per sample model_rpn.test_on_batch (RPN losses) -> predict_on_batch -> rpn_to_roi -> calc_iou -> get_selected_samples
        -> model_classifier.test_on_batch (detector losses, accuracy).  No optimizer step, no gradient arena touched, weights
        unchanged.  As in the reference a sample whose anchor labelling fails is skipped by the generator (utils.py:461-465) and
        a sample whose proposals overlap no box is skipped WHOLE -- its RPN losses are not recorded either (`continue` before the
        appends, train.py:508-509).  NumPy's global stream is drawn from exactly as the reference does (anchor subsampling in the
        validation generator, then get_selected_samples).  Returns the reference's record: means of the five losses, the mean
        number of positive RoIs ('mean_overlapping_bboxes', train.py:543) and the total (train.py:544); 'n' samples used."""
        eng = self.eng
        C = eng.C
        if self._lane_mode or self._head_last is not None:
            self.flush()
        samples = list(samples)
        n_max = max(len(samples), 1)
        rl = torch.zeros(n_max, 2, dtype=torch.float32, device=eng.dev)
        dl = torch.zeros(n_max, 3, dtype=torch.float32, device=eng.dev)
        used, n_pos_all, skipped, dropped = [], [], 0, 0
        for k, s in enumerate(samples):
            H, W = s["img"].shape[:2]
            tp = eng.anchor_targets_launch(self._gt(s), s["width"], s["height"], W, H, slot=self.VAL_SLOT)
            bp = eng.upload_image(s["img"], slot=self.VAL_SLOT)
            eng.base_forward(bp)
            rp = eng.rpn_forward(bp)
            try:
                ycls, yregr, _ = eng.anchor_targets_finish(tp)
            except KeyError as e:
                dropped += 1
                self._report_drop(s, e)
                continue
            eng.rpn_losses_only(rp, ycls, yregr, rl[k])
            R, Rn = eng.proposals(rp, overlap_thresh=0.7, max_boxes=300)
            rw, rh = new_img_size(s["width"], s["height"], C.img_size)
            P, cls, n = eng.roi_targets(R, Rn, self._gt(s), s["width"], s["height"], rw, rh)
            if self.capture is not None:
                self.capture.append(dict(pred=rp["pred"].cpu().numpy().copy(), R=R[:max(n, 0)].cpu().numpy().copy()))
            kept = np.nonzero(cls >= 0)[0]
            if n <= 0 or len(kept) == 0:
                skipped += 1
                continue
            sel_k, n_pos = E.select_samples(cls[kept], eng.bg, C.n_rois)
            sel = kept[np.asarray(sel_k, dtype=np.int64)]
            hp = eng._plan_head(C.n_rois, bp["fh"], bp["fw"], bp["F"])
            eng.pack_roi_batch(P, sel, hp)
            eng.head_losses_only(hp, dl[k])
            used.append(k)
            n_pos_all.append(n_pos)
        out = {"n": len(used), "skipped": skipped, "dropped": dropped}
        if used:
            r = rl.cpu().numpy()[used].astype(np.float64)
            d = dl.cpu().numpy()[used].astype(np.float64)
            out.update(rpn_cls=float(r[:, 0].mean()), rpn_regr=float(r[:, 1].mean()), det_cls=float(d[:, 0].mean()), det_regr=float(d[:, 1].mean()),
                       det_acc=float(d[:, 2].mean()), mean_overlapping_bboxes=float(sum(n_pos_all)) / len(n_pos_all))
            out["total"] = out["rpn_cls"] + out["rpn_regr"] + out["det_cls"] + out["det_regr"]
            out["per_sample"] = [dict(rpn_cls=float(r[i, 0]), rpn_regr=float(r[i, 1]), det_cls=float(d[i, 0]), det_regr=float(d[i, 1]), det_acc=float(d[i, 2]),
                                      n_pos=int(n_pos_all[i])) for i in range(len(used))]
        if getattr(eng, "_chain_plans", None):
            eng.check_chains()
        if hasattr(eng, "release_slot"):
            eng.release_slot(self.VAL_SLOT)       # the validation pass's buffer set does not outlive it (plans, buffers, graphs)
        return out

    def _gt(self, s):
        """Device copy of a sample's ground truth (cached on the sample: uploaded once)."""
        if "_gt_dev" not in s:
            cm = self.eng.C.class_mapping
            boxes = np.array([[b["x1"], b["y1"], b["x2"], b["y2"]] for b in s["bboxes"]], dtype=np.float64).reshape(-1, 4)
            isbg = np.array([1 if b["class"] == "bg" else 0 for b in s["bboxes"]], dtype=np.int32)
            cls = np.array([cm[b["class"]] for b in s["bboxes"]], dtype=np.int32)
            s["_gt_dev"] = self.eng.upload_gt(boxes, isbg, cls)
        return s["_gt_dev"]

    def _next_slot(self):
        self._slots += 1
        return (self._slots - 1) % self.NBUF

    def _launch_a(self, batch, slot):
        """Device half of phase A (anchor labelling + async copy of the label maps) and the weight-independent part of
        phase B (upload, frozen base forward) of every image of `batch`, into buffer set `slot`."""
        eng = self.eng
        nloc = len(batch)
        tp, plans = [], []
        for i, s in enumerate(batch):
            H, W = s["img"].shape[:2]
            tp.append(eng.anchor_targets_launch(self._gt(s), s["width"], s["height"], W, H, slot=slot * nloc + i))
        stacked = self.batched and nloc > 1 and len({s["img"].shape[:2] for s in batch}) == 1
        if stacked:
            bp = eng.upload_images([s["img"] for s in batch], slot=slot)
            eng.base_forward(bp)
            plans = [bp]
        else:
            for i, s in enumerate(batch):
                bp = eng.upload_image(s["img"], slot=slot * nloc + i)
                eng.base_forward(bp)
                plans.append(bp)
        return dict(batch=batch, tp=tp, plans=plans, rps=None, slot=slot, stacked=stacked)

    def _launch_b(self, st):
        """RPN forward of every image of a batch whose base forward is enqueued: reads the RPN weights."""
        st["rps"] = [self.eng.rpn_forward(bp) for bp in st["plans"]]      # stacked: one program over all images
        return st

    def _launch_ab(self, batch, slot):
        return self._launch_b(self._launch_a(batch, slot))

    def _launch_pair(self, batch_a, batch_b, slot_a, slot_b):
        """Phase A + frozen base forward of TWO announced one-image batches as one nb = 2 program (buffer set slot_a).
        Returns their two states; each sees its own feature map through a one-image view of the stacked plan."""
        eng = self.eng
        sts = []
        for batch, slot in ((batch_a, slot_a), (batch_b, slot_b)):
            s = batch[0]
            H, W = s["img"].shape[:2]
            tp = [eng.anchor_targets_launch(self._gt(s), s["width"], s["height"], W, H, slot=slot)]
            sts.append(dict(batch=batch, tp=tp, plans=None, rps=None, slot=slot, stacked=False))
        bp = eng.upload_images([batch_a[0]["img"], batch_b[0]["img"]], slot=slot_a)
        eng.base_forward(bp)
        for i, st in enumerate(sts):
            st["plans"] = [dict(F=bp["F"][i:i + 1], fh=bp["fh"], fw=bp["fw"], nb=1, x=bp["x"][i:i + 1], pair=bp)]
        return sts

    def _rpn_phase(self, st, ntot, mark):
        """Phases A (host half) + C + the device part of D for a batch whose forward passes are enqueued: label maps
        subsampled on the host RNG, RPN losses + backward over the local images, [all-reduce], Adam #1, re-prediction with
        the updated weights, proposals, RoI labelling (async copy of the class codes)."""
        eng = self.eng
        C = eng.C
        batch, plans, rps, slot = st["batch"], st["plans"], st["rps"], st["slot"]
        nloc = len(batch)
        # gradient arenas are zero here: allocated zeroed, and every Adam pass clears what it consumed
        # The labeller can fail exactly where the reference's does (KeyError while building the subsampling probabilities,
        # utils.py:789-797): the reference swallows that inside its generator (utils.py:461-465, `except: continue`) and the
        # image never reaches the model -- no optimizer effect, no draw from the RNG.  Same here: the image is dropped from
        # this batch before anything of it touches a gradient; the rest of the batch, and the head phase of the batch before
        # it (pipelined mode), go on.
        dead = st["dead"] = [False] * nloc
        stacked = st.get("stacked", False)
        n_live = 0
        for i in range(nloc):
            try:
                ycls, yregr, _ = eng.anchor_targets_finish(st["tp"][i])
            except KeyError as e:
                dead[i] = True
                self.dropped_images += 1
                self._report_drop(batch[i], e)
                if stacked:
                    eng.rpn_loss_image(rps[0], i, None, None)          # zero gradient rows for the dropped image
                continue
            mark("A: label maps on host, subsampled, packed")
            if stacked:
                eng.rpn_loss_image(rps[0], i, ycls, yregr, self._rpn_l[slot][i])
            else:
                eng.set_accumulate(rps[i]["bwd"], n_live > 0, prezeroed=True)
                eng.rpn_backward(rps[i], ycls, yregr, self._rpn_l[slot][i])
            n_live += 1
        if stacked and n_live > 0:
            eng.set_accumulate(rps[0]["bwd"], False, prezeroed=True)
            eng.rpn_backward_batched(rps[0])                           # ONE backward over all images' rows: the summed gradient
        st["roi"] = [None] * nloc
        if n_live == 0 and self.world == 1 and not FORCE_COLLECTIVES:
            st["adam1"] = eng.mark() if hasattr(eng, "mark") else None
            return                                     # whole batch dropped: no Adam step at all (the reference trains nothing)
        # world > 1: a rank whose images were all dropped still joins the exchange (zeros) and applies the same update as
        # its peers (replicas stay identical); the mean stays over the nominal batch, a dropped image contributing zero
        self._allreduce(eng.rpn_arena)
        eng.adam(eng.rpn_arena, grad_scale=1.0 / ntot)
        st["adam1"] = eng.mark() if hasattr(eng, "mark") else None
        mark("C: rpn backward + adam enqueued")
        if stacked:
            eng._run(rps[0].get("refwd", rps[0]["fwd"]))
        for i in range(nloc):
            if dead[i]:
                continue
            s = batch[i]
            if not stacked:
                eng._run(rps[i].get("refwd", rps[i]["fwd"]))      # same feature map as the first pass: no second input transform
            R, Rn = eng.proposals(eng.rpn_image(rps[0], i) if stacked else rps[i], overlap_thresh=0.7, max_boxes=300)
            rw, rh = new_img_size(s["width"], s["height"], C.img_size)       # rpn.py:189 recomputes it from the config
            st["roi"][i] = (R, eng.roi_targets_launch(R, Rn, self._gt(s), s["width"], s["height"], rw, rh, slot=slot * nloc + i))
        mark("D: rpn re-predict + proposals + roi targets enqueued")

    def _report_drop(self, sample, exc):
        if self.on_drop is not None:
            self.on_drop(sample, exc)
        else:
            import sys
            sys.stderr.write("radnet: anchor labelling failed (%s: %s); sample skipped as the reference's generator does\n"
                             % (type(exc).__name__, exc))

    def step(self, batch, next_batch=None, after_next=None, upcoming=None):
        """batch: list of dicts {img: uint8 BGR HWC (already at network size), bboxes: [{class,x1,x2,y1,y2}],
        width, height: source-frame size the boxes refer to}.  Losses of the step: self.losses().

        upcoming (optional, what a prefetching data loader knows: the batches of the next calls, in order -- up to three
        are used; next_batch / after_next name the first two individually) switches the PIPELINED mode on; the next calls
        must then pass these same objects, and flush() ends the run.  Three chains then share the GPU, each on its own
        stream and context (engine lanes), working on consecutive batches:
          side: a prefetch queue -- labelling kernels, upload and frozen base forward of every announced batch not yet
                started (it depends on nothing but a free buffer set, so it is always there to fill idle CUs)
          main: RPN forward, losses / backward, Adam #1, re-prediction, proposals and RoI labelling of batch i+1, as soon as
                batch i's RoI class codes have reached the host
          head: RoI batch, classifier forward / backward, Adam #2 of batch i
        They touch disjoint trainable weights (the base is frozen, the RPN and the classifier have their own optimizers),
        so every value is computed from exactly the operands the one-after-the-other order would use; no GEMM of this
        network fills 256 CUs for its whole duration (tails, split-K reductions, the one-workgroup NMS), co-scheduled
        chains fill those holes.  The host's order -- sample selection of this batch, then subsampling of the next
        batch's anchors -- and with it the order of draws from NumPy's global RNG is the reference's."""
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
        after = getattr(eng, "after", lambda ev: None)
        ahead = list(upcoming) if upcoming is not None else [b for b in (next_batch, after_next) if b is not None]
        ahead = ahead[:self.LOOKAHEAD]                 # lookahead; the further sets cover head phases still in flight
        pipelined = self.side_prefetch and bool(ahead) and not eng.ctx.timing_on
        # lane mode: from the first pipelined call until flush().  Base forwards and head phases then stay on their lanes
        # (contexts, recorded graphs) also in the calls that announce nothing -- the first and the last step of a run
        lanes = self.side_prefetch and not eng.ctx.timing_on and (pipelined or self._lane_mode)
        self._lane_mode = lanes
        q = self._queue                                # states of the coming batches, in call order
        st = q.popleft() if q and q[0]["batch"] is batch else None
        if st is None:
            if q and "roi" in q[0]:
                # the announced batch's RPN phase -- an optimizer step -- has already been applied: dropping it silently would
                # train the RPN on a batch whose classifier step never happens
                raise RuntimeError("TrainStep.step: the previous call announced a different next batch (pass the same object)")
            q.clear()                                  # not the announced batch: forward passes prepared for it are dropped
            slot = self._next_slot()
            if lanes:
                k = slot % getattr(eng, "n_side_lanes", 1)
                with eng.lane("side%d" % k if k else "side"):
                    after(self._head_done.get(slot))
                    st = self._launch_a(batch, slot)
                    st["done"] = eng.mark()
            else:
                st = self._launch_a(batch, slot)
        if "roi" not in st:                            # first step of a run, or the previous call was not pipelined
            after(st.get("done"))
            self._launch_b(st)
            self._rpn_phase(st, ntot, mark)
        for j in range(len(q)):                        # announced batches must be the ones prepared, in order
            if j >= len(ahead) or q[j]["batch"] is not ahead[j]:
                while len(q) > j:
                    q.pop()
                break
        if pipelined:
            j = len(q)
            # more batches will be announced later only if the caller filled the lookahead window: a lone pending batch then
            # waits one call for its partner (it is not needed before the call after next)
            more_coming = len(ahead) >= self.LOOKAHEAD
            while j < len(ahead):                      # prefetch queue: labelling kernels, upload, frozen base forward
                pairable = (self.stack_base and nloc == 1 and j + 1 < len(ahead) and len(ahead[j]) == 1 and len(ahead[j + 1]) == 1
                            and ahead[j][0]["img"].shape == ahead[j + 1][0]["img"].shape)
                if self.stack_base and nloc == 1 and not pairable and j + 1 >= len(ahead) and more_coming and j >= 2:
                    break                              # lone newcomer at the far end of the window: pair it up next call
                slot = self._next_slot()
                # consecutive batches (pairs) alternate lanes; a buffer set keeps its lane (and with it its context: one graph
                # per program)
                k = ((slot // 2) if pairable else slot) % getattr(eng, "n_side_lanes", 1)
                with eng.lane("side%d" % k if k else "side"):
                    after(self._head_done.get(slot))   # the head phase that last read this buffer set's feature map
                    if pairable:
                        slot_b = self._next_slot()
                        after(self._head_done.get(slot_b))
                        pair = self._launch_pair(ahead[j], ahead[j + 1], slot, slot_b)
                        ev = eng.mark()
                        for nb in pair:
                            nb["done"] = ev
                            q.append(nb)
                        j += 2
                    else:
                        nb = self._launch_a(ahead[j], slot)
                        nb["done"] = eng.mark()
                        q.append(nb)
                        j += 1
        elif not lanes:
            after(self._head_last)                     # the head phase below runs on the main lane
            if ahead and not q:
                q.append(self._launch_a(ahead[0], self._next_slot()))   # one lane: keeps the GPU busy across the sync below
        nxt = q[0] if q else None
        if pipelined and nxt.get("rps") is None:
            # The next batch's RPN forward needs only its base forward and the RPN weights of Adam #1 of THIS batch, both
            # enqueued long ago: it goes to the main lane now, ahead of the host's wait below, and is off the cycle
            # "RoI codes -> sample selection -> anchor subsampling -> RPN backward ... -> RoI codes" that bounds the step.
            after(nxt["done"])
            self._launch_b(nxt)
        mark("B: next batch's upload + base + rpn forward enqueued")
        # ---- phase D, host half: RoI class codes -> sample selection on the host RNG (the step's host sync)
        picks = []
        self.last_n_pos = [None] * nloc
        self.last_took_head = [False] * nloc
        for i in range(nloc):
            if st["roi"][i] is None:                               # dropped by the labeller (see _rpn_phase): not a head skip
                picks.append(None)
                continue
            R, P = st["roi"][i]
            P, cls, n = eng.roi_targets_finish(P)
            mark("D: roi classes on host")
            kept = np.nonzero(cls >= 0)[0]
            if n <= 0 or len(kept) == 0:                           # calc_iou -> None: the reference skips the head step
                self.skipped_head_steps += 1
                self.last_n_pos[i] = 0
                picks.append(None)
                continue
            sel_k, n_pos = E.select_samples(cls[kept], eng.bg, C.n_rois)
            self.last_n_pos[i] = int(n_pos)
            self.last_took_head[i] = True
            picks.append((P, kept[np.asarray(sel_k, dtype=np.int64)]))
            mark("D: samples selected")
            if self.capture is not None:
                view = eng.rpn_image(st["rps"][0], i) if st.get("stacked") else st["rps"][i]
                self.capture.append(dict(pred=view["pred"].cpu().numpy().copy(), R=R[:n].cpu().numpy().copy(),
                                         keep=(cls >= 0).copy(), cls=cls.copy(), sel_kept=list(sel_k)))
        # ---- pipelined: RoI packing + crop-resize of THIS batch on the RPN lane, now -- the lane is idle until the host has subsampled the
        # next batch's anchors (0.3 ms), and the classifier lane, the saturated one, starts at stage 5 (two launches fewer on it)
        crop_ev = None
        live_now = [i for i in range(nloc) if picks[i] is not None]
        if pipelined and self.crop_ahead and live_now and getattr(eng, "CROP_AHEAD", False):
            after(self._head_done.get(st["slot"]))      # the classifier phase that last used this buffer set's head plan
            if st.get("stacked"):
                bp0 = st["plans"][0]
                hp0 = eng._plan_head(nloc * C.n_rois, bp0["fh"], bp0["fw"], bp0["F"], groups=nloc)
                for i in range(nloc):
                    if picks[i] is None:
                        eng.idle_roi_group(hp0, i)
                    else:
                        eng.pack_roi_batch(picks[i][0], picks[i][1], hp0, group=i)
                eng.head_crop(hp0)
            else:
                for i in live_now:
                    bp0 = st["plans"][i]
                    hp0 = eng._plan_head(C.n_rois, bp0["fh"], bp0["fw"], bp0["F"])
                    eng.pack_roi_batch(picks[i][0], picks[i][1], hp0)
                    eng.head_crop(hp0)
            crop_ev = eng.mark()
        # ---- pipelined: the next batch's RPN phase goes first -- the host sync of the NEXT call waits for it
        if pipelined:
            self._rpn_phase(nxt, ntot, mark)           # its RPN forward is already enqueued (above)
        # ---- phase D, device half: classifier train step
        head_lane = (lambda: eng.lane("head")) if lanes else contextlib.nullcontext
        slot = st["slot"]
        n_head = 0
        det_rows = None                 # rows of the detector-loss slots that hold this step's losses (default: the first n_head)
        live = [i for i in range(nloc) if picks[i] is not None]
        # Bucketed exchange (deferred mode): during the LAST local image's backward each block's kernel gradients are
        # final as soon as that block is differentiated -- their all-reduce starts then, beside the blocks still to come
        # (res5c 18 MB, res5b 18 MB, res5a 24 MB); only the last slice and the small tail are left at the end.
        slices = eng.head_exchange_slices() if hasattr(eng, "head_exchange_slices") else None
        bucketed = self.defer_head_update and (self.world > 1 or FORCE_COLLECTIVES) and slices is not None and not self.native_comm
        works = []

        def exchange(lo, hi):
            works.append(allreduce_grad_arena_start(eng.head_arena.g[lo:hi], self.world, self.group_head))

        with head_lane():      # what it reads from the other lanes (feature map, RoI labels) is complete: the host waited
            after(crop_ev)
            crop_kw = {"cropped": True} if crop_ev is not None else {}
            if st.get("stacked") and live:
                # per-GPU mini-batch: all images' RoIs through stage 5 in ONE pass (GEMM M = nloc * n_rois * 49); losses per
                # image, an image without a classifier step contributes zero gradient rows
                bp = st["plans"][0]
                hp = eng._plan_head(nloc * C.n_rois, bp["fh"], bp["fw"], bp["F"], groups=nloc)
                slots = []
                for i in range(nloc):
                    if picks[i] is None:
                        if crop_ev is None:
                            eng.idle_roi_group(hp, i)
                        slots.append(None)
                    else:
                        if crop_ev is None:
                            eng.pack_roi_batch(picks[i][0], picks[i][1], hp, group=i)
                        slots.append(self._det_l[slot][i])            # row i = image i of the mini-batch
                        n_head += 1
                det_rows = [i for i in range(nloc) if picks[i] is not None]
                self._finish_head_update()               # deferred Adam #2 of the previous step: head weights are read next
                eng.head_forward(hp, training=True, loss_out=self._det_l[slot], group_live=[p is not None for p in picks], **crop_kw)
                eng.set_accumulate(hp["bwd"], False, prezeroed=True)
                eng.head_backward(hp, accumulate=True, loss_out=slots, on_part=exchange if bucketed else None)
                for i in det_rows:
                    self._log_losses(self._rpn_l[slot][i], self._det_l[slot][i])
            for i, bp in enumerate(st["plans"] if not st.get("stacked") else []):
                if picks[i] is None:
                    continue
                hp = eng._plan_head(C.n_rois, bp["fh"], bp["fw"], bp["F"])
                if crop_ev is None:
                    eng.pack_roi_batch(picks[i][0], picks[i][1], hp)
                self._finish_head_update()               # deferred Adam #2 of the previous step: head weights are read next
                eng.head_forward(hp, training=True, loss_out=self._det_l[slot][n_head], **crop_kw)
                eng.set_accumulate(hp["bwd"], n_head > 0, prezeroed=True)
                if bucketed and i == live[-1]:
                    eng.head_backward(hp, accumulate=True, loss_out=self._det_l[slot][n_head], on_part=exchange)
                else:
                    eng.head_backward(hp, accumulate=True, loss_out=self._det_l[slot][n_head])
                self._log_losses(self._rpn_l[slot][i], self._det_l[slot][n_head])
                n_head += 1
            if n_head > 0 or self.world > 1:
                self._finish_head_update()               # only still pending when every local image skipped its head phase
                if bucketed:
                    if not works:                        # every local image skipped its classifier step: the other ranks
                        for lo, hi in slices:            # still exchange slice by slice -- same collectives, zeros from here
                            exchange(lo, hi)
                    exchange(eng.head_bias_off, eng.head_arena.n)      # biases + dense heads
                    self._head_pending = (works, ntot)
                elif self.native_comm and self.defer_head_update:
                    self._head_pending = ([self._native_head_exchange()], ntot)
                elif self.native_comm:
                    self._native_head_exchange().wait()
                    eng.adam(eng.head_arena, grad_scale=1.0 / ntot)
                    eng.refresh_head_shift()
                elif self.defer_head_update:
                    self._head_pending = (allreduce_grad_arena_start(eng.head_arena.g, self.world, self.group_head), ntot)
                else:
                    self._allreduce(eng.head_arena)
                    eng.adam(eng.head_arena, grad_scale=1.0 / ntot)
                    eng.refresh_head_shift()
            if lanes:
                self._head_last = self._head_done[slot] = eng.mark()
        mark("D: head forward + backward + adam enqueued")
        self.last = (nloc, n_head, slot, list(st.get("dead", [False] * nloc)), det_rows if det_rows is not None else list(range(n_head)))
        return self

    def losses(self):
        """Host copy of the last step's mean losses (one device sync)."""
        nloc, n_head, slot, dead, det_rows = self.last
        if self._head_last is not None:
            self._head_last.synchronize()            # the head lane wrote the detector losses
        live = [i for i in range(nloc) if not dead[i]]
        r = self._rpn_l[slot][:nloc].cpu().numpy()[live].mean(0) if live else np.full(2, np.nan, np.float32)
        d = self._det_l[slot].cpu().numpy()[det_rows].mean(0) if n_head else np.zeros(3, np.float32)
        return {"rpn_cls": float(r[0]), "rpn_regr": float(r[1]), "det_cls": float(d[0]), "det_regr": float(d[1]), "det_acc": float(d[2]),
                "n_head": n_head, "dropped": nloc - len(live)}
