"""Composed C-ABI entry points driven from Python (include/radnet_hip.h, csrc/program.hip; SURVEY.md 8b "minimum exports").

The engine lays the reference's graph out as static layer programs; here whole phases are handed to the library in ONE call:

    rpn_forward(eng, bplan)            radnet_rpn_forward: base program + RPN program (model_rpn.predict, RADNet.py:552)
    predict_tile(eng, img_u8_dev, R)   radnet_predict_tile: preprocess .. proposals .. classifier outputs of the first R RoIs
    NativeTrainStep(eng).step(sample)  radnet_train_step: one reference iteration (train.py:288-402) on one image; the two
                                       NumPy-RNG-driven host steps arrive as callbacks (utils.py:785-813, train.py:93-129)
    comm_init / allreduce              radnet_comm_* / radnet_allreduce_grads: RCCL on the context's stream

The pipelined scheduler (trainer.TrainStep) issues the same kernels from its own lanes; radnet_train_step is the synchronous
single-stream form of the same iteration (tests compare the two bit for bit)."""
import ctypes as C

import numpy as np
import torch

from . import engine as E
from . import lib as L
from .trainer import new_img_size


def rpn_forward(eng, bplan):
    """model_rpn.predict as one library call; returns the RPN plan (outputs in rp['pred'])."""
    rp = eng._plan_rpn(bplan["fh"], bplan["fw"], bplan["F"], bplan.get("nb", 1))
    eng.ctx.check(eng.lib.radnet_rpn_forward(eng.ctx.h, eng._compile(bplan["ops"]), len(bplan["ops"]), eng._compile(rp["fwd"]), len(rp["fwd"])),
                  "radnet_rpn_forward")
    return rp


def _head_desc(eng, hp):
    if "live" not in hp and hasattr(eng, "sync_inference_filters"):
        eng.sync_inference_filters()            # inference plans run the classifier's 3x3 convs on transformed filters
    h = L.HeadDesc()
    h.fmap, h.fh, h.fw, h.fc = hp["F"].data_ptr(), hp["fh"], hp["fw"], 1024
    h.rois, h.n_rois, h.pool, h.pooled = hp["rois"].data_ptr(), hp["R"], 14, hp["pooled"].data_ptr()
    h.fwd_ops, h.n_fwd = eng._compile(hp["fwd"]), len(hp["fwd"])
    h.y5, h.hw, h.feat_c, h.feat = hp["y5"].data_ptr(), hp["hw"], 2048, hp["feat"].data_ptr()
    h.dense_w, h.dense_ld, h.dense_b, h.nc, h.nreg = eng.dense_w.data_ptr(), eng.dense_ld, eng.dense_b.data_ptr(), eng.nc, eng.nreg
    h.p_cls, h.p_regr = hp["pcls"].data_ptr(), hp["pregr"].data_ptr()
    h.tail_scratch = hp["tail_scratch"].data_ptr()
    return h


def predict_tile(eng, img_u8_dev, n_rois, overlap_thresh=0.7, max_boxes=300, slot=0):
    """One tile already on the device at network size (uint8 BGR HWC) -> (proposals int64 [n][4] host, P_cls, P_regr host) for
    the first n_rois proposals (padded with copies of the first, RADNet.py:115-122), everything in ONE radnet_predict_tile call."""
    H, W = int(img_u8_dev.shape[0]), int(img_u8_dev.shape[1])
    bp = eng._plan_base(1, H, W, slot)
    rp = eng._plan_rpn(bp["fh"], bp["fw"], bp["F"])
    hp = eng._plan_head(n_rois, bp["fh"], bp["fw"], bp["F"], training=False)
    head = _head_desc(eng, hp)
    t = L.TileDesc()
    t.img_u8, t.h, t.w, t.x = img_u8_dev.data_ptr(), H, W, bp["x"].data_ptr()
    t.base_ops, t.n_base = eng._compile(bp["ops"]), len(bp["ops"])
    t.rpn_ops, t.n_rpn = eng._compile(rp["fwd"]), len(rp["fwd"])
    t.pred, t.ld_pred, t.fh, t.fw, t.a = rp["pred"].data_ptr(), E.RPN_LD, rp["fh"], rp["fw"], eng.A
    t.anchor_wh_host = eng.anchor_wh.ctypes.data_as(C.POINTER(C.c_double))
    t.std_scaling, t.overlap_thresh, t.max_boxes = float(eng.C.std_scaling), float(overlap_thresh), int(max_boxes)
    t.R, t.Rp, t.Rn, t.prop_ws = rp["R"].data_ptr(), rp["Rp"].data_ptr(), rp["Rn"].data_ptr(), rp["prop_ws"].data_ptr()
    t.head = C.pointer(head)
    eng.ctx.check(eng.lib.radnet_predict_tile(eng.ctx.h, C.byref(t)), "radnet_predict_tile")
    n = int(rp["Rn"].cpu()[0])
    return rp["R"][:n].cpu().numpy(), hp["pcls"].cpu().numpy(), hp["pregr"].cpu().numpy()


class NativeTrainStep:
    """radnet_train_step: the reference iteration as one synchronous C call per image.  The NumPy global RNG is drawn from in
    the two callbacks, in the reference's order, through the very functions the pipelined scheduler uses (engine.subsample_valid,
    engine.select_samples)."""

    def __init__(self, eng, world=1):
        self.eng = eng
        self.world = world
        self.skipped_head_steps = 0
        self.dropped_images = 0
        self.capture = None
        self._descs = {}
        self._last = None
        bg, n_rois = int(eng.bg), int(eng.C.n_rois)

        self.force_drop = False        # tests: the labeller hook reports a failure / the RoI hook keeps nothing
        self.force_no_rois = False

        def subsample(user, valid_p, overlap_p, a, fh, fw):
            if self.force_drop:
                return -1
            valid = np.ctypeslib.as_array(valid_p, shape=(a, fh, fw))
            overlap = np.ctypeslib.as_array(overlap_p, shape=(a, fh, fw))
            try:
                return int(E.subsample_valid(valid, overlap))
            except KeyError:
                return -1

        def select(user, cls_p, n, sel_p, k):
            if self.force_no_rois:
                return 0
            cls = np.ctypeslib.as_array(cls_p, shape=(n,))
            kept = np.nonzero(cls >= 0)[0]
            if len(kept) == 0:
                return 0
            sel_k, _ = E.select_samples(cls[kept], bg, n_rois)
            out = np.ctypeslib.as_array(sel_p, shape=(k,))
            out[:] = kept[np.asarray(sel_k, dtype=np.int64)].astype(np.int32)
            if self.capture is not None:
                self.capture.append(dict(cls=cls.copy(), sel_kept=list(sel_k)))
            return k

        self._cb = (L.SUBSAMPLE_FN(subsample), L.SELECT_FN(select))         # kept alive with the object
        self.hooks = L.HostHooks(None, self._cb[0], self._cb[1])

    def _desc(self, sample):
        eng = self.eng
        Cc = eng.C
        H, W = sample["img"].shape[:2]
        key = (H, W)
        ent = self._descs.get(key)
        if ent is None:
            bp = eng._plan_base(1, H, W, 0)
            rp = eng._plan_rpn(bp["fh"], bp["fw"], bp["F"])
            hp = eng._plan_head(Cc.n_rois, bp["fh"], bp["fw"], bp["F"])
            fh, fw, A = bp["fh"], bp["fw"], eng.A
            dev = eng.dev
            buf = dict(raw=torch.empty(H, W, 3, dtype=torch.uint8, device=dev),
                       valid=torch.zeros(A, fh, fw, dtype=torch.uint8, device=dev), overlap=torch.zeros(A, fh, fw, dtype=torch.uint8, device=dev),
                       regr=torch.zeros(fh, fw, 4 * A, dtype=torch.float64, device=dev), best=torch.zeros(1024, 4, dtype=torch.int32, device=dev),
                       nfor=torch.zeros(1024, dtype=torch.int32, device=dev), scratch=torch.zeros(1024, dtype=torch.int64, device=dev),
                       h_valid=torch.zeros(A, fh, fw, dtype=torch.uint8).pin_memory(), h_overlap=torch.zeros(A, fh, fw, dtype=torch.uint8).pin_memory(),
                       ycls=torch.zeros(fh, fw, 2 * A, dtype=torch.float32, device=dev), yregr=torch.zeros(fh, fw, 8 * A, dtype=torch.float32, device=dev),
                       keep=torch.zeros(1024, dtype=torch.uint8, device=dev), cls=torch.zeros(1024, dtype=torch.int32, device=dev),
                       box=torch.zeros(1024, 4, dtype=torch.int32, device=dev), t=torch.zeros(1024, 4, dtype=torch.float64, device=dev),
                       iou=torch.zeros(1024, dtype=torch.float64, device=dev), h_cls=torch.zeros(1024, dtype=torch.int32).pin_memory(),
                       h_n=torch.zeros(1, dtype=torch.int32).pin_memory(), sel=torch.zeros(1024, dtype=torch.int32, device=dev),
                       h_sel=torch.zeros(1024, dtype=torch.int32).pin_memory())
            eng.set_accumulate(rp["bwd"], False, prezeroed=True)
            eng.set_accumulate(hp["bwd"], False, prezeroed=True)
            d = L.TrainDesc()
            d.h, d.w, d.x = H, W, bp["x"].data_ptr()
            d.img_u8 = buf["raw"].data_ptr()
            d.anchor_sizes_host, d.ns = eng.anchor_sizes.ctypes.data_as(C.POINTER(C.c_double)), len(eng.anchor_sizes)
            d.anchor_ratios_host, d.nr = eng.anchor_ratios.ctypes.data_as(C.POINTER(C.c_double)), len(eng.anchor_ratios)
            d.rpn_stride, d.rpn_max_overlap, d.std_scaling = float(Cc.rpn_stride), float(Cc.rpn_max_overlap), float(Cc.std_scaling)
            d.valid, d.overlap, d.regr = buf["valid"].data_ptr(), buf["overlap"].data_ptr(), buf["regr"].data_ptr()
            d.best_anchor, d.n_for_gt, d.at_scratch = buf["best"].data_ptr(), buf["nfor"].data_ptr(), buf["scratch"].data_ptr()
            d.h_valid, d.h_overlap = buf["h_valid"].data_ptr(), buf["h_overlap"].data_ptr()
            d.y_cls, d.y_regr = buf["ycls"].data_ptr(), buf["yregr"].data_ptr()
            d.base_ops, d.n_base = eng._compile(bp["ops"]), len(bp["ops"])
            d.rpn_fwd_ops, d.n_rpn_fwd = eng._compile(rp["fwd"]), len(rp["fwd"])
            d.rpn_bwd_ops, d.n_rpn_bwd = eng._compile(rp["bwd"]), len(rp["bwd"])
            d.rpn_refwd_ops, d.n_rpn_refwd = eng._compile(rp["refwd"]), len(rp["refwd"])
            d.pred, d.dz, d.ld_pred, d.fh, d.fw, d.a, d.bce_mode = rp["pred"].data_ptr(), rp["dz"].data_ptr(), E.RPN_LD, fh, fw, A, eng.bce_mode
            d.loss_scratch8, d.rpn_losses = eng.loss_scratch.data_ptr(), eng.rpn_losses.data_ptr()
            c1 = eng.convs["rpn_conv1"]
            if c1.wino_u is not None:
                d.wino_w, d.wino_c, d.wino_n, d.wino_ldw, d.wino_u = c1.weight.data_ptr(), c1.cin, c1.cout, c1.ldw, c1.wino_u.data_ptr()
                d.wino_form = c1.wino_m
            d.anchor_wh_host = eng.anchor_wh.ctypes.data_as(C.POINTER(C.c_double))
            d.overlap_thresh, d.max_boxes = 0.7, 300
            d.R, d.Rp, d.Rn, d.prop_ws = rp["R"].data_ptr(), rp["Rp"].data_ptr(), rp["Rn"].data_ptr(), rp["prop_ws"].data_ptr()
            d.min_overlap, d.max_overlap = float(Cc.classifier_min_overlap), float(Cc.classifier_max_overlap)
            d.regr_std_host4, d.bg_class = eng.regr_std.ctypes.data_as(C.POINTER(C.c_double)), int(eng.bg)
            d.keep, d.roi_cls, d.roi_box, d.roi_t, d.roi_iou = (buf[k].data_ptr() for k in ("keep", "cls", "box", "t", "iou"))
            d.h_roi_cls, d.h_n, d.sel, d.h_sel = buf["h_cls"].data_ptr(), buf["h_n"].data_ptr(), buf["sel"].data_ptr(), buf["h_sel"].data_ptr()
            head = _head_desc(eng, hp)
            d.head = C.pointer(head)
            d.y1, d.y2 = hp["y1"].data_ptr(), hp["y2"].data_ptr()
            d.head_dz, d.det_losses = hp["dz"].data_ptr(), eng.det_losses.data_ptr()
            d.dense_dw, d.dense_db, d.dfeat, d.g_last = eng.dense_dw.data_ptr(), eng.dense_db.data_ptr(), hp["dfeat"].data_ptr(), hp["g_last"].data_ptr()
            d.head_bwd_ops, d.n_head_bwd = eng._compile(hp["bwd"]), len(hp["bwd"])
            ha = eng.head_arena
            d.head_shift, d.head_scale, d.head_t0 = eng.head_shift.data_ptr(), eng.head_scale.data_ptr(), eng.head_t0.data_ptr()
            d.head_bias = ha.p[eng.head_bias_off:].data_ptr()
            d.head_bias_len = eng.head_bias_len
            d.tail_scratch = hp["tail_scratch"].data_ptr()
            wino = eng._head_adam_wino() if getattr(eng, "head_train_wino", False) else None
            if wino is not None:               # the classifier's training forward runs on Winograd filters: Adam #2 rewrites them
                d.head_wino, d.n_head_wino = C.cast(wino[0], C.c_void_p), wino[1]
            ent = self._descs[key] = (d, buf, head, bp, rp, hp)
        return ent

    def step(self, sample):
        """One image; returns self (losses())."""
        eng = self.eng
        d, buf, head, bp, rp, hp = self._desc(sample)
        Cc = eng.C
        cm = Cc.class_mapping
        boxes = np.array([[b["x1"], b["y1"], b["x2"], b["y2"]] for b in sample["bboxes"]], dtype=np.float64).reshape(-1, 4)
        isbg = np.array([1 if b["class"] == "bg" else 0 for b in sample["bboxes"]], dtype=np.int32)
        cls = np.array([cm[b["class"]] for b in sample["bboxes"]], dtype=np.int32)
        gt = eng.upload_gt(boxes, isbg, cls)
        self._gt_keep = gt
        d.g = gt["g"]
        d.gt = gt["boxes"].data_ptr() if gt["g"] else None
        d.gt_is_bg = gt["isbg"].data_ptr() if gt["g"] else None
        d.gt_cls = gt["cls"].data_ptr() if gt["g"] else None
        d.width, d.height = int(sample["width"]), int(sample["height"])
        d.rw, d.rh = new_img_size(sample["width"], sample["height"], Cc.img_size)
        buf["raw"].copy_(torch.from_numpy(np.ascontiguousarray(sample["img"])))
        for arena, opt in ((eng.rpn_arena, d.rpn_opt), (eng.head_arena, d.head_opt)):
            opt.p, opt.g, opt.m, opt.v, opt.n, opt.lr = arena.p.data_ptr(), arena.g.data_ptr(), arena.m.data_ptr(), arena.v.data_ptr(), arena.n, eng.lr
        d.rpn_opt.t = eng.rpn_arena.t + 1
        d.head_opt.t = eng.head_arena.t + 1
        d.world = self.world
        losses = (C.c_float * 5)()
        took = C.c_int32(0)
        n_drop = self.dropped_images
        eng.ctx.check(eng.lib.radnet_train_step(eng.ctx.h, C.byref(d), C.byref(self.hooks), losses, C.byref(took)), "radnet_train_step")
        if took.value < 0:                       # dropped by the labeller hook: nothing of this image was trained on
            self.dropped_images += 1
        elif not took.value:
            self.skipped_head_steps += 1
        # world > 1: the rank joined both exchanges and applied both optimizer steps whatever its own image did (program.hip)
        if took.value >= 0 or self.world > 1:
            eng.rpn_arena.t += 1
        if took.value > 0 or self.world > 1:
            eng.head_arena.t += 1
            # the inference head plans keep Winograd-transformed copies of the classifier's 3x3 filters (sync_inference_filters)
            eng._inference_filters_stale = True
        self._last = dict(rpn_cls=float(losses[0]), rpn_regr=float(losses[1]), det_cls=float(losses[2]), det_regr=float(losses[3]),
                          det_acc=float(losses[4]), n_head=max(int(took.value), 0), dropped=self.dropped_images - n_drop)
        if self.capture is not None and self.capture:
            n = int(rp["Rn"].cpu()[0])
            self.capture[-1].update(pred=rp["pred"].cpu().numpy().copy(), R=rp["R"][:n].cpu().numpy().copy())
        return self

    def losses(self):
        return dict(self._last)


def comm_init(eng, world, rank, dist=None, group=None, ctx=None):
    """RCCL communicator on the engine's main context (or on `ctx`, a context of the engine's library): rank 0 draws the id,
    torch.distributed (any backend -- the rendezvous the job already has) carries its 128 bytes to the other ranks.
    world == 1 needs no rendezvous."""
    ctx = ctx or eng.ctx
    ident = C.create_string_buffer(128)
    if rank == 0:
        rc = eng.lib.radnet_comm_unique_id(ident)
        if rc != 0:
            raise L.RadnetError("radnet_comm_unique_id failed (%d): librccl not available" % rc)
    if world > 1:
        t = torch.frombuffer(bytearray(ident.raw), dtype=torch.uint8).clone()
        if dist.get_backend(group) == "nccl":
            t = t.to(eng.dev)
        dist.broadcast(t, src=0, group=group)
        ident = C.create_string_buffer(bytes(t.cpu().numpy().tobytes()), 128)
    ctx.check(eng.lib.radnet_comm_init(ctx.h, world, rank, ident), "radnet_comm_init")


def allreduce(eng, flat, ctx=None):
    """In-place fp32 sum over the communicator of `ctx` (default: the engine's current context), enqueued on that context's stream."""
    ctx = ctx or eng.ctx
    ctx.check(eng.lib.radnet_allreduce_grads(ctx.h, flat.data_ptr(), C.c_int64(flat.numel())), "radnet_allreduce_grads")
