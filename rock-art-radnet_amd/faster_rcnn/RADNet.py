"""Detector facade with the reference's construct / predict surface (faster_rcnn/RADNet.py).

    RADNet(C, model_rpn, model_detector, preprocess_func)              RADNet.py:33-41
    .predict(images) -> [{'class','prob','x1','y1','x2','y2'}, ...]     RADNet.py:502-718
    .predict_from_path(path)                                           RADNet.py:482-500
    .format_img / .apply_spatial_pyramid_pooling / .final_nms / .get_real_coordinates
    load_radnet(config_path)                                           RADNet.py:721-775

The model objects are duck-typed exactly as in the reference (anything with .predict works, which is how the
golden tests drive this class with closed-form fake models); load_radnet() builds them on the HIP engine.
NMS runs in libradnet_hip.so through faster_rcnn.rpn; the tile resize runs on the device (radnet_resize_bicubic_u8).
"""
import pickle
import sys

import numpy as np

from . import rpn
from .utils import get_new_img_size  # noqa: F401  (re-exported like the reference's star import)


def _spans(length, tile, step):
    """Sliding-window spans along one axis (RADNet.py:519-535): starts every `step`, windows that fit, plus one
    window flush with the far edge; duplicates removed, sorted."""
    starts = np.arange(0, length, step)
    ends = starts + tile
    ok = ends <= length
    pairs = {(int(s), int(e)) for s, e in zip(starts[ok], ends[ok])}
    pairs.add((max(0, length - tile), int(length)))
    return sorted(pairs)


class RADNet():

    device_resident = True      # engine-backed models: keep tiles on the device between the stages (see _detect)

    def __init__(self, C, model_rpn, model_detector, preprocess_func):
        self.is_object_threshold = 0.5
        self.bbox_threshold = 0.7
        self.C = C
        self.model_rpn = model_rpn
        self.model_detector = model_detector
        self.preprocess_func = preprocess_func
        self.class_mapping = {v: k for k, v in C.class_mapping.items()}

    # ---- geometry helpers ------------------------------------------------------------------------------------
    def get_real_coordinates(self, ratio, x1, y1, x2, y2):
        """Back to source-image pixels: floor-division by the resize ratio, then round (RADNet.py:44-51)."""
        return tuple(int(round(v // ratio)) for v in (x1, y1, x2, y2))

    def format_img_size(self, img, keep_on_device=False, ctx=None):
        """Short side -> C.img_size, long side truncated (RADNet.py:53-74); bicubic resize on the device.
        keep_on_device: return the resized uint8 image as a device tensor (the device-resident tile path)."""
        side = float(self.C.img_size)
        height, width = img.shape[:2]
        if width <= height:
            ratio = side / width
            new_w, new_h = int(side), int(ratio * height)
        else:
            ratio = side / height
            new_w, new_h = int(ratio * width), int(side)
        if keep_on_device:
            return resize_cubic(img, new_w, new_h, to_host=False, ctx=ctx), ratio
        if (new_h, new_w) != (height, width):
            img = resize_cubic(img, new_w, new_h)
        return img, ratio

    def format_img_channels(self, img):
        img = img[:, :, (2, 1, 0)].astype(np.float32)          # BGR -> RGB (RADNet.py:83-84)
        return self.preprocess_func(np.expand_dims(img, axis=0))

    def format_img(self, img):
        img, ratio = self.format_img_size(img)
        return self.format_img_channels(img), ratio

    # ---- classifier head over the proposals -----------------------------------------------------------------------
    def apply_spatial_pyramid_pooling(self, R, feature_map):
        """RADNet.py:104-154.  R (n,4) xywh in feature-map units.  The RoIs go through the detector n_rois at a
        time, the last chunk padded with copies of its first RoI; confident non-background RoIs are decoded with
        their class's deltas and scaled to resized-image pixels."""
        chunks = self._spp_chunks(R)
        # The reference's Keras detector is built for exactly n_rois RoIs, hence its 15 calls per tile.  The HIP
        # detector takes any count (every RoI is independent under TimeDistributed), so all chunks -- padding rows
        # included, they are decoded too -- go through ONE head pass: GEMM M = 14 700 instead of 15 x 980.
        k = self.C.n_rois
        if chunks and getattr(self.model_detector, "accepts_any_roi_count", False):
            pc, pr = self.model_detector.predict([feature_map, np.concatenate(chunks, axis=1)])
            outs = [(pc[:, i * k:(i + 1) * k], pr[:, i * k:(i + 1) * k]) for i in range(len(chunks))]
        else:
            outs = [tuple(self.model_detector.predict([feature_map, ROIs])) for ROIs in chunks]
        return self._spp_decode(chunks, outs)

    def _spp_chunks(self, R):
        """RoIs n_rois at a time, the last chunk padded with copies of its first RoI (RADNet.py:110-122)."""
        k = self.C.n_rois
        chunks = []
        for start in range(0, R.shape[0], k):
            chunk = R[start:start + k, :]
            if chunk.shape[0] < k:
                padded = np.zeros((k, chunk.shape[1])).astype(chunk.dtype)
                padded[:chunk.shape[0]] = chunk
                padded[chunk.shape[0]:] = chunk[0]
                chunk = padded
            chunks.append(np.expand_dims(chunk, axis=0))
        return chunks

    def _spp_decode(self, chunks, outs):
        """Confident non-background RoIs decoded with their class's deltas, in resized-image pixels (RADNet.py:124-154)."""
        C = self.C
        stride = C.rpn_stride
        std = C.classifier_regr_std
        bboxes, probs = {}, {}
        for ROIs, (P_cls, P_regr) in zip(chunks, outs):
            for ii in range(P_cls.shape[1]):
                scores = P_cls[0, ii, :]
                best = int(np.argmax(scores))
                if np.max(scores) < self.bbox_threshold or best == P_cls.shape[2] - 1:
                    continue
                name = self.class_mapping[best]
                x, y, w, h = ROIs[0, ii, :]
                try:
                    tx, ty, tw, th = P_regr[0, ii, 4 * best:4 * (best + 1)]
                    x, y, w, h = rpn.apply_regr(x, y, w, h, tx / std[0], ty / std[1], tw / std[2], th / std[3])
                except Exception:
                    pass
                bboxes.setdefault(name, []).append([stride * x, stride * y, stride * (x + w), stride * (y + h)])
                probs.setdefault(name, []).append(np.max(scores))
        return bboxes, probs

    def final_nms(self, boxes, probs, obj_avg_threshold=0.2, obj_confidence_threshold=0.8, n_obj_avg=5):
        """RADNet.py:156-240: greedy clustering around the current best box (IoU > obj_avg_threshold); each cluster
        becomes the mean of its members above obj_confidence_threshold, or of its n_obj_avg best members."""
        if len(boxes) == 0:
            return []
        x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
        np.testing.assert_array_less(x1, x2)
        np.testing.assert_array_less(y1, y2)
        if boxes.dtype.kind == "i":
            boxes = boxes.astype("float")
        area = (x2 - x1) * (y2 - y1)
        order = np.argsort(probs, kind="stable")
        clusters = []
        while len(order) > 0:
            last = len(order) - 1
            top, rest = order[last], order[:last]
            iw = np.maximum(0, np.minimum(x2[top], x2[rest]) - np.maximum(x1[top], x1[rest]))
            ih = np.maximum(0, np.minimum(y2[top], y2[rest]) - np.maximum(y1[top], y1[rest]))
            inter = iw * ih
            overlap = inter / (area[top] + area[rest] - inter + 1e-6)
            members = np.concatenate((np.where(overlap > obj_avg_threshold)[0], [last]))
            mp = probs[order[members]]
            if mp.max() < obj_confidence_threshold:
                chosen = order[members][-n_obj_avg:]
            else:
                chosen = order[members][np.nonzero(mp > obj_confidence_threshold)[0]]
            clusters.append(chosen)
            order = np.delete(order, members)
        new_boxes = [np.rint(boxes[c].mean(axis=0)).astype('int') for c in clusters]
        new_probs = [probs[c].mean() for c in clusters]
        return np.array(new_boxes), np.array(new_probs)

    # ---- inference ---------------------------------------------------------------------------------------------------
    def _detect(self, img):
        """One network pass on an image or tile: {class: (boxes in source px, probs)} after the per-class NMS 0.2.
        With the engine-backed models the tile stays on the device from the resize to the classifier outputs (resize ->
        preprocess -> base -> RPN -> decode/sort/NMS -> RoI crop-resize -> classifier): PCIe carries the source tile in and
        ~40 KB of proposals and class scores out.  `device_resident = False` forces the NumPy-facing calls the reference
        makes (RADNet.py:540-560); both give the same detections (same kernels), tests compare them."""
        if self.device_resident and hasattr(self.model_rpn, "propose_device"):
            img_dev, ratio = self.format_img_size(img, keep_on_device=True)
            R, F = self.model_rpn.propose_device(img_dev, overlap_thresh=0.7)
        else:
            X, ratio = self.format_img(img)
            Y1, Y2, F = self.model_rpn.predict(X)
            R = rpn.rpn_to_roi(Y1, Y2, self.C, overlap_thresh=0.7)
        return self._finish_detect(R, F, ratio)

    def _finish_detect(self, R, F, ratio):
        R[:, 2] -= R[:, 0]
        R[:, 3] -= R[:, 1]
        return self._per_class_nms(*self.apply_spatial_pyramid_pooling(R, F), ratio)

    def _per_class_nms(self, bboxes, probs, ratio):
        """NMS 0.2 within each class, boxes back in source pixels (RADNet.py:562-575)."""
        out = {}
        for key in bboxes:
            nb, npr = rpn.non_max_suppression_fast(np.array(bboxes[key]), np.array(probs[key]), overlap_thresh=0.2)
            real = [self.get_real_coordinates(ratio, *nb[j, :]) for j in range(nb.shape[0])]
            out[key] = (real, [npr[j] for j in range(nb.shape[0])])
        return out

    def _detect_all(self, tiles):
        """_detect over a list of tiles, in order.  With the engine-backed models two tiles are in flight: while the
        classifier works on tile j (main lane), tile j+1 is uploaded, resized and run through the base network, the RPN and
        the proposal kernels on the engine's side lane, in the other buffer set.  Same kernels, same results as _detect."""
        eng = getattr(getattr(self.model_rpn, "_s", None), "eng", None)
        if not (self.device_resident and hasattr(self.model_rpn, "propose_launch") and eng is not None and hasattr(eng, "lane") and len(tiles) > 1):
            return [self._detect(t) for t in tiles]

        def launch(j, head_done):
            with eng.lane("side"):
                eng.after(head_done)                     # the classifier pass that last read this buffer set
                img_dev, ratio = self.format_img_size(tiles[j], keep_on_device=True, ctx=eng.ctx)
                h = self.model_rpn.propose_launch(img_dev, overlap_thresh=0.7, slot=j % 2)
                return h, ratio, eng.mark()

        k = self.C.n_rois
        out, done = [], [None, None]                     # per buffer set: event after the classifier pass that read it
        nxt = launch(0, None)
        for j in range(len(tiles)):
            (h, ratio, ready), nxt = nxt, None
            eng.after(ready)
            R, F = self.model_rpn.propose_finish(h)
            R[:, 2] -= R[:, 0]
            R[:, 3] -= R[:, 1]
            chunks = self._spp_chunks(R)
            hp = self.model_detector.predict_launch([F, np.concatenate(chunks, axis=1)])      # enqueued, not waited for
            done[j % 2] = eng.mark()
            if j + 1 < len(tiles):                       # the next tile's upload .. proposals run beside this classifier pass
                nxt = launch(j + 1, done[(j + 1) % 2])
            pc, pr = self.model_detector.predict_finish(hp)
            bboxes, probs = self._spp_decode(chunks, [(pc[:, i * k:(i + 1) * k], pr[:, i * k:(i + 1) * k]) for i in range(len(chunks))])
            out.append(self._per_class_nms(bboxes, probs, ratio))
        return out

    # ---- tiles over the ranks of a data-parallel job (inference) ------------------------------------------
    def set_distributed(self, group=None, enabled=True):
        """Shard the tiles of every image round-robin over the ranks of torch.distributed's `group` (default: the world):
        each rank runs the network passes of its tiles, the per-tile detections (a few hundred boxes: KB) are exchanged with
        all_gather_object, and every rank then runs the merge tail (final_nms per image, NMS across images) on the tiles IN
        THEIR ORIGINAL ORDER -- so every rank returns exactly what a single process returns.  The reference is single-process
        (predict.py:56-122); this is its tile loop (RADNet.py:540-600) spread over GPUs, nothing else changes."""
        self._dist_group = group
        self._dist_on = bool(enabled)

    def _detect_sharded(self, work):
        import torch.distributed as dist
        if not getattr(self, "_dist_on", False) or not (dist.is_available() and dist.is_initialized()):
            return self._detect_all(work)
        group = getattr(self, "_dist_group", None)
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        if world == 1:
            return self._detect_all(work)
        mine = list(range(rank, len(work), world))
        local = self._detect_all([work[j] for j in mine]) if mine else []
        # plain Python containers on the wire (class name -> ([[x1, y1, x2, y2], ...], [prob, ...]))
        payload = [(j, {k: ([[int(v) for v in b] for b in real], [float(p) for p in pr]) for k, (real, pr) in det.items()}) for j, det in zip(mine, local)]
        gathered = [None] * world
        dist.all_gather_object(gathered, payload, group=group)
        out = [None] * len(work)
        for part in gathered:
            for j, det in part:
                out[j] = {k: ([tuple(b) for b in real], [np.float32(p) for p in pr]) for k, (real, pr) in det.items()}
        assert all(o is not None for o in out)
        return out

    def predict(self, images):
        """RADNet.py:502-718: tile -> RPN -> NMS -> RoI crop-resize -> classifier -> per-class NMS, box-averaging
        merge per image, then NMS 0.4 across images."""
        C = self.C
        all_boxes, all_probs = {}, {}
        for img in images:
            boxes_img, probs_img = {}, {}

            def collect(det, ox, oy):
                for key, (real, pr) in det.items():
                    for (rx1, ry1, rx2, ry2), p in zip(real, pr):
                        boxes_img.setdefault(key, []).append([ox + rx1, oy + ry1, ox + rx2, oy + ry2])
                        probs_img.setdefault(key, []).append(p)

            if C.max_n_tiles_train > 0:                 # the reference gates tiling on this training knob (RADNet.py:511)
                h, w = img.shape[:2]
                spans = [(tx0, ty0, tx1, ty1) for (ty0, ty1) in _spans(h, C.tile_size, C.tile_overlap) for (tx0, tx1) in _spans(w, C.tile_size, C.tile_overlap)]
            else:
                spans = []
            work = [np.copy(img[ty0:ty1, tx0:tx1, :]) for (tx0, ty0, tx1, ty1) in spans] + ([img] if C.include_full_img else [])
            offs = [(tx0, ty0) for (tx0, ty0, tx1, ty1) in spans] + ([(0, 0)] if C.include_full_img else [])
            for det, (ox, oy) in zip(self._detect_sharded(work), offs):
                collect(det, ox, oy)
            for key in boxes_img:
                nb, npr = self.final_nms(np.array(boxes_img[key]), np.array(probs_img[key]), obj_avg_threshold=0.2,
                                         obj_confidence_threshold=0.8, n_obj_avg=5)
                for j in range(nb.shape[0]):
                    all_boxes.setdefault(key, []).append(list(nb[j, :]))
                    all_probs.setdefault(key, []).append(npr[j])
        dets = []
        for key in all_boxes:
            nb, npr = rpn.non_max_suppression_fast(np.array(all_boxes[key]), np.array(all_probs[key]), overlap_thresh=0.4)
            for j in range(nb.shape[0]):
                x1, y1, x2, y2 = nb[j, :]
                dets.append({'class': key, 'prob': npr[j], 'x1': x1, 'y1': y1, 'x2': x2, 'y2': y2})
        return dets

    def predict_from_path(self, img_path):
        """RADNet.py:482-500 (needs an image decoder; OpenCV is the reference's and is absent here)."""
        try:
            import cv2  # noqa: F401
        except ImportError as e:
            raise NotImplementedError("predict_from_path needs OpenCV to decode images; pass decoded BGR arrays to predict()") from e
        from .utils_io import get_image
        C = self.C
        types = C.img_types if C.use_img_type else [C.img_types[0]]
        return self.predict([get_image(img_path, [t], random_type=False) for t in types])


def resize_cubic(img, new_w, new_h, to_host=True, ctx=None):
    """cv2.resize(img, (new_w, new_h), interpolation=cv2.INTER_CUBIC) on the device (uint8 HWC).  to_host=False: the
    result stays a device tensor (an image already at the target size is just uploaded)."""
    import torch
    from radnet_hip import runtime as rt
    own = ctx is None
    ctx = rt.default_context() if own else ctx                  # a lane's context: the kernel goes to that lane's stream
    side = rt.thread_stream() if own else None                  # a worker thread resizes on its own stream (BackgroundFeed)
    import contextlib
    with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
        src = torch.from_numpy(np.ascontiguousarray(img, dtype=np.uint8)).cuda()
        if not to_host and (new_h, new_w) == tuple(img.shape[:2]):
            out = src
        else:
            out = torch.empty((new_h, new_w, img.shape[2]), dtype=torch.uint8, device="cuda")
            ctx.call("radnet_resize_bicubic_u8", src, img.shape[0], img.shape[1], out, new_h, new_w, img.shape[2])
            if to_host:
                return out.cpu().numpy()
    if side is not None:
        # a device result made on the worker's stream: its consumer's stream waits, and the caching allocator is told that the
        # memory is in use there too (it was allocated under `side`; without this it could be handed out again while the
        # consumer still reads it)
        torch.cuda.current_stream().wait_stream(side)
        out.record_stream(torch.cuda.current_stream())
    return out


def warp_affine_device(img, mat, dsize, ctx=None):
    """cv2.warpAffine(img, mat, dsize) (bilinear, constant border 0) on the device: augmentation.warp_affine_u8's arithmetic,
    bit for bit (tests/test_gpu_resize.py), about a hundred times faster than the NumPy form for a 300x300 tile.  uint8 HWC in,
    uint8 HWC (NumPy) out."""
    import contextlib
    import torch
    from radnet_hip import runtime as rt
    from .augmentation import warp_tables
    own = ctx is None
    ctx = rt.default_context() if own else ctx
    side = rt.thread_stream() if own else None                  # a worker thread warps on its own stream (BackgroundFeed)
    dw, dh = int(dsize[0]), int(dsize[1])
    adelta, bdelta, x0, y0 = warp_tables(mat, dsize)
    img = np.ascontiguousarray(img, dtype=np.uint8)
    ch = img.size // (img.shape[0] * img.shape[1])
    with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
        src = torch.from_numpy(img).cuda()
        col = torch.from_numpy(np.concatenate([adelta, bdelta]).astype(np.int32)).cuda()
        row = torch.from_numpy(np.concatenate([x0, y0]).astype(np.int32)).cuda()
        dst = torch.empty((dh, dw) + img.shape[2:], dtype=torch.uint8, device="cuda")
        ctx.call("radnet_warp_affine_u8", src, img.shape[0], img.shape[1], ch, dst, dh, dw, col, row)
        return dst.cpu().numpy()


class _ConfigUnpickler(pickle.Unpickler):
    """config.pickle holds a plain attribute bag (config.py:5-133; train.py:176-180 dumps it): only that class and builtin
    containers / scalars are admitted, so a crafted file cannot name arbitrary callables (the reference's bare
    pickle.load, RADNet.py:724, would execute them)."""
    _BUILTINS = {"dict", "list", "tuple", "set", "frozenset", "int", "float", "bool", "str", "bytes", "complex", "slice", "range"}

    def find_class(self, module, name):
        if (module, name) == ("faster_rcnn.config", "Config"):
            from .config import Config
            return Config
        if module == "builtins" and name in self._BUILTINS:
            import builtins
            return getattr(builtins, name)
        if (module, name) == ("collections", "OrderedDict"):
            import collections
            return collections.OrderedDict
        raise pickle.UnpicklingError("config pickle refers to %s.%s: only faster_rcnn.config.Config and builtin containers are loaded"
                                     % (module, name))


def load_radnet(config_path, device_index=0):
    """RADNet.py:721-775: unpickle the Config, build the RPN (3 outputs) and detector models, load C.weights_path."""
    from . import models
    with open(config_path, 'rb') as f:
        C = _ConfigUnpickler(f).load()
    if C.network == 'resnet50':
        from .base_models import resnet50 as base_model
    elif C.network == 'vgg16':
        from .base_models import vgg16 as base_model
    else:
        print('Not a valid base model!')
        sys.exit(1)
    _, _, model_all, model_rpn, model_detector = models.build_models(C, device_index=device_index, workload="predict")      # loads no train-step launch-shape table
    model_all.load_weights(str(C.weights_path).replace('\\', '/'), by_name=True)
    return RADNet(C, model_rpn, model_detector, base_model.preprocess)
