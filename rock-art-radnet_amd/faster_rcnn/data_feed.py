"""Host data feed for the train step: the input side of utils.get_tile_generator (utils.py:310-552)
without its label computation -- tiles, class balancing, box clipping, resize -- yielding the samples
`radnet_hip.trainer.TrainStep.step` takes ({img: uint8 BGR at network size, bboxes, width, height}); the anchor labels
the reference computes inside its generator (calc_region_props) are computed on the device by the step itself.

Same control flow and the same draws from the random stream as the reference, in the same order:
  np.random.shuffle(data) per epoch (train mode); per image the class-balance test; the tile grid; per tile
  [np.random.choice over the image types when C.use_img_type] -> np.random.randint(0, #tiles left) -> clip boxes to the
  tile (drop those left with less than C.tile_bbox_clip_threshold of their area) -> tile coordinates (int / ceil) ->
  skip empty tiles and tiles without the class whose turn it is -> resize to get_new_img_size (bicubic, on the device);
  then the full image when C.include_full_img.
Augmentation (augmentation.py:85-533) is `faster_rcnn/augmentation.py` of this package: every switch of
the reference's Config (flips, 90-degree and +-3-degree rotation, shear, brightness, the noise / contrast family) with the
reference's draws in its order; which parts are pinned by the reference's own outputs and which restate OpenCV /
scikit-image semantics unpinned is listed in that module's header.  Images are decoded by the caller
(`load_image(img_data, img_type) -> uint8 BGR HWC`): OpenCV, which the reference decodes with, is not part of this build.

rng: None = NumPy's global stream, i.e. exactly the reference's interleaving with the step's own draws when samples are
pulled one per step; pass a RandomState to pull samples AHEAD of the step (TrainStep's `upcoming` lookahead) without
disturbing the global stream the step draws from -- the tile choices then come from that private stream.
"""
import copy
import itertools
import math

import numpy as np

from . import augmentation
from .utils import get_new_img_size

AUGMENT_SWITCHES = augmentation.AUGMENT_SWITCHES              # augmentation.py:495-518
EXACT_AUGMENTATIONS = AUGMENT_SWITCHES[:3] + ("use_brightness",)   # no library semantics involved (see augmentation.py)


def brightness(img, rng=np.random):
    """augmentation.brightness on an image alone (kept for callers that have no boxes)."""
    return augmentation.brightness(img, [], rng=rng)[0]


def augment_geometric(img_data, img, C, rng=np.random, noise_rng=None, warp=None):
    """augmentation.augment (augmentation.py:481-533) for the feed: every switch of the reference, its draw order; returns
    (img_data copy with the augmented boxes / width / height, image)."""
    return augmentation.augment(img_data, img, C, augment=True, rng=rng, noise_rng=noise_rng, warp=warp)


def get_data(annot_path, data_path, img_types, load_image):
    """utils.get_data (utils.py:134-220): annotation CSV (columns img_path, label, xmin, ymin, xmax, ymax; one box per row)
    -> (data, class_count, class_mapping).  data: one {filepath, width, height, depth, bboxes} per image, in order of first
    appearance, filepath = data_path + '/' + img_path; an image is decoded once (load_image(img_data, img_types[0]), the
    caller's decoder) for its size, as the reference does; classes are numbered in order of first appearance and 'bg' is
    appended when no row carries it."""
    import csv
    images, class_count, class_mapping = {}, {}, {}
    with open(annot_path, newline="") as f:
        for row in csv.DictReader(f):
            name, cls = row["img_path"], row["label"]
            class_count[cls] = class_count.get(cls, 0) + 1
            if cls not in class_mapping:
                class_mapping[cls] = len(class_mapping)
            if name not in images:
                entry = {"filepath": data_path + "/" + name}
                img = load_image(entry, img_types[0])
                entry.update(width=img.shape[1], height=img.shape[0], depth=img.shape[2], bboxes=[])
                images[name] = entry
            # int() of the parsed number, like int(df.loc[i, 'xmin']): fractional coordinates truncate towards zero
            images[name]["bboxes"].append({"class": cls, "x1": int(float(row["xmin"])), "y1": int(float(row["ymin"])),
                                           "x2": int(float(row["xmax"])), "y2": int(float(row["ymax"]))})
    if "bg" not in class_count:
        class_count["bg"] = 0
        class_mapping["bg"] = len(class_mapping)
    return list(images.values()), class_count, class_mapping


class SampleSelector:
    """utils.py:19-59: cycle through the classes that occur; an image is skipped unless it holds the class whose turn it
    is, a tile likewise -- and a tile that does hold it advances the turn."""

    def __init__(self, class_count):
        self.classes = [c for c in class_count.keys() if class_count[c] > 0]
        self.class_cycle = itertools.cycle(self.classes)
        self.curr_class = next(self.class_cycle)

    def _has_current(self, img_data):
        return any(b["class"] == self.curr_class for b in img_data["bboxes"])

    def skip_image_for_balanced_class(self, img_data):
        return not self._has_current(img_data)

    def skip_tile_for_balanced_class(self, img_data):
        if self._has_current(img_data):
            self.curr_class = next(self.class_cycle)
            return False
        return True


def clip_box(bbox, img_box, alpha):
    """augmentation.py:33-83: clip (N, 4+) boxes x1 y1 x2 y2 to img_box; keep a box iff it touches the image box and loses
    less than (1 - alpha) of its area.  Returns (clipped boxes that are kept, keep mask over the input)."""
    bbox = np.asarray(bbox)
    x1, y1, x2, y2 = img_box[0], img_box[1], img_box[2], img_box[3]
    outside = (bbox[:, 0] > x2) | (bbox[:, 2] < x1) | (bbox[:, 1] > y2) | (bbox[:, 3] < y1)
    area = (bbox[:, 2] - bbox[:, 0]) * (bbox[:, 3] - bbox[:, 1])
    clipped = np.hstack((np.maximum(bbox[:, 0], x1).reshape(-1, 1), np.maximum(bbox[:, 1], y1).reshape(-1, 1),
                         np.minimum(bbox[:, 2], x2).reshape(-1, 1), np.minimum(bbox[:, 3], y2).reshape(-1, 1), bbox[:, 4:]))
    lost = (area - (clipped[:, 2] - clipped[:, 0]) * (clipped[:, 3] - clipped[:, 1])) / area
    mask = (outside == 0) & ((lost < (1 - alpha)).astype(int) == 1)
    return clipped[mask, :], mask


def _axis_spans(length, tile_size, step):
    start = np.arange(0, length, step)
    end = start + tile_size
    keep = end <= length
    start, end = np.append(start[keep], [max(0, length - tile_size)]), np.append(end[keep], [length])
    return np.unique(np.stack([start, end], axis=1), axis=0)            # sorted, duplicates (last == a regular one) merged


def tile_grid(width, height, tile_size, step):
    """utils.py:343-372: [x0, y0, x1, y1] of every tile, rows first; the last tile of an axis is flush with the border."""
    xs, ys = _axis_spans(width, tile_size, step), _axis_spans(height, tile_size, step)
    return [[int(x[0]), int(y[0]), int(x[1]), int(y[1])] for y in ys for x in xs]


class TileFeed:
    """Iterator over training samples (see module docstring).  data: list of {filepath, width, height, bboxes:[{class,x1,
    y1,x2,y2}]}; class_count: {class: number of boxes} (utils.get_data's third return value)."""

    def __init__(self, data, C, class_count, load_image, train_mode=True, rng=None, resize=None, noise_rng=None):
        self.data, self.C, self.load_image, self.train_mode = data, C, load_image, train_mode
        self.noise_rng = noise_rng                # numpy Generator of the noise augmentations' fields; None = unseeded, as scikit-image's
        self.warp = None                          # rotation / shear warp: the device kernel beside the device resize, else NumPy
        if resize is None and train_mode and (getattr(C, "use_rotations", False) or getattr(C, "use_shear", False)):
            from .RADNet import warp_affine_device
            self.warp = warp_affine_device
        self.rng = np.random if rng is None else rng
        self.selector = SampleSelector(class_count)
        self.resize = resize                      # (img, new_w, new_h) -> img; default: the device bicubic kernel

    def _image(self, img_data, random_type):
        types = self.C.img_types
        img_type = types[0]
        if random_type:                           # utils.get_image (utils.py:111-122)
            first = 0.5 if len(types) <= 3 else 0.3
            probs = [first] + [(1.0 - first) / (len(types) - 1) for _ in range(len(types) - 1)]
            img_type = self.rng.choice(types, 1, p=probs)[0]
        return self.load_image(img_data, img_type)

    def _sample(self, img, img_data):
        C = self.C
        width, height = img_data["width"], img_data["height"]
        if img.shape[1] != width or img.shape[0] != height:
            raise AssertionError("image size does not match its annotation")           # utils.py:437-438
        new_w, new_h = get_new_img_size(width, height, C.img_size)
        if (new_w, new_h) != (width, height):
            if self.resize is not None:
                img = self.resize(img, new_w, new_h)
            else:
                from .RADNet import resize_cubic
                img = resize_cubic(img, new_w, new_h)
        return dict(img=np.ascontiguousarray(img), bboxes=img_data["bboxes"], width=width, height=height,
                    filepath=img_data.get("filepath"))

    def __iter__(self):
        C, sel = self.C, self.selector
        balanced = self.train_mode and C.balanced_classes
        while True:
            if self.train_mode:
                self.rng.shuffle(self.data)
            for img_data in self.data:
                if balanced and sel.skip_image_for_balanced_class(img_data):
                    continue
                tiles = tile_grid(img_data["width"], img_data["height"], C.tile_size, C.tile_overlap)
                left = np.arange(0, len(tiles))
                n_tiles = min(len(tiles), C.max_n_tiles_train if self.train_mode else C.max_n_tiles_val)
                done = 0
                while done < n_tiles and len(left) > 0:
                    img = self._image(img_data, C.use_img_type)
                    pick = self.rng.randint(0, len(left))
                    tile = tiles[left[pick]]
                    left = np.delete(left, pick)
                    boxes = img_data["bboxes"]
                    arr = np.array([[b["x1"], b["y1"], b["x2"], b["y2"]] for b in boxes])
                    arr, keep = clip_box(arr, tile, C.tile_bbox_clip_threshold)
                    kept = [copy.deepcopy(boxes[i]) for i in range(keep.shape[0]) if keep[i] == 1]
                    if not kept:
                        continue
                    for i, b in enumerate(kept):
                        b["x1"], b["y1"] = int(arr[i, 0] - tile[0]), int(arr[i, 1] - tile[1])
                        b["x2"], b["y2"] = int(math.ceil(arr[i, 2] - tile[0])), int(math.ceil(arr[i, 3] - tile[1]))
                    crop = np.copy(img[tile[1]:tile[3], tile[0]:tile[2], :])
                    tile_data = dict(img_data, bboxes=kept, width=crop.shape[1], height=crop.shape[0])
                    if balanced and sel.skip_tile_for_balanced_class(tile_data):
                        continue
                    if self.train_mode:
                        tile_data, crop = augment_geometric(tile_data, crop, C, self.rng, self.noise_rng, self.warp)
                    done += 1
                    yield self._sample(crop, tile_data)
                if C.include_full_img:
                    if balanced and sel.skip_tile_for_balanced_class(img_data):
                        continue
                    img = self._image(img_data, C.use_img_type)
                    full = copy.deepcopy(img_data)
                    if self.train_mode:
                        full, img = augment_geometric(full, img, C, self.rng, self.noise_rng, self.warp)
                    yield self._sample(img, full)
            if not self.train_mode:
                return


class BackgroundFeed:
    """The same samples as `feed`, produced by a worker thread up to `depth` ahead of the consumer: tile cropping, augmentation
    and the resize run beside the train step instead of between two steps (NumPy releases the interpreter lock inside its array
    operations; the default Config's rotation / shear / noise cost 10-40 ms per 300-pixel tile on the host, the GPU step 2 ms).
    Only for feeds with a PRIVATE random stream (TileFeed(rng=RandomState)): a worker drawing from NumPy's global stream would
    interleave its draws with the step's.  TileFeed's device resize then runs on a context and HIP stream of the worker's own
    (radnet_hip.runtime.default_context), beside the step's lanes.
    Exceptions of the worker surface at the consumer's next(); close() (or exhausting the feed) ends the thread."""

    _END = object()

    def __init__(self, feed, depth=8):
        import queue
        import threading
        if getattr(feed, "rng", None) is np.random:
            raise ValueError("BackgroundFeed: the feed draws from NumPy's global random stream; give it its own RandomState")
        self._q = queue.Queue(maxsize=max(1, int(depth)))
        self._stop = threading.Event()
        self._feed = feed
        try:                                   # the worker must use THIS thread's GPU (per-thread current device; ranks of a DP job)
            from radnet_hip import runtime as _rt
            self._device = _rt.note_owner_device()
        except Exception:                      # no GPU stack importable: a host-only feed (resize hook given)
            self._device = None
        self._thread = threading.Thread(target=self._work, name="radnet-feed", daemon=True)
        self._thread.start()

    def _work(self):
        try:
            if self._device is not None:
                from radnet_hip import runtime as _rt
                _rt.bind_thread_device(self._device)
            for sample in self._feed:
                while not self._stop.is_set():
                    try:
                        self._q.put(sample, timeout=0.1)
                        break
                    except Exception:          # queue.Full: the consumer is behind, keep waiting (unless closed)
                        continue
                if self._stop.is_set():
                    return
            self._q.put(self._END)
        except BaseException as e:             # handed to the consumer
            self._q.put(e)

    def __iter__(self):
        return self

    def __next__(self):
        item = self._q.get()
        if item is self._END:
            self._q.put(self._END)
            raise StopIteration
        if isinstance(item, BaseException):
            self._q.put(item)
            raise item
        return item

    def close(self):
        self._stop.set()
        while not self._q.empty():             # unblock a worker waiting on a full queue
            try:
                self._q.get_nowait()
            except Exception:
                break
        self._thread.join(timeout=5)


def _insitu_tables(ts, first_batch, lookahead, next_batch, budget_s, cache_dir, dist_group, log):
    """run_training(tune=True): launch shapes tuned IN SITU for a job whose shapes no shipped table knows (radnet_hip/insitu.py).
    One table per (workload, network, panel size, images per step, device) under the user's cache directory: found -> loaded
    (rank 0 reads it, every rank loads the same text); missing -> walked over the first steps of THIS job, bounded by budget_s,
    rank 0 deciding for all ranks (every rank keeps stepping: the gradient exchanges stay symmetric), then written by rank 0.
    Returns the number of training steps the walk consumed."""
    import os
    import tempfile
    import torch
    from radnet_hip import insitu
    eng = ts.eng
    H, W = first_batch[0]["img"].shape[:2]
    comm = insitu.NoComm()
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(dist_group) > 1:
            comm = insitu.DistComm(dist, dist_group)
    except Exception:
        pass
    dev_name = torch.cuda.get_device_name(eng.dev) if torch.cuda.is_available() else "cpu"
    path = insitu.cache_path(getattr(eng, "NETWORK", "net"), getattr(eng, "workload", "train"), H, W, len(first_batch), dev_name, cache_dir)
    shipped = any("%dx%d_batch%d" % (H, W, len(first_batch)) in n for n in getattr(eng, "shipped_tuning", []))
    text = None
    if comm.rank == 0 and not shipped and os.path.exists(path):
        text = open(path).read()
    shipped, text = comm.bcast((shipped, text))
    if shipped:
        return 0                                    # this very workload has a shipped table: loaded when the engine was built
    if text is not None:
        with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
            f.write(text)
        eng.load_tuning(f.name)
        os.remove(f.name)
        log("radnet: in-situ table %s loaded" % path)
        return 0
    consumed = [0]
    if comm.rank == 0:
        os.makedirs(os.path.dirname(path), exist_ok=True)
    before, after, n_changed, left = insitu.tune_job(ts, next_batch, path, steps_done=lambda n: consumed.__setitem__(0, consumed[0] + n),
                                                     lookahead=lookahead, budget_s=budget_s, comm=comm, log=log,
                                                     measure=getattr(ts, "insitu_measure", None), sync=getattr(ts, "insitu_sync", None),
                                                     steps=getattr(ts, "insitu_steps", 100))
    log("radnet: launch shapes tuned in situ over %d steps: %.1f -> %.1f us per step (%d changes); table: %s" % (consumed[0], before, after, n_changed, path))
    return consumed[0], left


def run_training(ts, feed, n_steps, lookahead=3, on_step=None, tune=None, tune_budget_s=120.0, tune_cache_dir=None, dist_group=None, log=None):
    """Drive a TrainStep from a sample iterator, one sample per batch, announcing `lookahead` batches ahead (the pipelined
    step).  With lookahead > 0 the feed must draw from its own RandomState (see module docstring).  Returns the number of
    steps run.
    tune (default: the environment's RADNET_INSITU=1): tune the launch shapes in situ over the first steps of the job when neither
    a shipped nor a cached table knows this (network, panel size, batch, device) -- see _insitu_tables; the steps the walk runs are
    training steps of this job and count towards n_steps."""
    import os
    it = iter(feed)
    window = []
    done = 0
    log = log or (lambda m: __import__("sys").stderr.write(m + "\n"))
    if tune is None:
        tune = os.environ.get("RADNET_INSITU", "0") == "1"
    if tune and hasattr(ts, "eng") and hasattr(ts.eng, "load_tuning"):
        try:
            first = [next(it)]
        except StopIteration:
            return 0
        pending = [first]

        def next_batch():
            return pending.pop(0) if pending else [next(it)]

        got = _insitu_tables(ts, first, lookahead, next_batch, tune_budget_s, tune_cache_dir, dist_group, log)
        if isinstance(got, tuple):
            done, left = got
            window.extend(left)                     # batches the walk had announced but not yet stepped
        window[:0] = pending
        if on_step is not None and done:
            on_step(done, ts)

    def fill():
        while len(window) < lookahead + 1:
            try:
                window.append([next(it)])
            except StopIteration:
                break

    fill()
    while window and done < n_steps:
        batch = window.pop(0)
        fill()
        ts.step(batch, upcoming=window[:min(lookahead, n_steps - 1 - done)] if lookahead else None)
        done += 1
        if on_step is not None:
            on_step(done, ts)
    ts.flush()
    return done
