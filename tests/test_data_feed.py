"""Host data feed (faster_rcnn/data_feed.py) against vectors produced by the reference's own
get_tile_generator / SampleSelector / clip_box on a synthetic in-memory dataset (tools/gen_golden_feed.py):
same tiles in the same order, same clipped boxes, same consumption of NumPy's global random stream.  The reference computes
the anchor labels inside its generator (drawing from the same stream); here the step does that, so between two samples the
test advances the stream with the oracle's calc_region_props restatement (itself pinned by tests/golden/calc_region_props)."""
import json
import os

import numpy as np
import pytest

from faster_rcnn import data_feed as F
from faster_rcnn.config import Config
from oracle import glue

G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tile_feed.json")))
CLASSES = ["boat", "human", "other", "animal", "circle", "wheel"]
AUG = ("use_horizontal_flips", "use_vertical_flips", "use_90_rotations", "use_rotations", "use_shear", "use_brightness", "use_noise")


def dataset(seed, sizes):                       # the generator tools/gen_golden_feed.py used
    rs = np.random.RandomState(seed)
    data, imgs = [], {}
    for i, (w, h) in enumerate(sizes):
        path = "data/img_%d.png" % i
        boxes = []
        for j in range(int(rs.randint(3, 9))):
            bw, bh = int(rs.randint(30, 150)), int(rs.randint(30, 150))
            x1, y1 = int(rs.randint(0, max(1, w - bw))), int(rs.randint(0, max(1, h - bh)))
            boxes.append({"class": CLASSES[int(rs.randint(len(CLASSES)))], "x1": x1, "x2": x1 + bw, "y1": y1, "y2": y1 + bh})
        data.append({"filepath": path, "width": w, "height": h, "bboxes": boxes})
        imgs[path] = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
    return data, imgs


def test_clip_box_matches_reference():
    for c in G["clip_box"]:
        clipped, mask = F.clip_box(np.array(c["boxes"]), c["img_box"], c["alpha"])
        assert [bool(m) for m in mask] == c["mask"]
        assert np.array_equal(clipped, np.array(c["clipped"]).reshape(-1, 4))


def test_sample_selector_matches_reference():
    sel = F.SampleSelector(dict((k, v) for k, v in G["selector"]["counts"]))
    for s in G["selector"]["seq"]:
        img = {"bboxes": [{"class": c} for c in s["classes"]]}
        res = sel.skip_tile_for_balanced_class(img) if s["tile"] else sel.skip_image_for_balanced_class(img)
        assert bool(res) == s["skip"] and sel.curr_class == s["curr"]


def test_tile_grid():
    assert F.tile_grid(300, 300, 300, 150) == [[0, 0, 300, 300]]
    g = F.tile_grid(900, 700, 300, 150)
    assert g[0] == [0, 0, 300, 300] and g[-1] == [600, 400, 900, 700] and len(g) == 5 * 4
    assert [t[0] for t in g[:5]] == [0, 150, 300, 450, 600] and sorted({t[1] for t in g}) == [0, 150, 300, 400]
    assert F.tile_grid(200, 250, 300, 150) == [[0, 0, 200, 250]]          # image smaller than a tile: one tile, whole image


@pytest.mark.parametrize("case", G["cases"], ids=[c["name"] for c in G["cases"]])
def test_tile_feed_matches_reference_generator(case):
    C = Config()
    C.img_size, C.tile_size, C.tile_overlap = 300, 300, 150
    C.max_n_tiles_train, C.max_n_tiles_val = 2, 3
    C.balanced_classes, C.include_full_img, C.use_img_type = case["balanced"], bool(case.get("full", False)), False
    for k in AUG:
        setattr(C, k, False)
    data, imgs = dataset(case["data_seed"], [tuple(s) for s in case["sizes"]])
    class_count = {c: sum(1 for d in data for b in d["bboxes"] if b["class"] == c) for c in CLASSES}
    np.random.seed(case["seed"])
    # *_full cases (C.include_full_img, utils.py:484-549): the whole image follows its tiles, scaled to the network size.  The
    # reference ran with cv2.resize bound to oracle/resize.py (tools/gen_golden.py); the feed gets the same function as its
    # resize hook here (the device kernel that is its default is bit-identical to it: tests/test_gpu_resize.py, and
    # tests/test_gpu_radnet.py::test_tile_feed_full_image_pass_device_resize runs these cases through it)
    from oracle import resize as oresize
    feed = iter(F.TileFeed(data, C, class_count, lambda d, t: imgs[d["filepath"]], train_mode=case["train"],
                           resize=(lambda im, w, h: oresize.resize_bicubic_u8(np.ascontiguousarray(im), w, h)) if case.get("full") else None))
    n_full = 0
    got = 0
    for ref in case["yields"]:
        s = next(feed)
        assert (s["filepath"], s["width"], s["height"]) == (ref["filepath"], ref["width"], ref["height"])
        assert [{k: b[k] for k in ("class", "x1", "y1", "x2", "y2")} for b in s["bboxes"]] == ref["bboxes"]
        assert int(s["img"].astype(np.int64).sum()) == ref["img_sum"] and s["img"].dtype == np.uint8
        # the reference's generator now labels the anchors, drawing from the same stream (utils.py:451)
        boxes = np.array([[b["x1"], b["y1"], b["x2"], b["y2"]] for b in s["bboxes"]], dtype=np.float64)
        is_bg = np.array([b["class"] == "bg" for b in s["bboxes"]])
        rw, rh = glue.new_img_size(s["width"], s["height"], C.img_size)
        n_pos = glue.anchor_targets(C, boxes, is_bg, s["width"], s["height"], rw, rh, lambda w, h: (glue.resnet50_feat_len(w), glue.resnet50_feat_len(h)))[3]
        assert int(n_pos) == ref["n_pos"]
        got += 1
        if (s["width"], s["height"]) != (300, 300):
            n_full += 1
            assert s["img"].shape[:2] == (rh, rw) and min(rw, rh) == 300          # the sample carries the RESIZED panel
    assert (n_full > 0) == bool(case.get("full", False))
    if not case["train"]:
        with pytest.raises(StopIteration):          # one pass in validation mode (the reference ends it with an exception)
            next(feed)
    assert np.random.randint(0, 2 ** 31 - 1) == case["rng_after"]


def test_default_config_feeds_and_private_rng():
    C = Config()
    F.TileFeed([], C, {"boat": 1}, None, train_mode=True)              # the default Config has every augmentation on: accepted
    for k in AUG:
        setattr(C, k, False)
    C.img_size, C.tile_size, C.tile_overlap, C.balanced_classes = 300, 300, 150, False
    data, imgs = dataset(3, [(640, 480), (300, 300)])
    np.random.seed(1)
    probe = np.random.RandomState(1).randint(0, 2 ** 31 - 1)
    feed = iter(F.TileFeed(data, C, {c: 1 for c in CLASSES}, lambda d, t: imgs[d["filepath"]], rng=np.random.RandomState(9)))
    for _ in range(5):
        next(feed)
    assert np.random.randint(0, 2 ** 31 - 1) == probe                   # the global stream was not touched


def test_geometric_augmentations_vs_loop_restatement():
    """Flips / 90-degree rotations (augmentation.py:85-159): array reversals and transposes, no interpolation.  Parity
    unpinned against OpenCV's cv2.flip itself (absent); checked against explicit pixel loops, for every draw outcome, and
    that a box keeps covering the same pixels."""
    from oracle.evaluate import augment_geometric_loops

    class FixedRng:                                   # replays chosen outcomes in the order the reference draws them
        def __init__(self, coins, angle):
            self.coins, self.angle = list(coins), angle
        def random(self):
            return self.coins.pop(0)
        def choice(self, a, n):
            return np.array([self.angle])

    rs = np.random.RandomState(4)
    img = rs.randint(0, 256, (37, 53, 3)).astype(np.uint8)
    boxes = [{"class": "boat", "x1": 5, "y1": 7, "x2": 20, "y2": 30}, {"class": "human", "x1": 0, "y1": 0, "x2": 53, "y2": 37}]
    C = Config()
    C.use_horizontal_flips = C.use_vertical_flips = C.use_90_rotations = True
    C.use_brightness = C.use_rotations = C.use_shear = C.use_noise = False
    for hf in (False, True):
        for vf in (False, True):
            for angle in (None, 90, 180, 270):
                coins = [0.1 if hf else 0.9, 0.1 if vf else 0.9, 0.1 if angle else 0.9]
                data = {"filepath": "t.png", "bboxes": [dict(b) for b in boxes], "width": 53, "height": 37}
                d, out = F.augment_geometric(data, img, C, FixedRng(coins, angle))
                rb, rimg = augment_geometric_loops(boxes, img, (hf, vf), angle)
                assert np.array_equal(out, rimg) and d["bboxes"] == rb
                assert (d["width"], d["height"]) == (out.shape[1], out.shape[0])
                b0, o0 = d["bboxes"][0], boxes[0]                   # the box still frames the same pixels
                assert sorted(out[b0["y1"]:b0["y2"], b0["x1"]:b0["x2"]].ravel()) == sorted(img[o0["y1"]:o0["y2"], o0["x1"]:o0["x2"]].ravel())
    # a switched-off augmentation draws nothing
    C.use_vertical_flips = False
    rng = FixedRng([0.9, 0.9], 90)
    F.augment_geometric({"filepath": "t.png", "bboxes": [], "width": 53, "height": 37}, img, C, rng)
    assert rng.coins == []


def test_get_data_matches_reference(tmp_path):
    g = G["get_data"]
    p = tmp_path / "annot.csv"
    p.write_text("\n".join(g["csv"]) + "\n")
    sizes = g["sizes"]
    data, cc, cm = F.get_data(str(p), "D", ["t0", "t1"], lambda d, t: np.zeros((sizes[d["filepath"]][1], sizes[d["filepath"]][0], 3), np.uint8))
    assert data == g["data"]
    assert [[k, v] for k, v in cc.items()] == g["class_count"] and [[k, v] for k, v in cm.items()] == g["class_mapping"]


def test_run_training_announces_the_right_batches():
    """run_training pulls samples `lookahead` ahead of the step and announces exactly the batches the next calls bring, in
    order; it stops announcing when the run is about to end and flushes once."""

    class Recorder:
        def __init__(self):
            self.calls, self.flushed = [], 0
        def step(self, batch, upcoming=None):
            self.calls.append((batch[0]["id"], [b[0]["id"] for b in (upcoming or [])]))
        def flush(self):
            self.flushed += 1

    feed = ({"id": i} for i in range(100))
    ts = Recorder()
    assert F.run_training(ts, feed, 6, lookahead=3) == 6 and ts.flushed == 1
    assert ts.calls == [(0, [1, 2, 3]), (1, [2, 3, 4]), (2, [3, 4, 5]), (3, [4, 5]), (4, [5]), (5, [])]
    ts = Recorder()
    assert F.run_training(ts, iter([{"id": 0}, {"id": 1}]), 10, lookahead=3) == 2          # feed shorter than the run
    assert ts.calls == [(0, [1]), (1, [])]
    ts = Recorder()
    F.run_training(ts, ({"id": i} for i in range(9)), 3, lookahead=0)
    assert ts.calls == [(0, []), (1, []), (2, [])]


def test_brightness_matches_reference_outputs():
    """augmentation.brightness (augmentation.py:303-333) is pure NumPy in the reference: tests/golden/brightness.npz holds its
    outputs and the position of the global random stream afterwards (tools/gen_golden_feed.py) for six images x three seeds,
    incl. the no-draw early returns and channel-wise background pixels.  Byte-exact, same stream consumption."""
    import os
    from faster_rcnn import data_feed as F
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "brightness.npz"), allow_pickle=False)
    changed = 0
    for ci in range(int(g["n_cases"])):
        img = g["c%d_img" % ci]
        for si, seed in enumerate(g["seeds"]):
            np.random.seed(int(seed))
            out = F.brightness(img.copy())
            assert out.dtype == np.uint8 and np.array_equal(out, g["c%d_s%d_out" % (ci, si)]), (ci, si)
            assert int(np.random.randint(0, 2 ** 31 - 1)) == int(g["c%d_s%d_after" % (ci, si)]), (ci, si)
            changed += int(not np.array_equal(out, img))
    assert changed >= 9                          # the in-range images really were shifted


def test_background_feed_same_samples_and_error_passing():
    """BackgroundFeed: the worker thread yields exactly the feed's samples, in order, ahead of the consumer; refuses a feed on the
    global random stream; passes the worker's exception on; close() ends a worker blocked on a full queue."""
    import time
    C = Config()
    C.img_size, C.tile_size, C.tile_overlap, C.balanced_classes = 300, 300, 150, False
    C.img_types = ["rgb"]
    data, imgs = dataset(3, [(640, 480), (300, 300), (500, 620)])
    cc = {c: 1 for c in CLASSES}
    ident = lambda img, w, h: np.ascontiguousarray(img[:h, :w]) if img.shape[:2] != (h, w) else img      # host stand-in for the device resize
    load = lambda d, t: imgs[d["filepath"]]

    def take(it, n):
        out = []
        for s in it:
            out.append((s["filepath"], s["width"], s["height"], [tuple(sorted(b.items())) for b in s["bboxes"]], int(s["img"].astype(np.int64).sum())))
            if len(out) == n:
                break
        return out

    ref = take(iter(F.TileFeed([dict(d) for d in data], C, cc, load, rng=np.random.RandomState(5), resize=ident, noise_rng=np.random.default_rng(2))), 12)
    bg = F.BackgroundFeed(F.TileFeed([dict(d) for d in data], C, cc, load, rng=np.random.RandomState(5), resize=ident, noise_rng=np.random.default_rng(2)), depth=4)
    time.sleep(0.3)                                  # the worker runs ahead on its own
    assert bg._q.qsize() >= 1
    got = take(bg, 12)
    bg.close()
    assert got == ref and len(ref) == 12
    assert not bg._thread.is_alive()
    with pytest.raises(ValueError):
        F.BackgroundFeed(F.TileFeed([dict(d) for d in data], C, cc, load, resize=ident))         # global stream

    def broken():
        yield {"n": 1}
        raise KeyError("decoder failed")

    class G:                                          # any iterable with a private stream
        rng = np.random.RandomState(1)
        def __iter__(self):
            return broken()

    b2 = F.BackgroundFeed(G(), depth=2)
    assert next(b2) == {"n": 1}
    with pytest.raises(KeyError):
        next(b2)
    b2.close()
    val = F.BackgroundFeed(F.TileFeed([dict(d) for d in data], C, cc, load, train_mode=False, rng=np.random.RandomState(5), resize=ident), depth=3)
    n = len(list(val))                                # a finite feed ends the iteration
    assert n == len(list(F.TileFeed([dict(d) for d in data], C, cc, load, train_mode=False, rng=np.random.RandomState(5), resize=ident))) > 0
    val.close()
