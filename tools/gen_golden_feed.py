#!/usr/bin/env python3
"""Golden vectors for the host data feed (faster_rcnn/data_feed.py): the reference's own get_tile_generator
(utils.py:310-552), SampleSelector (utils.py:19-59) and augmentation.clip_box (augmentation.py:33-83), run HERE on a
synthetic in-memory dataset -- the image decoder (utils.get_image) is replaced by a dict lookup, the absent third-party
modules by the same stubs tools/gen_golden.py uses (tiles are produced at the network size, so the stubbed cv2.resize is an
identity).  Writes tests/golden/tile_feed.json (data only).

    python tools/gen_golden_feed.py
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G  # noqa: E402


def dataset(seed, sizes, classes):
    rs = np.random.RandomState(seed)
    data, imgs = [], {}
    for i, (w, h) in enumerate(sizes):
        path = "data/img_%d.png" % i
        boxes = []
        for j in range(int(rs.randint(3, 9))):
            bw, bh = int(rs.randint(30, 150)), int(rs.randint(30, 150))
            x1, y1 = int(rs.randint(0, max(1, w - bw))), int(rs.randint(0, max(1, h - bh)))
            boxes.append({"class": classes[int(rs.randint(len(classes)))], "x1": x1, "x2": x1 + bw, "y1": y1, "y2": y1 + bh})
        data.append({"filepath": path, "width": w, "height": h, "bboxes": boxes})
        imgs[path] = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
    return data, imgs


def main():
    rconfig, rrpn, rutils, rradnet = G._import_reference()
    import faster_rcnn.augmentation as raug
    out = {"numpy": np.__version__, "cases": [], "clip_box": [], "selector": []}
    classes = ["boat", "human", "other", "animal", "circle", "wheel"]

    # ---- clip_box ---------------------------------------------------------------------------------------------------
    rs = np.random.RandomState(21)
    for _ in range(12):
        n = int(rs.randint(1, 9))
        b = rs.randint(0, 600, (n, 4)).astype(np.float64)
        b[:, 2:] = b[:, :2] + rs.randint(5, 200, (n, 2))
        box = [int(v) for v in rs.randint(0, 400, 2)]
        box = box + [box[0] + 300, box[1] + 300]
        alpha = float(rs.choice([0.25, 0.5, 0.75, 0.9]))
        clipped, mask = raug.clip_box(b.copy(), box, alpha)
        out["clip_box"].append({"boxes": b.tolist(), "img_box": box, "alpha": alpha, "clipped": clipped.tolist(), "mask": [bool(m) for m in mask]})

    # ---- augmentation.brightness (pure NumPy in the reference): images + the global-stream seed -> outputs + stream position
    bright = {}
    rsb = np.random.RandomState(31)
    specs = [("mid", 120, 40), ("dark_skip", 40, 10), ("bright_skip", 230, 10), ("low_mid", 85, 30), ("high_mid", 170, 25), ("with_background", 130, 50)]
    for ci, (name, mean, spread) in enumerate(specs):
        img = np.clip(rsb.normal(mean, spread, (37, 53, 3)), 1, 255).astype(np.uint8)
        if name == "with_background":
            img[:10, :20] = 0
            img[20:, 30:, 1] = 0
        for si, seed in enumerate((5, 6, 7)):
            np.random.seed(seed)
            res, _ = raug.brightness(img.copy(), [])
            bright["c%d_s%d_out" % (ci, si)] = res
            bright["c%d_s%d_after" % (ci, si)] = np.int64(np.random.randint(0, 2 ** 31 - 1))
        bright["c%d_img" % ci] = img
    bright["n_cases"] = np.int64(len(specs))
    bright["seeds"] = np.array([5, 6, 7], dtype=np.int64)
    np.savez_compressed(os.path.join(G.OUT, "brightness.npz"), **bright)

    # ---- SampleSelector ---------------------------------------------------------------------------------------------
    counts = {"boat": 3, "human": 0, "other": 5, "animal": 1, "circle": 0, "wheel": 2}
    sel = rutils.SampleSelector(counts)
    rs = np.random.RandomState(22)
    seq = []
    for _ in range(40):
        img = {"bboxes": [{"class": classes[int(rs.randint(6))]} for _ in range(int(rs.randint(0, 4)))]}
        kind = int(rs.randint(2))
        res = sel.skip_tile_for_balanced_class(img) if kind else sel.skip_image_for_balanced_class(img)
        seq.append({"classes": [b["class"] for b in img["bboxes"]], "tile": kind, "skip": bool(res), "curr": sel.curr_class})
    out["selector"] = {"counts": [[k, v] for k, v in counts.items()], "seq": seq}      # pairs: the cycle follows dict order

    # ---- get_tile_generator -----------------------------------------------------------------------------------------
    # *_full cases: C.include_full_img (utils.py:484-549) -- after its tiles every image is yielded once more as a whole,
    # scaled to the network size by cv2.resize, which the stub binds to oracle/resize.py (tools/gen_golden.py): they pin the
    # reference's control flow, class balancing, random draws and anchor labels around the resize, not cv2's pixels
    for name, seed, train, balanced, n_take, sizes, full in (("train_balanced", 5, True, True, 12, [(900, 700), (650, 1000), (300, 300), (760, 450)], False),
                                                             ("train_plain", 6, True, False, 10, [(900, 700), (450, 620), (300, 300)], False),
                                                             ("val", 7, False, True, 99, [(900, 700), (650, 1000), (300, 300)], False),
                                                             ("train_plain_full", 8, True, False, 12, [(640, 480), (450, 620), (300, 300)], True),
                                                             ("train_balanced_full", 9, True, True, 14, [(640, 480), (500, 700), (300, 300), (420, 330)], True),
                                                             ("val_full", 10, False, False, 99, [(640, 480), (300, 300), (350, 520)], True)):
        C = rconfig.Config()
        C.img_size, C.tile_size, C.tile_overlap = 300, 300, 150
        C.max_n_tiles_train, C.max_n_tiles_val = 2, 3
        C.balanced_classes, C.include_full_img, C.use_img_type = balanced, full, False
        for k in ("use_horizontal_flips", "use_vertical_flips", "use_90_rotations", "use_rotations", "use_shear", "use_brightness", "use_noise"):
            setattr(C, k, False)
        data, imgs = dataset(seed, sizes, classes)
        class_count = {c: sum(1 for d in data for b in d["bboxes"] if b["class"] == c) for c in classes}
        rutils.get_image = lambda path, types, random_type=False: imgs[path]
        np.random.seed(100 + seed)
        gen = rutils.get_tile_generator([dict(d, bboxes=[dict(b) for b in d["bboxes"]]) for d in data], C, G.feat_size, class_count,
                                        lambda x: x, train_mode=train)
        yields = []
        try:
            for _ in range(n_take):
                x, Y, tile_data, dbg, best, n_pos = next(gen)
                yields.append({"filepath": tile_data["filepath"], "width": int(tile_data["width"]), "height": int(tile_data["height"]),
                               "bboxes": [{"class": b["class"], "x1": int(b["x1"]), "y1": int(b["y1"]), "x2": int(b["x2"]), "y2": int(b["y2"])}
                                          for b in tile_data["bboxes"]],
                               "img_sum": int(dbg.astype(np.int64).sum()), "n_pos": int(n_pos)})
        except (RuntimeError, StopIteration):        # val mode ends with `raise StopIteration` inside the generator (PEP 479)
            pass
        out["cases"].append({"name": name, "seed": 100 + seed, "train": train, "balanced": balanced, "sizes": sizes, "data_seed": seed, "full": full,
                             "yields": yields, "rng_after": int(np.random.randint(0, 2 ** 31 - 1))})
        print(name, len(yields), "yields")
    # ---- get_data ---------------------------------------------------------------------------------------------------
    import tempfile
    rs = np.random.RandomState(31)
    rows, sizes = ["img_path,label,xmin,ymin,xmax,ymax"], {}
    names = ["a/p1.png", "b/p2.png", "a/p3.png"]
    for i in range(17):
        nm = names[int(rs.randint(3))]
        sizes.setdefault("D/" + nm, (int(rs.randint(200, 900)), int(rs.randint(200, 900))))
        x1, y1 = rs.uniform(0, 150, 2)
        rows.append("%s,%s,%.2f,%.2f,%.2f,%.2f" % (nm, classes[int(rs.randint(1, 5))], x1, y1, x1 + rs.uniform(5, 90), y1 + rs.uniform(5, 90)))
    tmp = tempfile.mkdtemp()
    csv_path = os.path.join(tmp, "annot.csv")
    open(csv_path, "w").write("\n".join(rows) + "\n")
    rutils.get_image = lambda path, types, random_type=False: np.zeros((sizes[path][1], sizes[path][0], 3), np.uint8)
    data, cc, cm = rutils.get_data(csv_path, "D", ["t0", "t1"])
    out["get_data"] = {"csv": rows, "sizes": {k: list(v) for k, v in sizes.items()}, "data": data,
                       "class_count": [[k, int(v)] for k, v in cc.items()], "class_mapping": [[k, int(v)] for k, v in cm.items()]}
    with open(os.path.join(G.OUT, "tile_feed.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
