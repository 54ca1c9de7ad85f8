"""Soak: default-Config training from TileFeed -> BackgroundFeed -> TrainStep for N steps (300-pixel tiles, every augmentation on,
tile sizes changing), device memory and step rate reported every 100 steps.  usage: python tools/soak_feed.py [steps]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rock-art-radnet_amd")]

from faster_rcnn import data_feed as F  # noqa: E402
from faster_rcnn.config import Config  # noqa: E402
from radnet_hip import synth  # noqa: E402
from radnet_hip.engine import FasterRCNNEngine  # noqa: E402
from radnet_hip.trainer import TrainStep  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    C = Config()
    C.img_size, C.tile_size, C.tile_overlap, C.balanced_classes = 300, 300, 150, True
    rs = np.random.RandomState(1)
    classes = [k for k in C.class_mapping if k != "bg"]
    data, imgs = [], {}
    for i in range(6):
        w, h = int(rs.randint(500, 1100)), int(rs.randint(500, 900))
        boxes = []
        for j in range(10):
            bw, bh = int(rs.randint(40, 160)), int(rs.randint(40, 160))
            x1, y1 = int(rs.randint(0, w - bw)), int(rs.randint(0, h - bh))
            boxes.append({"class": classes[j % len(classes)], "x1": x1, "x2": x1 + bw, "y1": y1, "y2": y1 + bh})
        data.append({"filepath": "img%d" % i, "width": w, "height": h, "bboxes": boxes})
        imgs["img%d" % i] = rs.randint(1, 256, (h, w, 3)).astype(np.uint8)
    cc = {c: sum(1 for d in data for b in d["bboxes"] if b["class"] == c) for c in classes}
    eng = FasterRCNNEngine(C, autotune=2)
    eng.set_weights(synth.synthetic_weights(seed=3))
    np.random.seed(2)
    ts = TrainStep(eng)
    feed = F.BackgroundFeed(F.TileFeed(data, C, cc, lambda d, t: imgs[d["filepath"]], rng=np.random.RandomState(3)), depth=16)
    t0 = [time.perf_counter()]
    sizes = set()

    def on_step(k, t):
        if k % 100 == 0:
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0[0]
            lo = t.losses()
            print("step %5d  %.1f steps/s  device memory %.2f GB (reserved %.2f)  plans %d  graphs %d  skipped heads %d  dropped %d  rpn_cls %.4f det_cls %.4f" % (
                k, 100 / dt, torch.cuda.memory_allocated() / 2 ** 30, torch.cuda.memory_reserved() / 2 ** 30, len(eng._plans), len(eng._graphs),
                t.skipped_head_steps, t.dropped_images, lo["rpn_cls"], lo["det_cls"]), flush=True)
            t0[0] = time.perf_counter()

    try:
        n = F.run_training(ts, feed, steps, lookahead=3, on_step=on_step)
    finally:
        feed.close()
    print("ran", n, "steps")


if __name__ == "__main__":
    main()
