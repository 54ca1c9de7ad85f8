"""Seeded synthetic weights and panels (BASELINE.json: data = synthetic; no pretrained files exist offline --
README.md:38-39 points at a Dropbox download, vgg16.py:36-40 at keras' downloader).

He-normal convs with the stem scaled for mean-subtracted pixels, frozen-BN statistics near identity, so that
activations stay O(1)-O(10) through the 16 residual blocks.  Keys are the reference's Keras layer names
(resnet50.py:46-47,72-73,185-186,269,278; rpn.py:47,55,63) so real weights can replace these one-for-one.
"""
import numpy as np

from .engine import HEAD_STAGE, RES_STAGES


def conv_specs(n_anchors=12):
    specs = [("conv1", "bn_conv1", 7, 3, 64)]
    cin = 64
    for st, blocks, (f1, f2, f3), _ in RES_STAGES + (HEAD_STAGE,):
        if st == 5:
            cin = 1024
        for bl in blocks:
            base, bnb = "res%d%s_branch" % (st, bl), "bn%d%s_branch" % (st, bl)
            specs += [(base + "2a", bnb + "2a", 1, cin, f1), (base + "2b", bnb + "2b", 3, f1, f2), (base + "2c", bnb + "2c", 1, f2, f3)]
            if bl == "a":
                specs.append((base + "1", bnb + "1", 1, cin, f3))
            cin = f3
    specs += [("rpn_conv1", None, 3, 1024, 512), ("rpn_out_class", None, 1, 512, n_anchors), ("rpn_out_regress", None, 1, 512, 4 * n_anchors)]
    return specs


def synthetic_weights(seed=3, n_anchors=12, n_classes=7):
    rs = np.random.RandomState(seed)
    f = np.float32
    W = {}
    for name, bn, k, cin, cout in conv_specs(n_anchors):
        std = np.sqrt(2.0 / (k * k * cin))
        if bn is not None and name.endswith("2c"):
            std *= 0.25
        if name == "conv1":
            std /= 70.0
        W[name] = {"kernel": (rs.standard_normal((k, k, cin, cout)) * std).astype(f), "bias": (rs.standard_normal(cout) * 0.05).astype(f)}
        if bn is not None:
            W[bn] = {"gamma": rs.uniform(0.5, 1.5, cout).astype(f), "beta": (rs.standard_normal(cout) * 0.1).astype(f),
                     "mean": (rs.standard_normal(cout) * 0.1).astype(f), "var": rs.uniform(0.5, 1.5, cout).astype(f)}
    W["rpn_conv1"]["kernel"] *= f(0.1)
    W["rpn_out_class"]["kernel"] *= f(0.5)
    W["rpn_out_regress"]["kernel"] *= f(0.5)
    W["dense_class_%d" % n_classes] = {"kernel": (rs.standard_normal((2048, n_classes)) * 0.004).astype(f),
                                       "bias": (rs.standard_normal(n_classes) * 0.01).astype(f)}
    W["dense_regress_%d" % n_classes] = {"kernel": (rs.standard_normal((2048, 4 * (n_classes - 1))) * 0.004).astype(f),
                                         "bias": (rs.standard_normal(4 * (n_classes - 1)) * 0.01).astype(f)}
    return W


def synthetic_weights_vgg16(seed=5, n_anchors=9, n_classes=7):
    """VGG16 variant (vgg16.py:29-124 layer names: blockB_convI, fc1, fc2, rpn_*, dense_*); fc1 alone is 411 MB."""
    rs = np.random.RandomState(seed)
    f = np.float32
    W = {}
    cin = 3
    for b, n, ch in ((1, 2, 64), (2, 2, 128), (3, 3, 256), (4, 3, 512), (5, 3, 512)):
        for i in range(1, n + 1):
            name = "block%d_conv%d" % (b, i)
            std = np.sqrt(2.0 / (9 * cin)) * (1.0 / 70.0 if name == "block1_conv1" else 1.0)
            W[name] = {"kernel": (rs.standard_normal((3, 3, cin, ch)) * std).astype(f), "bias": (rs.standard_normal(ch) * 0.05).astype(f)}
            cin = ch
    W["rpn_conv1"] = {"kernel": (rs.standard_normal((3, 3, 512, 512)) * np.sqrt(2.0 / (9 * 512)) * 0.3).astype(f), "bias": (rs.standard_normal(512) * 0.05).astype(f)}
    W["rpn_out_class"] = {"kernel": (rs.standard_normal((1, 1, 512, n_anchors)) * 0.03).astype(f), "bias": np.zeros(n_anchors, f)}
    W["rpn_out_regress"] = {"kernel": (rs.standard_normal((1, 1, 512, 4 * n_anchors)) * 0.03).astype(f), "bias": np.zeros(4 * n_anchors, f)}
    W["fc1"] = {"kernel": (rs.standard_normal((7 * 7 * 512, 4096)) * np.sqrt(2.0 / (7 * 7 * 512))).astype(f), "bias": (rs.standard_normal(4096) * 0.05).astype(f)}
    W["fc2"] = {"kernel": (rs.standard_normal((4096, 4096)) * np.sqrt(2.0 / 4096)).astype(f), "bias": (rs.standard_normal(4096) * 0.05).astype(f)}
    nreg = 4 * (n_classes - 1)
    W["dense_class_%d" % n_classes] = {"kernel": (rs.standard_normal((4096, n_classes)) * 0.01).astype(f), "bias": (rs.standard_normal(n_classes) * 0.01).astype(f)}
    W["dense_regress_%d" % n_classes] = {"kernel": (rs.standard_normal((4096, nreg)) * 0.01).astype(f), "bias": (rs.standard_normal(nreg) * 0.01).astype(f)}
    return W


def synthetic_panel(seed, height=600, width=1000):
    """uint8 BGR panel (cfg 2/4 panels are 600x1000, seed 1 + offsets)."""
    return np.random.RandomState(seed).randint(0, 256, (height, width, 3)).astype(np.uint8)


def synthetic_gt(seed, n=8, src_w=2000, src_h=1200, class_names=("boat", "human", "other", "animal", "circle", "wheel"), smin=64, smax=400):
    """n ground-truth boxes in a src_w x src_h source frame (8 boxes, seed 2, sizes U[64,400]),
    classes round-robin over the foreground names (config.py:100-108)."""
    rs = np.random.RandomState(seed)
    out = []
    for i in range(n):
        bw, bh = int(rs.randint(smin, smax)), int(rs.randint(smin, smax))
        x1, y1 = int(rs.randint(0, max(1, src_w - bw))), int(rs.randint(0, max(1, src_h - bh)))
        out.append({"class": class_names[i % len(class_names)], "x1": x1, "x2": x1 + bw, "y1": y1, "y2": y1 + bh})
    return {"bboxes": out, "width": src_w, "height": src_h}
