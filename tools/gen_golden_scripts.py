#!/usr/bin/env python3
"""Golden vectors for the pure-NumPy functions that live in the reference's SCRIPTS (not in its package):

    train.get_selected_samples   (train.py:93-129)     -> tests/golden/selected_samples.json
    test.get_objects             (test.py:48-115)      -> tests/golden/voc_ap.json
    test.calc_class_ap           (test.py:119-173)

Runs ONLY in the build container (needs /root/reference).  `train.py` / `test.py` are imported from where they lie; their
`main()` is guarded (train.py:711, test.py:264), the module bodies are imports + constants + defs.  The third-party modules
they import at the top and that are absent here (cv2, tensorflow, keras.*) are satisfied with EMPTY stub modules -- none of
the three functions touches them; matplotlib.pyplot is stubbed too (no display back end is wanted), sklearn / pandas / tqdm
are really present.  The fixtures hold data only: inputs and what the reference's functions returned.

    python tools/gen_golden_scripts.py
"""
import copy
import json
import os
import sys
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G  # noqa: E402

OUT = G.OUT


class _Anything:
    """Attribute sink: `from keras.x import Name` succeeds, nothing is ever called."""

    def __getattr__(self, name):
        return _Anything()

    def __call__(self, *a, **k):
        raise RuntimeError("stub called: the goldens must not depend on absent third-party code")


def _stub_script_imports():
    G._stub_modules()                      # cv2, skimage, keras.layers / keras.models

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    any_ = _Anything()
    mod("tensorflow", float32="float32")
    keras = sys.modules["keras"]
    keras.backend = mod("keras.backend")
    keras.objectives = mod("keras.objectives", categorical_crossentropy=any_)
    keras.optimizers = mod("keras.optimizers", Adam=any_, SGD=any_, Nadam=any_)
    keras.callbacks = mod("keras.callbacks", TensorBoard=any_)
    keras.utils = mod("keras.utils", generic_utils=any_)
    mpl = mod("matplotlib")
    mpl.pyplot = mod("matplotlib.pyplot")


def import_scripts():
    _stub_script_imports()
    sys.path.insert(0, G.REF)
    import importlib
    rtrain = importlib.import_module("train")
    rtest = importlib.import_module("test")
    for m in (rtrain, rtest):
        assert os.path.realpath(m.__file__).startswith(os.path.realpath(G.REF)), m.__file__
    return rtrain, rtest


CLASSES = ("boat", "human", "other", "animal", "circle", "wheel")


def random_case(seed, n_gt, n_pred, tie_probs=False):
    rs = np.random.RandomState(seed)
    gt, pred = [], []
    for i in range(n_gt):
        x1, y1 = int(rs.randint(0, 900)), int(rs.randint(0, 500))
        w, h = int(rs.randint(20, 200)), int(rs.randint(20, 200))
        gt.append({"class": CLASSES[rs.randint(len(CLASSES))], "x1": x1, "y1": y1, "x2": x1 + w, "y2": y1 + h})
    for i in range(n_pred):
        if gt and rs.rand() < 0.6:                        # jittered copy of a ground-truth box
            g = gt[rs.randint(len(gt))]
            j = rs.randint(-25, 26, 4)
            box = (g["x1"] + j[0], g["y1"] + j[1], g["x2"] + j[2], g["y2"] + j[3])
            cls = g["class"] if rs.rand() < 0.8 else CLASSES[rs.randint(len(CLASSES))]
        else:
            x1, y1 = int(rs.randint(0, 900)), int(rs.randint(0, 500))
            box = (x1, y1, x1 + int(rs.randint(-5, 200)), y1 + int(rs.randint(-5, 200)))      # may be degenerate
            cls = CLASSES[rs.randint(len(CLASSES))]
        # distinct probabilities unless ties are asked for: np.argsort's order among equal keys is implementation-defined
        prob = float(rs.choice([0.81, 0.9, 0.95])) if tie_probs else float(0.8 + 0.2 * (i + rs.uniform(0.1, 0.9)) / max(n_pred, 1))
        pred.append({"class": cls, "x1": int(box[0]), "y1": int(box[1]), "x2": int(box[2]), "y2": int(box[3]), "prob": prob})
    order = rs.permutation(n_pred)
    return [pred[k] for k in order], gt


def main():
    rtrain, rtest = import_scripts()
    os.makedirs(OUT, exist_ok=True)

    # ---- get_selected_samples: every branch, global NumPy RNG seeded, stream position pinned afterwards ------------------
    class Cfg:
        pass

    cases = []
    specs = [  # (name, n, n_pos, n_rois, seed)
        ("few_pos_many_neg", 120, 3, 20, 64),          # pos < n_rois//2: all positives, negatives without replacement
        ("many_pos_many_neg", 300, 57, 20, 65),        # choice(pos, 10) then choice(neg, 10)
        ("exactly_half_pos", 90, 10, 20, 66),          # len(pos) == n_rois//2: the else branch (choice of all 10)
        ("neg_short_replace", 16, 9, 20, 67),          # 7 negatives < 11 needed: ValueError -> replace=True
        ("one_neg", 30, 29, 20, 68),                   # a single negative drawn 10 times with replacement
        ("no_neg", 14, 14, 20, 69),                    # no bg row: positives permuted, then with replacement
        ("no_neg_many_pos", 40, 40, 20, 70),           # no bg row, more positives than n_rois: returns 40 + (-20 -> error?)
        ("no_pos", 50, 0, 20, 71),                     # only bg rows
        ("n_rois_4", 33, 6, 4, 72),
        ("n_rois_odd", 45, 11, 7, 73),
    ]
    for name, n, n_pos, n_rois, seed in specs:
        rs = np.random.RandomState(1000 + seed)
        nc = 7
        cls = np.full(n, nc - 1, dtype=np.int64)
        pos_rows = rs.permutation(n)[:n_pos]
        cls[pos_rows] = rs.randint(0, nc - 1, n_pos)
        Y1 = np.zeros((1, n, nc), dtype=np.float64)
        Y1[0, np.arange(n), cls] = 1.0
        C = Cfg(); C.n_rois = n_rois
        np.random.seed(seed)
        rec = {"name": name, "n_rois": n_rois, "seed": seed, "cls": cls.tolist(), "bg": nc - 1}
        try:
            sel, npos = rtrain.get_selected_samples(Y1, C)
            rec.update(raised=None, sel=[int(v) for v in sel], n_pos=int(npos))
        except Exception as e:          # the reference's own behaviour for this input (recorded, not judged)
            rec.update(raised=type(e).__name__, sel=None, n_pos=None)
        rec["rng_after"] = int(np.random.randint(0, 2 ** 31 - 1))
        cases.append(rec)
    with open(os.path.join(OUT, "selected_samples.json"), "w") as f:
        json.dump({"numpy": np.__version__, "source": "train.get_selected_samples (train.py:93-129) run in the build container",
                   "cases": cases}, f, indent=0)

    # ---- get_objects / calc_class_ap -------------------------------------------------------------------------------------
    out = {"numpy": np.__version__, "source": "test.get_objects / test.calc_class_ap (test.py:48-173) run in the build container",
           "objects": [], "ap": []}
    for seed, n_gt, n_pred, ties in [(0, 12, 40, False), (1, 30, 25, False), (2, 0, 10, False), (3, 9, 0, False), (4, 60, 300, False),
                                     (5, 1, 1, False), (6, 25, 80, False), (7, 10, 30, True)]:
        pred, gt = random_case(seed, n_gt, n_pred, ties)
        gt_run = copy.deepcopy(gt)
        if n_pred == 0:
            # np.argsort of an empty float array works; the reference handles it (loop does not run)
            pass
        T, P = rtest.get_objects(copy.deepcopy(pred), gt_run, rtest.GT_IOU_THRESHOLD)
        rec = {"seed": seed, "ties": ties, "threshold": rtest.GT_IOU_THRESHOLD, "pred": pred, "gt": gt, "keys": list(T.keys()),
               "T": {k: [int(v) for v in T[k]] for k in T}, "P": {k: [float(v) for v in P[k]] for k in P},
               "matched": [bool(g["bbox_matched"]) for g in gt_run]}
        out["objects"].append(rec)
        for k in T:
            if ties:
                continue                      # equal scores: argsort order is implementation-defined, AP curve not pinned
            ap, prec, rec_, ip, ir = rtest.calc_class_ap(T[k], P[k])
            out["ap"].append({"y_true": [int(v) for v in T[k]], "y_pred": [float(v) for v in P[k]], "ap": float(ap),
                              "precision": [float(v) for v in prec], "recall": [float(v) for v in rec_],
                              "interp_precision": [float(v) for v in ip], "interp_recall": [float(v) for v in ir]})
    # hand-made AP edge cases
    for yt, yp in [([1, 0, 1, 1], [0.9, 0.8, 0.7, 0.0]), ([0, 0, 0], [0.9, 0.85, 0.8]), ([1, 1], [0.0, 0.0]), ([1], [0.99]),
                   ([1, 1, 0, 1, 0, 0, 1], [0.99, 0.97, 0.96, 0.9, 0.85, 0.83, 0.0])]:
        ap, prec, rec_, ip, ir = rtest.calc_class_ap(yt, yp)
        out["ap"].append({"y_true": yt, "y_pred": yp, "ap": float(ap), "precision": [float(v) for v in prec],
                          "recall": [float(v) for v in rec_], "interp_precision": [float(v) for v in ip],
                          "interp_recall": [float(v) for v in ir]})
    with open(os.path.join(OUT, "voc_ap.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote selected_samples.json (%d cases), voc_ap.json (%d object cases, %d AP curves)" % (len(cases), len(out["objects"]), len(out["ap"])))


if __name__ == "__main__":
    main()
