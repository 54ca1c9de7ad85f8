#!/usr/bin/env python3
"""Golden vectors for faster_rcnn/augmentation.py of this package: the reference's own augmentation module
(/root/reference/faster_rcnn/augmentation.py) run HERE on synthetic tiles.  Writes tests/golden/augment.npz (data only).

Two kinds of cases:
  pure   -- functions of the reference that need neither OpenCV nor scikit-image (strap_img, random_crop, the SciPy
            truncated normal): the reference's outputs as they are.
  around -- the reference's any_degree_rotation / shear / contrast / noise functions and its augment() dispatcher, with the
            library calls they make (cv2.flip / getRotationMatrix2D / warpAffine, skimage.exposure.rescale_intensity,
            skimage.util.random_noise / img_as_ubyte -- both libraries are absent from this image) bound to THIS repo's
            restatements.  These cases pin everything the reference does around those calls (draw order and count, parameters,
            matrix shift, canvas sizes, box arithmetic, strapping, clipping, rounding, dispatch, background handling); they
            say nothing about the libraries' own arithmetic, which stays "parity unpinned" (augmentation.py header).
            The noise field is drawn from numpy.random.default_rng(NOISE_SEED) per call, because scikit-image's own
            unseeded generator would make the reference's output irreproducible.

    python tools/gen_golden_augment.py
"""
import copy
import importlib.util
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G  # noqa: E402

NOISE_SEED = 77
HERE = os.path.dirname(os.path.abspath(__file__))


def load_mine():
    spec = importlib.util.spec_from_file_location("radnet_augmentation", os.path.join(HERE, "..", "rock-art-radnet_amd", "faster_rcnn", "augmentation.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def tile(rs, h, w, grey, holes=True):
    if grey:
        img = np.repeat(rs.randint(1, 256, (h, w, 1)), 3, axis=2).astype(np.uint8)
    else:
        img = rs.randint(1, 256, (h, w, 3)).astype(np.uint8)
    if holes:                                                        # background: exact zeros
        img[rs.randint(0, h // 2):rs.randint(h // 2, h), rs.randint(0, w // 3)] = 0
        img[:3, :5] = 0
    return img


def boxes_for(rs, h, w, n):
    out = []
    for _ in range(n):
        bw, bh = int(rs.randint(6, w // 2)), int(rs.randint(6, h // 2))
        x1, y1 = int(rs.randint(0, w - bw)), int(rs.randint(0, h - bh))
        out.append({"class": "c%d" % rs.randint(3), "x1": x1, "y1": y1, "x2": x1 + bw, "y2": y1 + bh})
    return out


def box_rows(bboxes):
    return np.array([[b["x1"], b["y1"], b["x2"], b["y2"]] for b in bboxes], dtype=np.float64).reshape(-1, 4)


def main():
    rconfig, _, _, _ = G._import_reference()
    import faster_rcnn.augmentation as raug
    assert os.path.realpath(raug.__file__).startswith(os.path.realpath(G.REF))
    mine = load_mine()
    out = {"noise_seed": np.int64(NOISE_SEED)}

    # ---- pure: strap_img -------------------------------------------------------------------------------------------
    rs = np.random.RandomState(41)
    n = 0
    for k in range(6):
        img = np.zeros((40, 56, 3), dtype=np.float32 if k >= 4 else np.uint8)
        r0, c0 = int(rs.randint(0, 15)), int(rs.randint(0, 20))
        img[r0:r0 + int(rs.randint(5, 20)), c0:c0 + int(rs.randint(5, 30))] = 9
        if k == 3:
            img[:, :, 1] = 0
            img[7, 9, 1] = 1
            img[22, 31, 1] = 1
        if k >= 4:
            img[2:30, 3:40, 1] = np.nan if k == 4 else np.inf
        out["strap%d_img" % n] = img
        out["strap%d_out" % n] = np.array(raug.strap_img(img), dtype=np.int64)
        n += 1
    out["n_strap"] = np.int64(n)

    # ---- pure: random_crop, truncated normal -------------------------------------------------------------------------
    rs = np.random.RandomState(42)
    for k in range(8):
        img = tile(rs, 90, 120, False)
        bb = boxes_for(rs, 90, 120, 5)
        np.random.seed(300 + k)
        res, rb = raug.random_crop(img.copy(), copy.deepcopy(bb))
        out["crop%d_img" % k], out["crop%d_boxes" % k] = img, box_rows(bb)
        out["crop%d_out" % k], out["crop%d_out_boxes" % k] = res, box_rows(rb)
        out["crop%d_after" % k] = np.int64(np.random.randint(0, 2 ** 31 - 1))
    out["n_crop"] = np.int64(8)
    np.random.seed(7)
    out["truncnorm"] = np.array([raug.get_truncated_normal(mean=0.5, sd=0.1, low=0, upp=1).rvs(size=1)[0] for _ in range(5)])
    out["truncnorm_after"] = np.int64(np.random.randint(0, 2 ** 31 - 1))

    # ---- around: bind the absent libraries' entry points to this repo's restatements ----------------------------------
    cv2, sk = raug.cv2, raug.skimage
    cv2.flip = mine.flip_u8
    cv2.getRotationMatrix2D = mine.rotation_matrix_2d
    cv2.warpAffine = lambda img, mat, size: mine.warp_affine_u8(img, mat, size)
    sk.exposure, sk.util = sys.modules["skimage.exposure"], sys.modules["skimage.util"]
    sk.exposure.rescale_intensity = lambda img, in_range: mine.rescale_intensity(img, in_range)
    sk.util.random_noise = lambda img, mode, clip=True, **kw: mine.random_noise(img, mode, np.random.default_rng(NOISE_SEED), **kw)
    sk.util.img_as_ubyte = mine.img_as_ubyte

    singles = (("rot", lambda i, b, t: raug.any_degree_rotation(i, b)), ("shear", lambda i, b, t: raug.shear(i, b)),
               ("contrast", lambda i, b, t: raug.contrast(i, b)), ("sp", lambda i, b, t: raug.salt_and_pepper_noise(i, b, t)),
               ("gauss", lambda i, b, t: raug.gaussian_noise(i, b, t)), ("poisson", lambda i, b, t: raug.poisson_noise(i, b, t)))
    rs = np.random.RandomState(43)
    for name, fn in singles:
        for k in range(6):
            grey = k % 2 == 1
            img = tile(rs, 70 + 9 * k, 100 - 7 * k, grey, holes=name not in ("rot", "shear") or k % 3 == 0)
            bb = boxes_for(rs, img.shape[0], img.shape[1], 4)
            np.random.seed(500 + k)
            res, rb = fn(img.copy(), copy.deepcopy(bb), "grey_x" if grey else "rgb")
            key = "%s%d" % (name, k)
            out[key + "_img"], out[key + "_boxes"], out[key + "_grey"] = img, box_rows(bb), np.int64(grey)
            out[key + "_out"], out[key + "_out_boxes"] = res, box_rows(rb)
            out[key + "_after"] = np.int64(np.random.randint(0, 2 ** 31 - 1))
    out["n_single"] = np.int64(6)

    # ---- around: the dispatcher, every switch on ------------------------------------------------------------------------
    rs = np.random.RandomState(44)
    n_aug = 48
    for k in range(n_aug):
        grey = k % 3 == 2
        C = rconfig.Config()
        for sw in mine.AUGMENT_SWITCHES:
            setattr(C, sw, True)
        if k >= 40:                                                   # a few with switches off: the coins must not be drawn
            C.use_rotations = k % 2 == 0
            C.use_shear = False
            C.use_noise = k % 4 == 1
        C.img_types = ["grey_x" if grey else "rgb"]
        h, w = int(rs.randint(60, 110)), int(rs.randint(60, 130))
        img = tile(rs, h, w, grey, holes=k % 2 == 0)
        data = {"filepath": "t%d.png" % k, "width": w, "height": h, "bboxes": boxes_for(rs, h, w, int(rs.randint(1, 6)))}
        np.random.seed(900 + k)
        rd, res = raug.augment(copy.deepcopy(data), img.copy(), C, augment=True)
        key = "aug%d" % k
        out[key + "_img"], out[key + "_boxes"], out[key + "_grey"] = img, box_rows(data["bboxes"]), np.int64(grey)
        out[key + "_switches"] = np.array([bool(getattr(C, sw)) for sw in mine.AUGMENT_SWITCHES])
        out[key + "_out"], out[key + "_out_boxes"] = res, box_rows(rd["bboxes"])
        out[key + "_out_classes"] = np.array([b["class"] for b in rd["bboxes"]] or [""], dtype="U8")
        out[key + "_wh"] = np.array([rd["width"], rd["height"]], dtype=np.int64)
        out[key + "_after"] = np.int64(np.random.randint(0, 2 ** 31 - 1))
    out["n_aug"] = np.int64(n_aug)
    np.savez_compressed(os.path.join(G.OUT, "augment.npz"), **out)
    print("augment.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
