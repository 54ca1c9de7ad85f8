"""Shared inputs of the affine-warp parity tests (host restatement: tests/test_augmentation.py; device kernel:
tests/test_gpu_resize.py): the matrices augmentation.py builds (centre rotation with the canvas grown and shifted, |f| shear on
a widened canvas) and a few others; sources whose warped footprint leaves the destination, destinations that reach outside the
source (border 0), 1-3 pixel sources.  Matrices come from oracle.imglib, never from the product."""
import numpy as np

from oracle import imglib

CASES = ["identity", "shift", "rot+2.7", "rot-3", "rot31_grow", "shear0.3", "shear0.07", "scale_down", "one_channel", "px1", "px2x3", "px3x1",
         "singular"]


def make(case):
    """-> (uint8 image, 2x3 float64 matrix, (dst_w, dst_h))."""
    rs = np.random.RandomState(len(case))
    h, w = 157, 203
    img = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
    if case == "identity":
        return img, np.array([[1, 0, 0], [0, 1, 0]], float), (w, h)
    if case == "shift":
        return img, np.array([[1, 0, 7.37], [0, 1, -3.21]], float), (w + 9, h + 4)
    if case.startswith("rot"):
        ang = {"rot+2.7": 2.7, "rot-3": -3.0, "rot31_grow": 31.0}[case]
        m = imglib.get_rotation_matrix_2d((w // 2, h // 2), ang, 1.0)
        c, s = abs(m[0, 0]), abs(m[0, 1])
        nw, nh = int(h * s + w * c), int(h * c + w * s)          # augmentation.py:176-182: the canvas grows, the centre moves
        m[0, 2] += nw / 2 - w // 2
        m[1, 2] += nh / 2 - h // 2
        return img, m, (nw, nh)
    if case.startswith("shear"):
        f = float(case[5:])
        return img, np.array([[1, f, 0], [0, 1, 0]], float), (int(w + f * h), h)
    if case == "scale_down":
        return img, np.array([[0.61, 0.02, 3.3], [-0.04, 0.57, 11.0]], float), (150, 120)
    if case == "one_channel":
        return img[:, :, :1].copy(), imglib.get_rotation_matrix_2d((101, 78), 1.3, 1.0), (w, h)
    if case == "px1":
        return img[:1, :1].copy(), np.array([[2.5, 0.3, 1.2], [-0.2, 3.0, 0.7]], float), (9, 11)
    if case == "px2x3":
        return img[:2, :3].copy(), imglib.get_rotation_matrix_2d((1, 1), 17.0, 2.0), (8, 7)
    if case == "px3x1":
        return img[:3, :1].copy(), np.array([[1.0, 0.5, 2.0], [0.0, 1.0, 0.4]], float), (7, 6)
    if case == "singular":                                           # det = 0: OpenCV inverts with D = 0 (a zero matrix), no error
        return img[:20, :30].copy(), np.array([[1.0, 2.0, 3.0], [2.0, 4.0, 1.0]], float), (16, 12)
    raise KeyError(case)
