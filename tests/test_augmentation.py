"""faster_rcnn/augmentation.py against tests/golden/augment.npz (tools/gen_golden_augment.py): outputs of
the reference's own augmentation module.  `pure` cases need no absent library and pin the function outright; `around` cases ran
the reference with its OpenCV / scikit-image calls bound to this repo's restatements, so they pin the reference's control
flow, draws, box arithmetic and rounding around those calls -- not the libraries' arithmetic (parity unpinned, stated in the
module header).  Everything is byte-exact, including the position of NumPy's global stream afterwards."""
import copy
import os

import numpy as np
import pytest

from faster_rcnn import augmentation as A
from faster_rcnn.config import Config

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "augment.npz"))
NOISE_SEED = int(G["noise_seed"])


def boxes_of(rows):
    return [{"class": "c", "x1": int(r[0]), "y1": int(r[1]), "x2": int(r[2]), "y2": int(r[3])} for r in rows]


def rows_of(bboxes):
    return np.array([[b["x1"], b["y1"], b["x2"], b["y2"]] for b in bboxes], dtype=np.float64).reshape(-1, 4)


def test_strap_img_pure():
    for k in range(int(G["n_strap"])):
        assert list(A.strap_img(G["strap%d_img" % k])) == list(G["strap%d_out" % k])


def test_random_crop_and_truncated_normal_pure():
    for k in range(int(G["n_crop"])):
        np.random.seed(300 + k)
        img, bb = A.random_crop(G["crop%d_img" % k].copy(), boxes_of(G["crop%d_boxes" % k]))
        assert np.array_equal(img, G["crop%d_out" % k]) and np.array_equal(rows_of(bb), G["crop%d_out_boxes" % k])
        assert np.random.randint(0, 2 ** 31 - 1) == int(G["crop%d_after" % k])
    np.random.seed(7)
    got = [A.get_truncated_normal(mean=0.5, sd=0.1, low=0, upp=1).rvs(size=1)[0] for _ in range(5)]
    assert np.array_equal(np.array(got), G["truncnorm"]) and np.random.randint(0, 2 ** 31 - 1) == int(G["truncnorm_after"])


SINGLES = {"rot": lambda i, b, t, g: A.any_degree_rotation(i, b), "shear": lambda i, b, t, g: A.shear(i, b),
           "contrast": lambda i, b, t, g: A.contrast(i, b), "sp": lambda i, b, t, g: A.salt_and_pepper_noise(i, b, t, noise_rng=g),
           "gauss": lambda i, b, t, g: A.gaussian_noise(i, b, t, noise_rng=g), "poisson": lambda i, b, t, g: A.poisson_noise(i, b, t, noise_rng=g)}


@pytest.mark.parametrize("name", sorted(SINGLES))
def test_single_functions_around_library_calls(name):
    for k in range(int(G["n_single"])):
        key = "%s%d" % (name, k)
        np.random.seed(500 + k)
        img, bb = SINGLES[name](G[key + "_img"].copy(), boxes_of(G[key + "_boxes"]), "grey_x" if int(G[key + "_grey"]) else "rgb",
                                np.random.default_rng(NOISE_SEED))
        assert img.shape == G[key + "_out"].shape and np.array_equal(img, G[key + "_out"]), key
        assert np.array_equal(rows_of(bb), G[key + "_out_boxes"]), key
        assert np.random.randint(0, 2 ** 31 - 1) == int(G[key + "_after"]), key


def test_augment_dispatcher_every_switch_same_draws_boxes_and_pixels():
    ran = set()
    for k in range(int(G["n_aug"])):
        key = "aug%d" % k
        C = Config()
        for sw, on in zip(A.AUGMENT_SWITCHES, G[key + "_switches"]):
            setattr(C, sw, bool(on))
        C.img_types = ["grey_x" if int(G[key + "_grey"]) else "rgb"]
        img = G[key + "_img"]
        data = {"filepath": "t%d.png" % k, "width": img.shape[1], "height": img.shape[0], "bboxes": boxes_of(G[key + "_boxes"])}
        before = copy.deepcopy(data)
        np.random.seed(900 + k)
        rd, res = A.augment(data, img.copy(), C, augment=True, noise_rng=np.random.default_rng(NOISE_SEED))
        assert data == before                                                     # the caller's record is not edited
        assert np.array_equal(res, G[key + "_out"]), key
        assert np.array_equal(rows_of(rd["bboxes"]), G[key + "_out_boxes"]), key
        assert [rd["width"], rd["height"]] == list(G[key + "_wh"]) == [res.shape[1], res.shape[0]]
        assert np.random.randint(0, 2 ** 31 - 1) == int(G[key + "_after"]), key
        ran.add(res.shape != img.shape)
    assert ran == {True, False}                                                   # both resized and same-size outcomes occurred
    # augment=False: a deep copy and the image, no draw
    np.random.seed(3)
    probe = np.random.RandomState(3).randint(0, 2 ** 31 - 1)
    rd, res = A.augment(data, img, Config(), augment=False)
    assert rd == data and rd is not data and np.array_equal(res, img) and np.random.randint(0, 2 ** 31 - 1) == probe


def test_private_stream_gives_the_same_result_as_the_global_one():
    key = "aug0"
    img = G[key + "_img"]
    C = Config()
    C.img_types = ["rgb"]
    data = {"filepath": "t.png", "width": img.shape[1], "height": img.shape[0], "bboxes": boxes_of(G[key + "_boxes"])}
    for seed in range(12):
        np.random.seed(seed)
        a = A.augment(data, img.copy(), C, noise_rng=np.random.default_rng(1))
        b = A.augment(data, img.copy(), C, rng=np.random.RandomState(seed), noise_rng=np.random.default_rng(1))
        assert a[0] == b[0] and np.array_equal(a[1], b[1])


def test_warp_affine_restatement_properties():
    """cv2.warpAffine restated (parity unpinned): identity and integer shifts are exact copies, outside is 0, a half-pixel
    shift averages neighbours with round-half-up, and a rotation moves a marked pixel where the matrix says."""
    rs = np.random.RandomState(0)
    img = rs.randint(0, 256, (23, 31, 3)).astype(np.uint8)
    ident = np.array([[1, 0, 0], [0, 1, 0]], dtype=np.float64)
    assert np.array_equal(A.warp_affine_u8(img, ident, (31, 23)), img)
    sh = A.warp_affine_u8(img, np.array([[1, 0, 4], [0, 1, 2]], dtype=np.float64), (40, 30))
    assert np.array_equal(sh[2:25, 4:35], img) and not sh[:2].any() and not sh[:, :4].any() and not sh[25:].any() and not sh[:, 35:].any()
    half = A.warp_affine_u8(img, np.array([[1, 0, 0.5], [0, 1, 0]], dtype=np.float64), (31, 23))
    want = (img[:, :-1].astype(int) + img[:, 1:].astype(int) + 1) >> 1
    assert np.array_equal(half[:, 1:], want)
    dot = np.zeros((41, 41, 3), np.uint8)
    dot[10, 30] = 255
    m = A.rotation_matrix_2d((20, 20), 90, 1.0)                   # counter-clockwise on the screen: (30, 10) -> (10, 10)
    out = A.warp_affine_u8(dot, m, (41, 41))
    assert out[10, 10, 0] == 255 and out.sum() == 255 * 3
    assert np.allclose(m @ np.array([30, 10, 1.0]), [10, 10])


def test_noise_and_contrast_restatements_statistics():
    img = np.full((200, 200, 3), 128, np.uint8)
    g = np.random.default_rng(5)
    sp = A.img_as_ubyte(A.random_noise(img, "s&p", g, amount=0.2, salt_vs_pepper=0.25))
    hit = sp != 128
    assert abs(hit.mean() - 0.2) < 0.01 and abs((sp[hit] == 255).mean() - 0.25) < 0.02 and set(np.unique(sp)) == {0, 128, 255}
    ga = A.random_noise(img, "gaussian", g, mean=0.02, var=0.004)
    assert abs(ga.mean() - (128 / 255 + 0.02)) < 2e-3 and abs(ga.std() - 0.004 ** 0.5) < 2e-3
    ramp = np.repeat(np.arange(200, dtype=np.uint8)[None, :, None], 200, 0).repeat(3, 2)
    po = A.random_noise(ramp, "poisson", g)                        # 200 distinct values -> v = 256
    assert abs(po.mean() - ramp.mean() / 255) < 2e-3 and np.all(po * 256 == np.rint(po * 256))
    ct = A.rescale_intensity(ramp, (50.0, 150.0))
    assert ct[0, 50, 0] == 0 and ct[0, 150, 0] == 255 and ct[0, 20, 0] == 0 and ct[0, 199, 0] == 255 and ct[0, 100, 0] == 127


# ---- library arithmetic against the independent restatement (oracle/imglib.py; parity vs cv2 / scikit-image unpinned) -------

from warp_cases import CASES as WARP_CASES, make as warp_case  # noqa: E402


@pytest.mark.parametrize("case", WARP_CASES)
def test_warp_affine_host_restatement_equals_oracle(case):
    """faster_rcnn.augmentation.warp_affine_u8 (vectorised, what TileFeed runs without a GPU resize hook) against
    oracle.imglib.warp_affine_u8 (one pixel at a time, written from OpenCV's published warpAffine / remapBilinear definition
    without the product module): bit for bit.  VERDICT r2: the device kernel used to be compared with this product function
    only, which itself had no checker."""
    from oracle import imglib
    img, m, ds = warp_case(case)
    ref = imglib.warp_affine_u8(img, m, ds)
    got = A.warp_affine_u8(img, m, ds)
    assert got.shape == ref.shape and got.dtype == np.uint8
    assert np.array_equal(got, ref), int((got != ref).sum())
    if case not in ("singular",):
        assert ref.any()


def test_rotation_matrix_equals_oracle():
    from oracle import imglib
    rs = np.random.RandomState(3)
    for _ in range(50):
        c = (float(rs.uniform(0, 500)), float(rs.uniform(0, 500)))
        ang, sc = float(rs.uniform(-180, 180)), float(rs.uniform(0.3, 2.0))
        assert np.array_equal(A.rotation_matrix_2d(c, ang, sc), imglib.get_rotation_matrix_2d(c, ang, sc))
    assert np.array_equal(A.rotation_matrix_2d((101, 78), 3, 1.0), imglib.get_rotation_matrix_2d((101, 78), 3, 1.0))


def test_contrast_and_noise_restatements_equal_oracle():
    from oracle import imglib
    rs = np.random.RandomState(12)
    img = rs.randint(0, 256, (23, 31, 3)).astype(np.uint8)
    for lo, hi in ((0, 255), (13.5, 201.25), (74.99, 180.0), (0.0, 254.999), (2.0, 198.0)):      # augmentation.py:343: lo in [0, 75), hi in [180, 255)
        assert np.array_equal(A.rescale_intensity(img, (lo, hi)), imglib.rescale_intensity_u8(img, (lo, hi))), (lo, hi)
    for mode, kw in (("gaussian", {}), ("gaussian", {"mean": 0.1, "var": 0.003}), ("poisson", {}), ("s&p", {}), ("s&p", {"amount": 0.2, "salt_vs_pepper": 0.8})):
        a = A.random_noise(img, mode, noise_rng=np.random.default_rng(77), **kw)
        b = imglib.random_noise(img, mode, np.random.default_rng(77), **kw)
        assert a.dtype == np.float64 and np.array_equal(a, b), (mode, kw)
        assert np.array_equal(A.img_as_ubyte(a), imglib.img_as_ubyte(b))
    assert np.array_equal(A.img_as_ubyte(np.array([0.0, 0.5 / 255, 1.5 / 255, 2.5 / 255, 1.0])), np.array([0, 0, 2, 2, 255], np.uint8))   # half to even
