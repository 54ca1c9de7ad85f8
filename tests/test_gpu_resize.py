"""csrc/resize.hip (radnet_resize_bicubic_u8, through the C ABI) against oracle/resize.py -- the independent two-pass NumPy
restatement of OpenCV's 8-bit INTER_CUBIC definition -- BIT FOR BIT (byte output).  Parity against cv2 itself is unpinned
(OpenCV absent here, no resized fixture in the reference: oracle/resize.py header); what is pinned is that the kernel on the
predict path and the tile feed (RADNet.py:53-74, utils.py:442-446) computes exactly the documented algorithm: half-pixel
mapping, replicated borders, 11-bit coefficient rounding, the single >> 22 rounding shift, saturation."""
import numpy as np
import pytest

from oracle import resize as OR

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ctx():
    from radnet_hip import lib as L
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test on a machine without a GPU")
    c = L.Context(0)
    yield c
    c.close()


def gpu_resize(ctx, img, new_w, new_h):
    src = torch.from_numpy(np.ascontiguousarray(img)).cuda()
    dst = torch.zeros(new_h, new_w, img.shape[2], dtype=torch.uint8, device="cuda")
    ctx.call("radnet_resize_bicubic_u8", src, img.shape[0], img.shape[1], dst, new_h, new_w, img.shape[2])
    ctx.sync()
    return dst.cpu().numpy()


CASES = [  # (src_h, src_w, dst_h, dst_w, channels, kind)
    (2048, 2048, 600, 600, 3, "noise"),        # BASELINE cfg 3: tile -> img_size 600
    (2048, 2048, 1000, 1000, 3, "noise"),      # cfg 3 at img_size 1000
    (1500, 2000, 600, 800, 3, "noise"),        # 2000x1500 frame -> 800x600
    (1200, 2000, 600, 1000, 3, "smooth"),      # cfg 2 panel geometry, smooth content (weights matter, not noise)
    (600, 900, 600, 900, 3, "noise"),          # identity size
    (300, 500, 600, 1000, 3, "noise"),         # 2x up
    (211, 317, 389, 701, 3, "noise"),          # non-integer up-scale
    (701, 389, 97, 53, 3, "noise"),            # strong non-integer down-scale
    (1, 1, 5, 7, 3, "noise"),                  # 1-pixel source: all taps clamp to it
    (2, 3, 9, 11, 3, "noise"),
    (3, 1, 8, 6, 1, "noise"),                  # 1 px wide, one channel
    (5, 2, 3, 9, 4, "noise"),                  # 2 px wide, four channels
    (64, 64, 64, 64, 3, "extremes"),           # 0/255 checkerboard: overshoot must saturate, identity exact
    (64, 64, 150, 150, 3, "extremes"),         # up-scaled 0/255 edges: negative lobes -> clamp at 0 and 255
]


def make(kind, h, w, c, seed):
    rs = np.random.RandomState(seed)
    if kind == "noise":
        return rs.randint(0, 256, (h, w, c)).astype(np.uint8)
    if kind == "smooth":
        y, x = np.mgrid[0:h, 0:w]
        base = 127.5 + 100.0 * np.sin(x / 37.0) * np.cos(y / 23.0)
        return np.clip(base[:, :, None] + np.arange(c)[None, None, :] * 9.0, 0, 255).astype(np.uint8)
    y, x = np.mgrid[0:h, 0:w]
    return (((x // 3 + y // 5) % 2) * 255).astype(np.uint8)[:, :, None].repeat(c, 2)


@pytest.mark.parametrize("case", CASES, ids=["%dx%d_to_%dx%d_c%d_%s" % c for c in CASES])
def test_resize_bicubic_bit_exact_vs_oracle(ctx, case):
    sh, sw, dh, dw, ch, kind = case
    img = make(kind, sh, sw, ch, seed=sh * 7 + dw)
    got = gpu_resize(ctx, img, dw, dh)
    ref = OR.resize_bicubic_u8(img, dw, dh)
    assert got.shape == ref.shape and got.dtype == np.uint8
    bad = np.argwhere(got != ref)
    assert len(bad) == 0, "%d differing bytes, first at %s: gpu %d oracle %d" % (len(bad), bad[0], got[tuple(bad[0])], ref[tuple(bad[0])])
    if (sh, sw) == (dh, dw):
        assert np.array_equal(got, img)                       # identity: weights are exactly (0, 2048, 0, 0)
    if kind == "extremes" and (sh, sw) != (dh, dw):
        assert got.min() == 0 and got.max() == 255            # saturation exercised


def test_borders_replicate(ctx):
    """Constant rows/columns at the border stay constant under up-scaling only if out-of-range taps replicate the edge."""
    img = np.full((40, 50, 3), 90, np.uint8)
    img[:, :4] = 200
    img[:3, :] = 17
    got = gpu_resize(ctx, img, 173, 131)
    assert np.array_equal(got, OR.resize_bicubic_u8(img, 173, 131))
    assert (got[0, 20:] == 17).all() and (got[40:, 0] == 200).all()


from warp_cases import CASES as WARP_CASES, make as warp_case  # noqa: E402


@pytest.mark.parametrize("case", WARP_CASES)
def test_warp_affine_device_equals_oracle(case):
    """radnet_warp_affine_u8 (the rotation / shear warp of the train-time augmentation on the device) against
    oracle.imglib.warp_affine_u8 -- the independent per-pixel restatement of OpenCV's 8-bit bilinear warpAffine (10-bit
    coordinate tables, 1/32-pixel fractions, 15-bit weights, constant border; parity vs cv2 itself unpinned: OpenCV is not
    importable) -- bit for bit, and against the host form the feed uses without a device."""
    from faster_rcnn import augmentation as A
    from faster_rcnn.RADNet import warp_affine_device
    from oracle import imglib
    img, m, ds = warp_case(case)
    ref = imglib.warp_affine_u8(img, m, ds)
    got = warp_affine_device(img, m, ds)
    assert got.shape == ref.shape and got.dtype == np.uint8
    assert np.array_equal(got, ref), int((got != ref).sum())
    assert np.array_equal(A.warp_affine_u8(img, m, ds), ref)
