"""Train-time augmentation of a tile and its boxes: the reference's faster_rcnn/augmentation.py surface
-- same function names, argument order, (img, bboxes) returns, draws from the random stream in the same order -- on
NumPy only.  The reference leans on OpenCV and scikit-image for five things; neither is part of this build, so each has a
restatement of the library's documented algorithm here, and the status of every function is:

  pinned by the reference's own outputs (tests/golden/augment.npz, brightness.npz, tile_feed.json; tools/gen_golden_augment.py)
      clip_box, strap_img, random_crop, brightness, get_truncated_normal (SciPy, present), and -- with the five library calls
      below replaced by THIS module's restatements while the reference's functions run -- everything around them:
      the order and number of draws, angle / shear / amount parameters, the rotation matrix shift, output sizes, the
      corner arithmetic of the boxes, strapping, clipping, int / ceil rounding, the noise dispatch, background masks.
  parity unpinned (library semantics restated, nothing here can check them against the library)
      flip_u8 (cv2.flip: an index reversal, no arithmetic), rotation_matrix_2d (cv2.getRotationMatrix2D),
      warp_affine_u8 (cv2.warpAffine, INTER_LINEAR + BORDER_CONSTANT 0: 10-bit fixed-point source coordinates, 1/32-pixel
      fractions, 15-bit bilinear weights), rescale_intensity (skimage.exposure), random_noise / img_as_ubyte (skimage.util).
      scikit-image draws its noise field from a generator of its own that the reference never seeds, so the reference's
      noisy pixels are not reproducible even by the reference: only the distribution is stated.  `noise_rng` makes ours
      reproducible for tests.

rng: NumPy's global stream by default (the reference's), or a RandomState (the feed's private stream, data_feed.py).
"""
import copy
import math

import numpy as np

AUGMENT_SWITCHES = ("use_horizontal_flips", "use_vertical_flips", "use_90_rotations", "use_rotations", "use_shear",
                    "use_brightness", "use_noise")          # augmentation.py:495-518, in draw order


# ---- library restatements -------------------------------------------------------------------------------------------

def flip_u8(img, code):
    """cv2.flip: code 0 reverses rows, > 0 reverses columns, < 0 both."""
    if code == 0:
        return np.ascontiguousarray(img[::-1])
    return np.ascontiguousarray(img[:, ::-1] if code > 0 else img[::-1, ::-1])


def rotation_matrix_2d(center, angle, scale):
    """cv2.getRotationMatrix2D: counter-clockwise degrees about `center` in image coordinates (y down), float64 2x3."""
    a = math.radians(angle)
    al, be = scale * math.cos(a), scale * math.sin(a)
    cx, cy = float(center[0]), float(center[1])
    return np.array([[al, be, (1.0 - al) * cx - be * cy], [-be, al, be * cx + (1.0 - al) * cy]], dtype=np.float64)


def _sat_i32(v):
    return np.clip(np.rint(v), -2 ** 31, 2 ** 31 - 1).astype(np.int64)          # saturate_cast<int>(double): round half to even


def warp_tables(mat, dsize):
    """The inverse map of cv2.warpAffine in 10-bit fixed point, as OpenCV precomputes it: per destination column
    adelta[x] = round(A00 x 1024), bdelta[x] = round(A10 x 1024); per destination row x0[y] = round((A01 y + A02) 1024) + 16,
    y0[y] = round((A11 y + A12) 1024) + 16 -- A the inverse of `mat`, 16 = half a 1/32-pixel step.  Destination (x, y) then reads
    the source at ((x0[y] + adelta[x]) >> 5, (y0[y] + bdelta[x]) >> 5) in 1/32 pixels.  int64 arrays (every entry fits int32)."""
    dw, dh = int(dsize[0]), int(dsize[1])
    m = np.array(mat, dtype=np.float64).reshape(2, 3).copy()
    det = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    det = 1.0 / det if det != 0 else 0.0
    a11, a22 = m[1, 1] * det, m[0, 0] * det
    m[0, 0], m[0, 1], m[1, 0], m[1, 1] = a11, -m[0, 1] * det, -m[1, 0] * det, a22
    b1 = -m[0, 0] * m[0, 2] - m[0, 1] * m[1, 2]
    b2 = -m[1, 0] * m[0, 2] - m[1, 1] * m[1, 2]
    m[0, 2], m[1, 2] = b1, b2
    ab_scale, half_step = 1024.0, 16
    xs, ys = np.arange(dw, dtype=np.float64), np.arange(dh, dtype=np.float64)
    adelta, bdelta = _sat_i32(m[0, 0] * xs * ab_scale), _sat_i32(m[1, 0] * xs * ab_scale)
    x0 = _sat_i32((m[0, 1] * ys + m[0, 2]) * ab_scale) + half_step
    y0 = _sat_i32((m[1, 1] * ys + m[1, 2]) * ab_scale) + half_step
    return adelta, bdelta, x0, y0


def warp_affine_u8(src, mat, dsize):
    """cv2.warpAffine(src, mat, dsize) with its defaults (INTER_LINEAR, BORDER_CONSTANT, border value 0) for uint8 HWC.
    `mat` maps source to destination and is inverted first (warp_tables); the four neighbours of the 1/32-pixel source position
    (0 outside the image) are mixed with weights 32 (32 - fx)(32 - fy) ... that sum to 2^15, rounded: (sum + 2^14) >> 15.
    The device form of the same arithmetic is radnet_warp_affine_u8 (RADNet.warp_affine_device), bit-identical."""
    dw, dh = int(dsize[0]), int(dsize[1])
    adelta, bdelta, x0, y0 = warp_tables(mat, dsize)
    X = (x0[:, None] + adelta[None, :]) >> 5
    Y = (y0[:, None] + bdelta[None, :]) >> 5
    sx, sy = np.clip(X >> 5, -32768, 32767), np.clip(Y >> 5, -32768, 32767)
    fx, fy = X & 31, Y & 31
    sh, sw = src.shape[:2]
    ch = src.size // (sh * sw)
    # source with a one-pixel border of zeros: a tap outside the image (clamped to the border) reads the constant 0
    padded = np.zeros((sh + 2, sw + 2, ch), dtype=np.int32)
    padded[1:-1, 1:-1] = src.reshape(sh, sw, ch)
    flat = padded.reshape(-1, ch)
    y0, y1 = np.clip(sy, -1, sh) + 1, np.clip(sy + 1, -1, sh) + 1
    x0, x1 = np.clip(sx, -1, sw) + 1, np.clip(sx + 1, -1, sw) + 1
    pitch = sw + 2
    fx32, fy32 = fx.astype(np.int32), fy.astype(np.int32)
    w00, w01 = (32 * (32 - fx32) * (32 - fy32))[..., None], (32 * fx32 * (32 - fy32))[..., None]
    w10, w11 = (32 * (32 - fx32) * fy32)[..., None], (32 * fx32 * fy32)[..., None]
    acc = flat[y0 * pitch + x0] * w00 + flat[y0 * pitch + x1] * w01 + flat[y1 * pitch + x0] * w10 + flat[y1 * pitch + x1] * w11
    out = ((acc + (1 << 14)) >> 15).astype(np.uint8)                  # weights sum to 2^15: the result stays in 0..255
    return out.reshape((dh, dw) + src.shape[2:])


def rescale_intensity(img, in_range):
    """skimage.exposure.rescale_intensity(img, in_range=(lo, hi)) for a uint8 image with the default out_range ('dtype' =
    0..255): clip to the range, stretch to 0..255 in float64, truncate to uint8."""
    lo, hi = float(in_range[0]), float(in_range[1])
    f = np.clip(img.astype(np.float64), lo, hi)
    if lo != hi:
        f = (f - lo) / (hi - lo)
    return np.asarray(f * 255.0, dtype=np.uint8)


def random_noise(img, mode, noise_rng=None, **kw):
    """skimage.util.random_noise(img, mode=..., clip=True, ...) for uint8 input: works on img / 255 in float64 and returns
    float64 in [0, 1].  'gaussian': + N(mean, var); 'poisson': Poisson(img * v) / v with v = the number of distinct values
    rounded up to a power of two; 's&p': a pixel is replaced with probability `amount`, by 1 with probability
    `salt_vs_pepper`, else by 0.  The field comes from `noise_rng` (a numpy Generator; a fresh unseeded one, like
    scikit-image's, when None)."""
    g = np.random.default_rng() if noise_rng is None else noise_rng
    f = img.astype(np.float64) / 255.0
    if mode == "gaussian":
        out = f + g.normal(kw.get("mean", 0.0), kw.get("var", 0.01) ** 0.5, f.shape)
    elif mode == "poisson":
        v = 2.0 ** np.ceil(np.log2(len(np.unique(f))))
        out = g.poisson(f * v) / float(v)
    elif mode == "s&p":
        out = f.copy()
        hit = g.random(f.shape) <= kw.get("amount", 0.05)
        salt = g.random(f.shape) <= kw.get("salt_vs_pepper", 0.5)
        out[hit & salt] = 1.0
        out[hit & ~salt] = 0.0
    else:
        raise ValueError("random_noise: mode %r is not one the reference uses" % (mode,))
    return np.clip(out, 0.0, 1.0)


def img_as_ubyte(f):
    """skimage.util.img_as_ubyte for float input in [0, 1]: round(f * 255) to nearest even, clipped."""
    return np.clip(np.rint(f * 255.0), 0, 255).astype(np.uint8)


# ---- the reference's functions ---------------------------------------------------------------------------------------

def get_truncated_normal(mean=0, sd=1, low=0, upp=1):
    """augmentation.py:14-15 (a frozen scipy.stats.truncnorm)."""
    from scipy.stats import truncnorm
    return truncnorm((low - mean) / sd, (upp - mean) / sd, loc=mean, scale=sd)


def strap_img(img):
    """augmentation.py:17-31: bounding rows / columns of the pixels whose channel 1 is non-zero (of the finite ones when the
    image holds non-finite values).  The callers slice [min:max], i.e. they drop the last such row and column."""
    ch = img[:, :, 1]
    finite = np.isfinite(ch)
    r, c = np.nonzero(ch) if finite.all() else np.nonzero(finite)
    return r.min(), r.max(), c.min(), c.max()


def clip_box(bbox, img_box, alpha):
    """augmentation.py:33-83: clip N x (4+) boxes to img_box = (x1, y1, x2, y2); keep those that touch it and lose less than
    1 - alpha of their area.  Returns (kept clipped boxes, keep mask)."""
    outside = (bbox[:, 0] > img_box[2]) | (bbox[:, 2] < img_box[0]) | (bbox[:, 1] > img_box[3]) | (bbox[:, 3] < img_box[1])
    area = (bbox[:, 2] - bbox[:, 0]) * (bbox[:, 3] - bbox[:, 1])
    lo = np.maximum(bbox[:, :2], np.asarray(img_box[:2]))
    hi = np.minimum(bbox[:, 2:4], np.asarray(img_box[2:4]))
    clipped = np.hstack((lo, hi, bbox[:, 4:]))
    lost = (area - (clipped[:, 2] - clipped[:, 0]) * (clipped[:, 3] - clipped[:, 1])) / area
    mask = ~outside & (lost < (1 - alpha))
    return clipped[mask, :], mask


def _boxes_array(bboxes):
    return np.array([[b["x1"], b["y1"], b["x2"], b["y2"]] for b in bboxes])


def _write_back(bboxes, arr, col_min, row_min):
    """The reference's rounding of transformed boxes into the cropped frame: floor-to-int for the low corner, ceil for the
    high one (augmentation.py:227-230, 263-266, 295-298)."""
    for b, (x1, y1, x2, y2) in zip(bboxes, arr[:, :4]):
        b["x1"], b["y1"] = int(x1 - col_min), int(y1 - row_min)
        b["x2"], b["y2"] = int(math.ceil(x2 - col_min)), int(math.ceil(y2 - row_min))


def horizontal_flip(img, bboxes, verbose=False):
    """augmentation.py:85-99."""
    cols = img.shape[1]
    for b in bboxes:
        b["x1"], b["x2"] = cols - b["x2"], cols - b["x1"]
    return flip_u8(img, 1), bboxes


def vertical_flip(img, bboxes, verbose=False):
    """augmentation.py:101-115."""
    rows = img.shape[0]
    for b in bboxes:
        b["y1"], b["y2"] = rows - b["y2"], rows - b["y1"]
    return flip_u8(img, 0), bboxes


def ninety_degree_rotation(img, bboxes, verbose=False, rng=np.random):
    """augmentation.py:117-156: one draw (choice over 90 / 180 / 270)."""
    rows, cols = img.shape[:2]
    angle = rng.choice([90, 180, 270], 1)[0]
    if angle == 180:
        img = flip_u8(img, -1)
    else:
        img = flip_u8(np.transpose(img, (1, 0, 2)), 0 if angle == 270 else 1)
    for b in bboxes:
        x1, x2, y1, y2 = b["x1"], b["x2"], b["y1"], b["y2"]
        if angle == 270:
            b["x1"], b["x2"], b["y1"], b["y2"] = y1, y2, cols - x2, cols - x1
        elif angle == 180:
            b["x1"], b["x2"], b["y1"], b["y2"] = cols - x2, cols - x1, rows - y2, rows - y1
        else:
            b["x1"], b["x2"], b["y1"], b["y2"] = rows - y2, rows - y1, x1, x2
    return img, bboxes


def any_degree_rotation(img, bboxes, verbose=False, rng=np.random, warp=None):
    """augmentation.py:158-232: one draw, U(-3, 3) degrees about (w // 2, h // 2); the canvas grows to hold the rotated image,
    each box becomes the axis-aligned hull of its four rotated corners, the result is strapped to its non-black extent and
    the boxes clipped to that (dropped when less than half is left)."""
    arr = _boxes_array(bboxes)
    h, w = img.shape[:2]
    angle = rng.uniform(-3.0, 3.0)
    cx, cy = w // 2, h // 2
    mat = rotation_matrix_2d((cx, cy), angle, 1.0)
    c, s = abs(mat[0, 0]), abs(mat[0, 1])
    new_w, new_h = int(h * s + w * c), int(h * c + w * s)
    mat[0, 2] += new_w / 2 - cx
    mat[1, 2] += new_h / 2 - cy
    img = (warp or warp_affine_u8)(img, mat, (new_w, new_h))
    if arr.ndim == 2:
        x1, y1, x2, y2 = (arr[:, i] for i in range(4))
        px = np.stack((x1, x1 + (x2 - x1), x1, x2), 1)                    # corner order of the reference: tl, tr, bl, br
        py = np.stack((y1, y1, y1 + (y2 - y1), y2), 1)
        homog = np.stack((px.ravel(), py.ravel(), np.ones(px.size, dtype=px.dtype)), 1)
        moved = np.dot(mat, homog.T).T.reshape(-1, 4, 2)                   # the reference's product, same operand shapes
        rx, ry = moved[:, :, 0], moved[:, :, 1]
        arr = np.stack((rx.min(1), ry.min(1), rx.max(1), ry.max(1)), 1)
    row_min, row_max, col_min, col_max = strap_img(img)
    img = img[row_min:row_max, col_min:col_max, :]
    if arr.ndim == 2:
        arr, mask = clip_box(arr, [col_min, row_min, col_max, row_max], 0.5)
        bboxes = [b for b, m in zip(bboxes, mask) if m]
        _write_back(bboxes, arr, col_min, row_min)
    return img, bboxes


def shear(img, bboxes, verbose=False, rng=np.random, warp=None):
    """augmentation.py:234-271: one draw, U(-0.3, 0.3); x' = x + |f| y on a canvas widened by |f| h (negative factors: the
    same between two horizontal flips); box x coordinates move by int(|f| y) of their own corner, then the strap offset."""
    f = rng.uniform(-0.3, 0.3)
    if f < 0.0:
        img, bboxes = horizontal_flip(img, bboxes)
    h, w = img.shape[:2]
    arr = _boxes_array(bboxes)
    if arr.ndim == 2:
        arr[:, [0, 2]] += (arr[:, [1, 3]] * abs(f)).astype(int)
    img = (warp or warp_affine_u8)(img, np.array([[1, abs(f), 0], [0, 1, 0]], dtype=np.float64), (int(w + abs(f * h)), h))
    row_min, row_max, col_min, col_max = strap_img(img)
    img = img[row_min:row_max, col_min:col_max, :]
    if arr.ndim == 2:
        _write_back(bboxes, arr, col_min, row_min)
    if f < 0.0:
        img, bboxes = horizontal_flip(img, bboxes)
    return img, bboxes


def random_crop(img, bboxes, verbose=False, rng=np.random):
    """augmentation.py:273-300 (not called by augment()): four draws -- width and height in [0.4, 0.8) of the image, then
    the corner."""
    h, w = img.shape[:2]
    new_w = rng.randint(int(0.4 * w), int(0.8 * w))
    new_h = rng.randint(int(0.4 * h), int(0.8 * h))
    col_min = rng.randint(0, w - new_w)
    row_min = rng.randint(0, h - new_h)
    col_max, row_max = col_min + new_w, row_min + new_h
    img = img[row_min:row_max, col_min:col_max, :]
    arr, mask = clip_box(_boxes_array(bboxes), [col_min, row_min, col_max, row_max], 0.5)
    bboxes = [b for b, m in zip(bboxes, mask) if m]
    _write_back(bboxes, arr, col_min, row_min)
    return img, bboxes


def brightness(img, bboxes, verbose=False, rng=np.random):
    """augmentation.py:303-333: pixels that are exactly 0 (in a channel) are background and stay 0; the rest moves darker
    with probability p = (mean - 75) / 105 by U[0,1) * (mean - 75), else lighter by U[0,1) * (180 - mean), in float32,
    clipped to [0, 255] and truncated to uint8.  Images whose foreground mean is outside (75, 180) come back unchanged
    WITHOUT drawing; two draws otherwise."""
    background = img == 0
    f = img.astype("float32")
    lo, hi = 75, 180
    avg = f[~background].mean()
    if avg <= lo or avg >= hi:
        return f.astype("uint8"), bboxes
    if rng.random() < (avg - lo) / (hi - lo):
        f -= rng.random() * (avg - lo)
    else:
        f += rng.random() * (hi - avg)
    out = np.clip(f, 0, 255).astype("uint8")
    out[background] = 0
    return out, bboxes


def contrast(img, bboxes, verbose=False, rng=np.random):
    """augmentation.py:335-351: two draws; intensities between 75 u and 180 + 75 v are stretched to 0..255."""
    lo = 75 * rng.random()
    hi = (255 - 180) * rng.random() + 180
    return rescale_intensity(img, (lo, hi)), bboxes


def _noisy(img, img_type, mode, noise_rng, **kw):
    """Shared tail of the three noise functions (augmentation.py:362-397, 408-441, 449-478): grey image types get ONE noise
    plane (from channel 0) copied to all three channels, others are treated per channel; background (exact zeros of the
    input) is restored afterwards."""
    if "grey" in img_type:
        plane = img_as_ubyte(random_noise(img[:, :, 0], mode, noise_rng, **kw))
        plane[img[:, :, 0] == 0] = 0
        return np.repeat(plane[:, :, None], 3, axis=2)
    out = img_as_ubyte(random_noise(img, mode, noise_rng, **kw))
    out[img == 0] = 0
    return out


def salt_and_pepper_noise(img, bboxes, img_type, verbose=False, rng=np.random, noise_rng=None):
    """augmentation.py:353-397: amount = U(0.01, 0.3) (one draw), salt share from a normal (0.5, 0.1) truncated to [0, 1]
    (SciPy's rvs on the same stream)."""
    amount = (0.3 - 0.01) * rng.random() + 0.01
    svp = get_truncated_normal(mean=0.5, sd=0.1, low=0, upp=1).rvs(size=1, random_state=None if rng is np.random else rng)[0]
    return _noisy(img, img_type, "s&p", noise_rng, amount=amount, salt_vs_pepper=svp), bboxes


def gaussian_noise(img, bboxes, img_type, verbose=False, rng=np.random, noise_rng=None):
    """augmentation.py:399-441: mean = U(-0.05, 0.05), var = U(0.001, 0.01): two draws."""
    mean = (0.05 + 0.05) * rng.random() - 0.05
    var = (0.01 - 0.001) * rng.random() + 0.001
    return _noisy(img, img_type, "gaussian", noise_rng, mean=mean, var=var), bboxes


def poisson_noise(img, bboxes, img_type, verbose=False, rng=np.random, noise_rng=None):
    """augmentation.py:443-478: no draw from the shared stream."""
    return _noisy(img, img_type, "poisson", noise_rng), bboxes


def augment(img_data, img, config, augment=True, verbose=False, rng=np.random, noise_rng=None, warp=None):
    """augmentation.py:481-533: a deep copy of img_data with the boxes of the augmented image, and the image.  One coin per
    enabled switch, in the reference's order and with its thresholds (shear: 0.25, the others 0.5); the noise family
    draws a second time for which of its four members runs.  warp: the affine warp of the rotation and the shear -- None = the
    NumPy form above, or the bit-identical device kernel (RADNet.warp_affine_device; what TileFeed uses with the device resize)."""
    for k in ("filepath", "bboxes", "width", "height"):
        assert k in img_data
    out = copy.deepcopy(img_data)
    if augment:
        C, boxes = config, out["bboxes"]
        if C.use_horizontal_flips and rng.random() < 0.5:
            img, boxes = horizontal_flip(img, boxes, verbose)
        if C.use_vertical_flips and rng.random() < 0.5:
            img, boxes = vertical_flip(img, boxes, verbose)
        if C.use_90_rotations and rng.random() < 0.5:
            img, boxes = ninety_degree_rotation(img, boxes, verbose, rng=rng)
        if C.use_rotations and rng.random() < 0.5:
            img, boxes = any_degree_rotation(img, boxes, verbose, rng=rng, warp=warp)
        if C.use_shear and rng.random() < 0.25:
            img, boxes = shear(img, boxes, verbose, rng=rng, warp=warp)
        if C.use_brightness and rng.random() < 0.5:
            img, boxes = brightness(np.ascontiguousarray(img), boxes, verbose, rng=rng)
        if C.use_noise and rng.random() < 0.5:
            which = rng.randint(0, 4)
            kind = C.img_types[0]
            if which == 0:
                img, boxes = salt_and_pepper_noise(img, boxes, kind, verbose, rng=rng, noise_rng=noise_rng)
            elif which == 1:
                img, boxes = gaussian_noise(img, boxes, kind, verbose, rng=rng, noise_rng=noise_rng)
            elif which == 2:
                img, boxes = poisson_noise(img, boxes, kind, verbose, rng=rng, noise_rng=noise_rng)
            else:
                img, boxes = contrast(img, boxes, verbose, rng=rng)
        out["bboxes"] = boxes
        out["width"], out["height"] = img.shape[1], img.shape[0]
    return out, np.ascontiguousarray(img)
