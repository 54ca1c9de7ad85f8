"""ORACLE (test infrastructure, not product): restatement of the five image-library calls the reference's augmentation
makes -- cv2.getRotationMatrix2D / cv2.warpAffine (augmentation.py:174,184,257), skimage.exposure.rescale_intensity
(augmentation.py:343), skimage.util.random_noise / img_as_ubyte (augmentation.py:367-469) -- written from the libraries'
published algorithms, scalar / row-at-a-time, WITHOUT importing anything from the product package.  Only tests/ may import it.

PARITY UNPINNED against the libraries themselves: OpenCV and scikit-image are third-party dependencies of the reference with no
pinned version, neither is importable here, and the reference holds no fixture of their outputs.  What this file pins is the
product's two implementations (faster_rcnn.augmentation.warp_affine_u8 on the host, radnet_warp_affine_u8 on the device)
against an independent reading of the same definitions (tests/test_augmentation.py, tests/test_gpu_resize.py).

cv2.warpAffine for CV_8U, flags = INTER_LINEAR, borderMode = BORDER_CONSTANT, borderValue = 0 (OpenCV imgproc, imgwarp.cpp:
`warpAffine` -> `WarpAffineInvoker` -> `remap` / `remapBilinear`):
  1. M (2x3, source -> destination) is inverted in double:  D = M00 M11 - M01 M10;  D = D != 0 ? 1/D : 0;
     A11 = M11 D, A22 = M00 D;  M00 = A11, M01 *= -D, M10 *= -D, M11 = A22;  b1 = -M00 M02 - M01 M12;  b2 = -M10 M02 - M11 M12.
  2. Fixed point: AB_BITS = 10 (AB_SCALE = 1024), INTER_BITS = 5 (32 sub-pixel positions), round_delta = AB_SCALE / 32 / 2 = 16.
     adelta[x] = saturate_cast<int>(M00 x AB_SCALE), bdelta[x] = saturate_cast<int>(M10 x AB_SCALE)   (cvRound: half to even)
     per row y:  X0 = saturate_cast<int>((M01 y + M02) AB_SCALE) + round_delta,  Y0 likewise with M11, M12
     per pixel:  X = (X0 + adelta[x]) >> (AB_BITS - INTER_BITS);  Y likewise;
                 sx = saturate_cast<short>(X >> INTER_BITS), sy likewise;  alpha = (Y & 31) * 32 + (X & 31).
  3. remapBilinear with the integer table BilinearTab_i[alpha][2][2] = saturate_cast<short>((1-fy|fy)(1-fx|fx) * 32768) where
     fx = (X & 31)/32, fy = (Y & 31)/32: for 1/32 fractions the four products are exact integers 32 (32-a)(32-b) .. that sum
     to 2^15, so the table's sum-correction step never fires.  Result = (sum of tap * weight + 2^14) >> 15 (FixedPtCast).
  4. BORDER_CONSTANT: a tap outside the source reads borderValue (0); a pixel whose 2x2 footprint lies wholly outside is the
     border value.
cv2.getRotationMatrix2D(center, angle, scale): angle in degrees, positive = counter-clockwise (origin top-left):
     a = scale cos(angle), b = scale sin(angle);  [[a, b, (1-a) cx - b cy], [-b, a, b cx + (1-a) cy]].
"""
import math

import numpy as np

AB_BITS = 10
INTER_BITS = 5
REMAP_COEF_BITS = 15


def _cv_round(v):
    """saturate_cast<int>(double) = cvRound (lrint: round half to even), saturated to the int32 range."""
    r = float(np.rint(v))
    return int(min(max(r, -2147483648.0), 2147483647.0))


def get_rotation_matrix_2d(center, angle, scale):
    rad = angle * (math.pi / 180.0)          # `angle *= CV_PI / 180`: the constant is folded first
    a = math.cos(rad) * scale
    b = math.sin(rad) * scale
    cx, cy = float(center[0]), float(center[1])
    return np.array([[a, b, (1.0 - a) * cx - b * cy], [-b, a, b * cx + (1.0 - a) * cy]], dtype=np.float64)


def invert_affine(M):
    m = [float(v) for v in np.asarray(M, dtype=np.float64).reshape(6)]
    D = m[0] * m[4] - m[1] * m[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m[4] * D, m[0] * D
    m[0] = A11
    m[1] *= -D
    m[3] *= -D
    m[4] = A22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2], m[5] = b1, b2
    return m


def warp_affine_u8(src, M, dsize):
    """cv2.warpAffine(src, M, dsize) with the defaults the reference uses (see module docstring).  src: HxW or HxWxC uint8;
    dsize = (width, height).  One destination pixel at a time."""
    src = np.asarray(src)
    assert src.dtype == np.uint8
    sh, sw = src.shape[:2]
    planes = src.reshape(sh, sw, -1).astype(np.int64)
    ch = planes.shape[2]
    dw, dh = int(dsize[0]), int(dsize[1])
    m = invert_affine(M)
    ab_scale = float(1 << AB_BITS)
    round_delta = (1 << AB_BITS) // (1 << INTER_BITS) // 2
    adelta = [_cv_round(m[0] * x * ab_scale) for x in range(dw)]
    bdelta = [_cv_round(m[3] * x * ab_scale) for x in range(dw)]
    out = np.zeros((dh, dw, ch), dtype=np.uint8)
    half = 1 << (REMAP_COEF_BITS - 1)

    def tap(y, x):
        if 0 <= y < sh and 0 <= x < sw:
            return planes[y, x]
        return np.zeros(ch, dtype=np.int64)          # BORDER_CONSTANT, borderValue 0

    for y in range(dh):
        X0 = _cv_round((m[1] * y + m[2]) * ab_scale) + round_delta
        Y0 = _cv_round((m[4] * y + m[5]) * ab_scale) + round_delta
        for x in range(dw):
            X = (X0 + adelta[x]) >> (AB_BITS - INTER_BITS)
            Y = (Y0 + bdelta[x]) >> (AB_BITS - INTER_BITS)
            sx = min(max(X >> INTER_BITS, -32768), 32767)
            sy = min(max(Y >> INTER_BITS, -32768), 32767)
            a, b = X & 31, Y & 31
            if sx >= sw or sx + 1 < 0 or sy >= sh or sy + 1 < 0:
                continue                              # wholly outside: border value
            w00, w01, w10, w11 = 32 * (32 - a) * (32 - b), 32 * a * (32 - b), 32 * (32 - a) * b, 32 * a * b
            acc = tap(sy, sx) * w00 + tap(sy, sx + 1) * w01 + tap(sy + 1, sx) * w10 + tap(sy + 1, sx + 1) * w11
            out[y, x] = (acc + half) >> REMAP_COEF_BITS
    return out.reshape((dh, dw) + src.shape[2:])


# ---- scikit-image ------------------------------------------------------------------------------------------------------

def rescale_intensity_u8(img, in_range):
    """skimage.exposure.rescale_intensity(img, in_range=(lo, hi)) for uint8 input, out_range = 'dtype' (0..255):
    image = clip(image, imin, imax); image = (image - imin) / (imax - imin) if imin != imax; return
    asarray(image * (omax - omin) + omin, dtype=uint8) -- float64 arithmetic, truncation by the final cast."""
    imin, imax = float(in_range[0]), float(in_range[1])
    out = np.empty(img.shape, dtype=np.uint8)
    flat_in, flat_out = img.reshape(-1), out.reshape(-1)
    for i in range(flat_in.size):
        v = min(max(float(flat_in[i]), imin), imax)
        if imin != imax:
            v = (v - imin) / (imax - imin)
        flat_out[i] = int(v * 255.0 + 0.0)
    return out


def img_as_float_u8(img):
    """skimage.util.img_as_float for uint8: divide by 255 in float64."""
    return img.astype(np.float64) / 255.0


def img_as_ubyte(f):
    """skimage.util.img_as_ubyte for float input in [0, 1] (dtype.convert, float -> uint8): rint(f * 255) clipped to 0..255."""
    return np.clip(np.rint(np.asarray(f, dtype=np.float64) * 255.0), 0, 255).astype(np.uint8)


def random_noise(img, mode, rng, **kw):
    """skimage.util.random_noise(img, mode, clip=True, **kw) for uint8 input, with the random field drawn from `rng` (a
    numpy Generator) in the order scikit-image draws it.  scikit-image itself seeds a fresh generator per call (seed=None), so
    the reference's noisy pixels are not reproducible even by the reference; only the arithmetic around the field is stated:
      gaussian: image + normal(mean, sqrt(var));   defaults mean = 0, var = 0.01
      poisson : vals = 2 ** ceil(log2(number of unique values)); poisson(image * vals) / vals
      s&p     : flipped = random(shape) <= amount; salted = random(shape) <= salt_vs_pepper;
                out[flipped & salted] = 1, out[flipped & ~salted] = 0;   defaults amount = 0.05, salt_vs_pepper = 0.5
    followed by clip to [0, 1].  Returns float64."""
    image = img_as_float_u8(img)
    if mode == "gaussian":
        noise = rng.normal(kw.get("mean", 0.0), kw.get("var", 0.01) ** 0.5, image.shape)
        out = image + noise
    elif mode == "poisson":
        vals = len(np.unique(image))
        vals = 2 ** np.ceil(np.log2(vals))
        out = rng.poisson(image * vals) / float(vals)
    elif mode == "s&p":
        out = image.copy()
        flipped = rng.random(image.shape) <= kw.get("amount", 0.05)
        salted = rng.random(image.shape) <= kw.get("salt_vs_pepper", 0.5)
        out[flipped & salted] = 1.0
        out[flipped & ~salted] = 0.0
    else:
        raise ValueError(mode)
    return np.clip(out, 0.0, 1.0)
