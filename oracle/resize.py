"""ORACLE (test infrastructure, not product): NumPy restatement of OpenCV's 8-bit `cv2.resize(..., INTER_CUBIC)`, the
resize the reference applies to every tile (RADNet.py:53-74 `format_img_size`; utils.py:442-446 in the tile generator).
Only tests/ may import this module.

PARITY UNPINNED against cv2 itself: OpenCV is a third-party dependency of the reference with no pinned version (the
`environment.yml` README.md:37 mentions is not in the tree), it is absent from this image, and the reference holds no resized
fixture.  This file restates the algorithm of OpenCV's imgproc `resize` for CV_8U / INTER_CUBIC as published
(modules/imgproc/src/resize.cpp, 3.x/4.x: `resizeGeneric_`, `HResizeCubic`, `VResizeCubic`, `FixedPtCast`):

  * scale = 1 / (dst / src)  in double (`inv_scale_x = dsize.width / ssize.width; scale_x = 1. / inv_scale_x`);
  * half-pixel centres: fx = float((dx + 0.5) * scale - 0.5); sx = floor(fx); fx -= sx   (fx is a float32);
  * cubic kernel with A = -0.75 evaluated in float32 (`interpolateCubic`), the fourth weight = 1 - w0 - w1 - w2;
  * weights -> 11-bit fixed point: saturate_cast<short>(w * 2048) = round-half-to-even;
  * horizontal pass: 4 taps at sx-1 .. sx+2, indices clamped to the row (replicated border), exact int32 sums;
  * vertical pass: 4 rows at sy-1 .. sy+2, clamped; (sum + 2^21) >> 22, saturated to 0..255.

Known source of build-dependent differences in real OpenCV binaries (not modelled): the vectorised vertical pass
(`VResizeCubicVec_32s8u`) evaluates the same sum in float32 and rounds, which can differ by one grey level from the
fixed-point tail in rare pixels.  The product kernel (csrc/resize.hip) implements the fixed-point definition above and is
compared with this file bit for bit (tests/test_gpu_resize.py)."""
import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def _axis_tables(n_src, n_dst):
    """Per destination index: the 4 clamped source indices [n_dst][4] and the 4 fixed-point weights [n_dst][4] (int32)."""
    scale = 1.0 / (float(n_dst) / float(n_src))
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    x = (f - s.astype(np.float32)).astype(np.float32)
    A = np.float32(-0.75)
    one = np.float32(1.0)
    xp1 = x + one
    xm1 = one - x
    w0 = ((A * xp1 - np.float32(5) * A) * xp1 + np.float32(8) * A) * xp1 - np.float32(4) * A
    w1 = ((A + np.float32(2)) * x - (A + np.float32(3))) * x * x + one
    w2 = ((A + np.float32(2)) * xm1 - (A + np.float32(3))) * xm1 * xm1 + one
    w3 = one - w0 - w1 - w2
    w = np.stack([w0, w1, w2, w3], axis=1).astype(np.float32)
    wi = np.clip(np.rint(w * np.float32(COEF_SCALE)), -32768, 32767).astype(np.int32)     # cvRound: half to even
    idx = np.clip(s[:, None] + np.arange(-1, 3)[None, :], 0, n_src - 1)
    return idx, wi


def resize_bicubic_u8(img, new_w, new_h):
    """img: uint8 [h][w][c] (or [h][w]) -> uint8 [new_h][new_w][c].  Argument order of the size as cv2.resize's dsize."""
    img = np.asarray(img)
    if img.dtype != np.uint8:
        raise TypeError("resize_bicubic_u8: uint8 only (the 8-bit fixed-point path)")
    squeeze = img.ndim == 2
    if squeeze:
        img = img[:, :, None]
    h, w, c = img.shape
    xi, xw = _axis_tables(w, new_w)
    yi, yw = _axis_tables(h, new_h)
    src = img.astype(np.int32)
    # horizontal pass into int32 rows: [h][new_w][c]
    rows = np.zeros((h, new_w, c), dtype=np.int32)
    for t in range(4):
        rows += src[:, xi[:, t], :] * xw[None, :, t, None]
    # vertical pass
    acc = np.zeros((new_h, new_w, c), dtype=np.int32)
    for t in range(4):
        acc += rows[yi[:, t], :, :] * yw[:, t, None, None]
    out = np.clip((acc + (1 << (2 * COEF_BITS - 1))) >> (2 * COEF_BITS), 0, 255).astype(np.uint8)
    return out[:, :, 0] if squeeze else out


def resize_bicubic_u8_loops(img, new_w, new_h):
    """The same definition pixel by pixel in Python integers (tiny inputs only): a cross-check of the vectorised form."""
    img = np.asarray(img)
    h, w, c = img.shape
    xi, xw = _axis_tables(w, new_w)
    yi, yw = _axis_tables(h, new_h)
    out = np.zeros((new_h, new_w, c), dtype=np.uint8)
    for dy in range(new_h):
        for dx in range(new_w):
            for k in range(c):
                acc = 0
                for j in range(4):
                    row = 0
                    for i in range(4):
                        row += int(img[yi[dy, j], xi[dx, i], k]) * int(xw[dx, i])
                    acc += row * int(yw[dy, j])
                out[dy, dx, k] = min(max((acc + (1 << 21)) >> 22, 0), 255)
    return out
