// Bicubic resize of uint8 HWC panels on the device, standing in for cv2.resize(..., INTER_CUBIC) in
// RADNet.format_img_size (RADNet.py:53-74) and the tile generator (utils.py:442-446).
//
// PARITY UNPINNED: OpenCV is absent from this image and the reference ships no resized fixture, so this follows
// OpenCV's documented algorithm for 8-bit images (bicubic kernel a = -0.75, half-pixel centres
// src = (dst + 0.5) * scale - 0.5, replicated borders, 11-bit fixed-point coefficients, horizontal pass into
// 32-bit integers then vertical pass with a single rounding shift of 22 bits and saturation).  What CAN be checked is
// checked: bit-for-bit against the independent NumPy restatement of that algorithm in oracle/resize.py
// (tests/test_gpu_resize.py).  Compiled with -ffp-contract=off: the float32 weight polynomials must round as written.
#include "radnet_internal.h"

namespace {

__device__ __forceinline__ void cubic_coeffs(float x, short* c) {
  const float A = -0.75f;
  float w[4];
  w[0] = ((A * (x + 1.f) - 5.f * A) * (x + 1.f) + 8.f * A) * (x + 1.f) - 4.f * A;
  w[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
  w[2] = ((A + 2.f) * (1.f - x) - (A + 3.f)) * (1.f - x) * (1.f - x) + 1.f;
  w[3] = 1.f - w[0] - w[1] - w[2];
  for (int k = 0; k < 4; ++k) {
    float v = rintf(w[k] * 2048.f);                  // saturate_cast<short>(w * INTER_RESIZE_COEF_SCALE)
    v = fminf(fmaxf(v, -32768.f), 32767.f);
    c[k] = (short)v;
  }
}

__global__ void __launch_bounds__(256) resize_bicubic_u8_kernel(const uint8_t* __restrict__ src, int sh, int sw, uint8_t* __restrict__ dst, int dh,
                                                                int dw, int ch) {
  const long long total = (long long)dh * dw;
  const double scale_x = 1.0 / ((double)dw / sw), scale_y = 1.0 / ((double)dh / sh);      // OpenCV: 1. / inv_scale
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int dx = (int)(idx % dw), dy = (int)(idx / dw);
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx -= sx;
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = (int)floorf(fy);
    fy -= sy;
    short ax[4], ay[4];
    cubic_coeffs(fx, ax);
    cubic_coeffs(fy, ay);
    for (int c = 0; c < ch; ++c) {
      int acc = 0;
      for (int j = 0; j < 4; ++j) {
        const int yy = min(max(sy - 1 + j, 0), sh - 1);
        int row = 0;
        for (int i = 0; i < 4; ++i) {
          const int xx = min(max(sx - 1 + i, 0), sw - 1);
          row += (int)src[((long long)yy * sw + xx) * ch + c] * ax[i];
        }
        acc += row * ay[j];
      }
      const int v = (acc + (1 << 21)) >> 22;
      dst[idx * ch + c] = (uint8_t)min(max(v, 0), 255);
    }
  }
}

}  // namespace

extern "C" int radnet_resize_bicubic_u8(radnet_ctx* ctx, const uint8_t* src, int32_t sh, int32_t sw, uint8_t* dst, int32_t dh, int32_t dw,
                                        int32_t channels) {
  if (!ctx || !src || !dst || sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || channels <= 0) return RADNET_ERR_ARG;
  long long total = (long long)dh * dw;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(resize_bicubic_u8_kernel, dim3(blocks), dim3(256), 0, ctx->stream, src, sh, sw, dst, dh, dw, channels);
  RADNET_CHECK_LAUNCH(ctx, "resize_bicubic_u8");
  return RADNET_OK;
}

// ---- affine warp of uint8 HWC tiles: the arbitrary-angle rotation and the shear of the train-time augmentation -------------
// (augmentation.py:158-271: cv2.warpAffine with its defaults -- bilinear, constant border 0).  PARITY UNPINNED against OpenCV
// like the resize above; bit-for-bit equal to the NumPy restatement faster_rcnn/augmentation.py:warp_affine_u8, which documents
// the arithmetic: the host inverts the matrix and rounds the per-column / per-row terms of the inverse map to 10-bit fixed
// point in float64 (x_tab, y_tab: what OpenCV precomputes too), the kernel adds them, splits 1/32-pixel fractions off, and mixes
// the four neighbours (0 outside the image) with the integer weights 32 (32 - fx)(32 - fy) ..., rounding (sum + 2^14) >> 15.
namespace {
__global__ void __launch_bounds__(256) warp_affine_u8_kernel(const uint8_t* __restrict__ src, int sh, int sw, int ch, uint8_t* __restrict__ dst, int dh,
                                                             int dw, const int* __restrict__ col_tab, const int* __restrict__ row_tab) {
  const unsigned total = (unsigned)dh * (unsigned)dw;
  for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const unsigned y = idx / (unsigned)dw, x = idx - y * (unsigned)dw;
    // col_tab = {adelta[dw], bdelta[dw]}, row_tab = {x0[dh], y0[dh]} (x0 / y0 carry the half-step rounding offset)
    const int X = (row_tab[y] + col_tab[x]) >> 5, Y = (row_tab[dh + y] + col_tab[dw + x]) >> 5;
    int sx = X >> 5, sy = Y >> 5;
    sx = min(max(sx, -32768), 32767);
    sy = min(max(sy, -32768), 32767);
    const int fx = X & 31, fy = Y & 31;
    const int w00 = 32 * (32 - fx) * (32 - fy), w01 = 32 * fx * (32 - fy), w10 = 32 * (32 - fx) * fy, w11 = 32 * fx * fy;
    const bool y0in = (unsigned)sy < (unsigned)sh, y1in = (unsigned)(sy + 1) < (unsigned)sh;
    const bool x0in = (unsigned)sx < (unsigned)sw, x1in = (unsigned)(sx + 1) < (unsigned)sw;
    for (int c = 0; c < ch; ++c) {
      const int p00 = (y0in && x0in) ? src[((long long)sy * sw + sx) * ch + c] : 0;
      const int p01 = (y0in && x1in) ? src[((long long)sy * sw + sx + 1) * ch + c] : 0;
      const int p10 = (y1in && x0in) ? src[((long long)(sy + 1) * sw + sx) * ch + c] : 0;
      const int p11 = (y1in && x1in) ? src[((long long)(sy + 1) * sw + sx + 1) * ch + c] : 0;
      dst[(long long)idx * ch + c] = (uint8_t)((p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + (1 << 14)) >> 15);
    }
  }
}
}  // namespace

extern "C" int radnet_warp_affine_u8(radnet_ctx* ctx, const uint8_t* src, int32_t sh, int32_t sw, int32_t channels, uint8_t* dst, int32_t dh,
                                     int32_t dw, const int32_t* col_tab, const int32_t* row_tab) {
  if (!ctx || !src || !dst || !col_tab || !row_tab || sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || channels <= 0) return RADNET_ERR_ARG;
  const long long total = (long long)dh * dw;
  if (total >= (1ll << 31)) RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "warp_affine: %lld output pixels", total);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(warp_affine_u8_kernel, dim3(blocks), dim3(256), 0, ctx->stream, src, sh, sw, channels, dst, dh, dw, col_tab, row_tab);
  RADNET_CHECK_LAUNCH(ctx, "warp_affine_u8");
  return RADNET_OK;
}
