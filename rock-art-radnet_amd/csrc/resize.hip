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
