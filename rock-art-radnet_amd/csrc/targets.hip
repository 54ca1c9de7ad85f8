// Training-target generation in fp64 with the reference's exact semantics:
//   anchor targets  = utils.calc_region_props before its random subsampling (utils.py:585-766)
//   RoI labelling   = rpn.calc_iou (rpn.py:176-296)
// Compiled with -ffp-contract=off (Python/NumPy round every operation separately).
//
// The reference walks five nested Python loops (size, ratio, ix, jy, gt) keeping running "best"
// values with strict '>' comparisons.  Here every anchor is one thread; the order-dependent parts
// are reproduced with order-encoding atomics:
//   * best anchor per GT = first anchor in visiting order with the largest fp32 IoU: atomicMax on
//     (fp32 IoU bits << 32 | ~visit_index);
//   * the per-GT fallback (utils.py:741-766) is applied afterwards by one thread in GT order, so a
//     later GT overwrites an earlier one exactly as the reference's sequential loop does.
#include "radnet_internal.h"

namespace {

constexpr int kMaxGt = 1024;

struct AnchorArgs {
  const double* gt;          // [g][4] x1,y1,x2,y2 source pixels
  const int* gt_is_bg;
  int g, width, height, rw, rh, fw, fh, ns, nr;
  double stride, max_overlap;
  double sizes[16];
  double ratios[16][2];
};

__device__ __forceinline__ double iou_ref(double ax1, double ay1, double ax2, double ay2, double bx1, double by1, double bx2, double by2) {
  // utils.py:77-109 with a = GT box, b = anchor / RoI
  if (ax1 >= ax2 || ay1 >= ay2 || bx1 >= bx2 || by1 >= by2) return 0.0;
  const double x = fmax(ax1, bx1), y = fmax(ay1, by1);
  const double w = fmin(ax2, bx2) - x, h = fmin(ay2, by2) - y;
  const double inter = (w < 0 || h < 0) ? 0.0 : w * h;
  const double uni = (ax2 - ax1) * (ay2 - ay1) + (bx2 - bx1) * (by2 - by1) - inter;
  return inter / (uni + 1e-6);
}

struct AnchorBox {
  double x1, y1, x2, y2;
  bool inside;
};

__device__ __forceinline__ AnchorBox anchor_box(const AnchorArgs& g, int si, int ri, int ix, int jy) {
  const double ax = g.sizes[si] * g.ratios[ri][0], ay = g.sizes[si] * g.ratios[ri][1];
  AnchorBox b;
  b.x1 = g.stride * (ix + 0.5) - ax / 2;
  b.x2 = g.stride * (ix + 0.5) + ax / 2;
  b.y1 = g.stride * (jy + 0.5) - ay / 2;
  b.y2 = g.stride * (jy + 0.5) + ay / 2;
  b.inside = !(b.x1 < 0 || b.x2 > g.rw) && !(b.y1 < 0 || b.y2 > g.rh);   // utils.py:629,638
  return b;
}

__device__ __forceinline__ void gt_resized(const AnchorArgs& g, int k, double& x1, double& x2, double& y1, double& y2) {
  // utils.py:610-613
  x1 = g.gt[4 * k + 0] * ((double)g.rw / (double)g.width);
  x2 = g.gt[4 * k + 2] * ((double)g.rw / (double)g.width);
  y1 = g.gt[4 * k + 1] * ((double)g.rh / (double)g.height);
  y2 = g.gt[4 * k + 3] * ((double)g.rh / (double)g.height);
}

__device__ __forceinline__ void deltas(double gx1, double gx2, double gy1, double gy2, const AnchorBox& b, double* t) {
  // utils.py:669-687
  const double cx = (gx1 + gx2) / 2.0, cy = (gy1 + gy2) / 2.0;
  const double cxa = (b.x1 + b.x2) / 2.0, cya = (b.y1 + b.y2) / 2.0;
  t[0] = (cx - cxa) / (b.x2 - b.x1);
  t[1] = (cy - cya) / (b.y2 - b.y1);
  t[2] = log((gx2 - gx1) / (b.x2 - b.x1));
  t[3] = log((gy2 - gy1) / (b.y2 - b.y1));
}

__global__ void __launch_bounds__(256) anchor_targets_kernel(AnchorArgs g, uint8_t* __restrict__ valid, uint8_t* __restrict__ overlap,
                                                             double* __restrict__ regr, unsigned long long* __restrict__ best,
                                                             int* __restrict__ n_for_gt) {
  __shared__ double sgt[kMaxGt][4];      // resized x1,x2,y1,y2
  __shared__ int sbg[kMaxGt];
  for (int k = threadIdx.x; k < g.g; k += blockDim.x) {
    gt_resized(g, k, sgt[k][0], sgt[k][1], sgt[k][2], sgt[k][3]);
    sbg[k] = g.gt_is_bg[k];
  }
  __syncthreads();
  const int A = g.ns * g.nr;
  const int total = A * g.fw * g.fh;
  const int o = blockIdx.x * blockDim.x + threadIdx.x;     // visiting order: ((si*nr + ri)*fw + ix)*fh + jy
  if (o >= total) return;
  const int jy = o % g.fh;
  const int ix = (o / g.fh) % g.fw;
  const int ar = o / (g.fh * g.fw);
  const int ri = ar % g.nr, si = ar / g.nr;
  const AnchorBox b = anchor_box(g, si, ri, ix, jy);
  // every anchor owns one (valid, overlap, regr[4]) cell and writes it, zeros included: no memset of the maps
  const int ch = ri + g.nr * si;
  const size_t chw = ((size_t)ch * g.fh + jy) * g.fw + ix;
  double* r = regr + ((size_t)jy * g.fw + ix) * 4 * A + 4 * ch;
  if (!b.inside || g.g == 0) {                             // label write sits inside the GT loop (utils.py:723-738)
    valid[chw] = 0;
    overlap[chw] = 0;
    r[0] = r[1] = r[2] = r[3] = 0.0;
    return;
  }
  bool pos = false;
  double best_loc = 0.0;
  int best_k = -1;
  for (int k = 0; k < g.g; ++k) {
    const double v = iou_ref(sgt[k][0], sgt[k][2], sgt[k][1], sgt[k][3], b.x1, b.y1, b.x2, b.y2);
    if (sbg[k]) continue;
    const float v32 = (float)v;                            // numpy-2: fp32 bookkeeping compared in fp32
    if (v32 > 0.0f) {
      const unsigned long long key = ((unsigned long long)__float_as_uint(v32) << 32) | (0xFFFFFFFFu - (unsigned int)o);
      atomicMax(best + k, key);
    }
    if (v > g.max_overlap) {
      pos = true;
      atomicAdd(n_for_gt + k, 1);
      if (v > best_loc) { best_loc = v; best_k = k; }
    }
  }
  valid[chw] = 1;
  overlap[chw] = pos ? 1 : 0;
  double t[4] = {0.0, 0.0, 0.0, 0.0};
  if (pos) deltas(sgt[best_k][0], sgt[best_k][1], sgt[best_k][2], sgt[best_k][3], b, t);
  r[0] = t[0]; r[1] = t[1]; r[2] = t[2]; r[3] = t[3];
}

// per-GT scratch of the launch above (best-anchor keys, positive counts): one small launch instead of two memsets
__global__ void anchor_clear_kernel(unsigned long long* __restrict__ best, int* __restrict__ n_for_gt, int g) {
  for (int k = threadIdx.x; k < g; k += blockDim.x) {
    best[k] = 0ull;
    n_for_gt[k] = 0;
  }
}

__global__ void anchor_fallback_kernel(AnchorArgs g, uint8_t* __restrict__ valid, uint8_t* __restrict__ overlap, double* __restrict__ regr,
                                       const unsigned long long* __restrict__ best, const int* __restrict__ n_for_gt,
                                       int* __restrict__ best_anchor) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const int A = g.ns * g.nr;
  for (int k = 0; k < g.g; ++k) {
    const unsigned long long key = best[k];
    if (key == 0ull) {
      best_anchor[4 * k + 0] = best_anchor[4 * k + 1] = best_anchor[4 * k + 2] = best_anchor[4 * k + 3] = -1;
      continue;
    }
    const int o = (int)(0xFFFFFFFFu - (unsigned int)(key & 0xFFFFFFFFull));
    const int jy = o % g.fh, ix = (o / g.fh) % g.fw, ar = o / (g.fh * g.fw);
    const int ri = ar % g.nr, si = ar / g.nr;
    best_anchor[4 * k + 0] = jy; best_anchor[4 * k + 1] = ix; best_anchor[4 * k + 2] = ri; best_anchor[4 * k + 3] = si;
    if (n_for_gt[k] == 0) {
      const AnchorBox b = anchor_box(g, si, ri, ix, jy);
      double gx1, gx2, gy1, gy2, t[4];
      gt_resized(g, k, gx1, gx2, gy1, gy2);
      deltas(gx1, gx2, gy1, gy2, b, t);
      const int ch = ri + g.nr * si;
      const size_t chw = ((size_t)ch * g.fh + jy) * g.fw + ix;
      valid[chw] = 1;
      overlap[chw] = 1;
      double* r = regr + ((size_t)jy * g.fw + ix) * 4 * A + 4 * ch;
      for (int q = 0; q < 4; ++q) r[q] = (double)(float)t[q];      // best_dx_for_bbox is fp32 (utils.py:605,700)
    }
  }
}

__global__ void __launch_bounds__(256) anchor_pack_kernel(const uint8_t* __restrict__ valid, const uint8_t* __restrict__ overlap,
                                                          const double* __restrict__ regr, int fw, int fh, int a, double std_scaling,
                                                          float* __restrict__ ycls, float* __restrict__ yregr) {
  const int hw = fw * fh;
  const int total = hw * 8 * a;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int pix = idx / (8 * a), col = idx - pix * 8 * a;
    float v;
    if (col < 4 * a) v = (float)overlap[(size_t)(col >> 2) * hw + pix];              // np.repeat(overlap, 4)
    else v = (float)(regr[(size_t)pix * 4 * a + (col - 4 * a)] * std_scaling);         // utils.py:475
    yregr[idx] = v;
    if (col < 2 * a) {
      const uint8_t* src = col < a ? valid : overlap;
      const int ch = col < a ? col : col - a;
      ycls[(size_t)pix * 2 * a + col] = (float)src[(size_t)ch * hw + pix];
    }
  }
}

// ---- RoI labelling ---------------------------------------------------------------------------------------
struct RoiArgs {
  const long long* rois;
  const double* gt;
  const int* gt_cls;
  const int* n_dev;
  int n, g, width, height, rw, rh, bg;
  double stride, min_ov, max_ov;
  double std[4];
};

__global__ void __launch_bounds__(256) roi_targets_kernel(RoiArgs g, uint8_t* __restrict__ keep, int* __restrict__ cls, int* __restrict__ box,
                                                          double* __restrict__ t, double* __restrict__ iou_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= g.n) return;
  if (g.n_dev != nullptr && i >= *g.n_dev) {      // rows beyond the device-side proposal count: not kept
    keep[i] = 0;
    cls[i] = -1;
    return;
  }
  const double x1 = (double)g.rois[4 * i], y1 = (double)g.rois[4 * i + 1], x2 = (double)g.rois[4 * i + 2], y2 = (double)g.rois[4 * i + 3];
  double best = 0.0;
  int bk = -1;
  double bx1 = 0, bx2 = 0, by1 = 0, by2 = 0;
  for (int k = 0; k < g.g; ++k) {
    // rpn.py:197-200: int(round(coord * (resized/orig) / stride)), Python round = half-to-even
    const double gx1 = rint(g.gt[4 * k + 0] * ((double)g.rw / (double)g.width) / g.stride);
    const double gx2 = rint(g.gt[4 * k + 2] * ((double)g.rw / (double)g.width) / g.stride);
    const double gy1 = rint(g.gt[4 * k + 1] * ((double)g.rh / (double)g.height) / g.stride);
    const double gy2 = rint(g.gt[4 * k + 3] * ((double)g.rh / (double)g.height) / g.stride);
    const double v = iou_ref(gx1, gy1, gx2, gy2, x1, y1, x2, y2);
    if (v > best) { best = v; bk = k; bx1 = gx1; bx2 = gx2; by1 = gy1; by2 = gy2; }
  }
  double tt[4] = {0, 0, 0, 0};
  int c = g.bg;
  const bool kp = !(best < g.min_ov);
  const double w = x2 - x1, h = y2 - y1;
  if (kp && !(best < g.max_ov)) {
    c = g.gt_cls[bk];
    if (c != g.bg) {
      const double cxg = (bx1 + bx2) / 2.0, cyg = (by1 + by2) / 2.0;
      const double cx = x1 + w / 2.0, cy = y1 + h / 2.0;
      tt[0] = g.std[0] * ((cxg - cx) / w);
      tt[1] = g.std[1] * ((cyg - cy) / h);
      tt[2] = g.std[2] * log((bx2 - bx1) / w);
      tt[3] = g.std[3] * log((by2 - by1) / h);
    }
  }
  keep[i] = kp ? 1 : 0;
  cls[i] = kp ? c : -1;                           // -1 = dropped (IoU < min_overlap): one array tells the host everything
  box[4 * i + 0] = (int)x1; box[4 * i + 1] = (int)y1; box[4 * i + 2] = (int)w; box[4 * i + 3] = (int)h;
  for (int q = 0; q < 4; ++q) t[4 * i + q] = tt[q];
  iou_out[i] = best;
}

__global__ void roi_batch_pack_kernel(const int* __restrict__ sel, int r, const int* __restrict__ cls, const int* __restrict__ box,
                                      const double* __restrict__ t, int nc, int bg, float* __restrict__ rois_out, float* __restrict__ y1,
                                      float* __restrict__ y2) {
  const int j = blockIdx.x;
  if (j >= r) return;
  const int src = sel[j];
  const int c = cls[src];
  const int nreg = 4 * (nc - 1);
  for (int q = threadIdx.x; q < 4; q += blockDim.x) rois_out[4 * j + q] = (float)box[4 * src + q];
  for (int q = threadIdx.x; q < nc; q += blockDim.x) y1[(size_t)j * nc + q] = q == c ? 1.f : 0.f;
  for (int q = threadIdx.x; q < nreg; q += blockDim.x) {
    const bool on = c != bg && (q >> 2) == c;
    y2[(size_t)j * 2 * nreg + q] = on ? 1.f : 0.f;
    y2[(size_t)j * 2 * nreg + nreg + q] = on ? (float)t[4 * src + (q & 3)] : 0.f;
  }
}

}  // namespace

extern "C" int radnet_anchor_targets(radnet_ctx* ctx, const double* gt, const int32_t* gt_is_bg, int32_t g, int32_t width, int32_t height,
                                     int32_t rw, int32_t rh, int32_t fw, int32_t fh, const double* anchor_sizes_host, int32_t ns,
                                     const double* anchor_ratios_host, int32_t nr, double rpn_stride, double max_overlap, uint8_t* valid,
                                     uint8_t* overlap, double* regr, int32_t* best_anchor, int32_t* n_for_gt, void* scratch) {
  if (!ctx || !valid || !overlap || !regr || !anchor_sizes_host || !anchor_ratios_host) return RADNET_ERR_ARG;
  if (ns < 1 || ns > 16 || nr < 1 || nr > 16) RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "anchor_targets: ns=%d nr=%d (max 16 each)", ns, nr);
  if (g < 0 || g > kMaxGt) RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "anchor_targets: %d GT boxes (max %d)", g, kMaxGt);
  if (g > 0 && (!gt || !gt_is_bg || !best_anchor || !n_for_gt || !scratch)) return RADNET_ERR_ARG;
  AnchorArgs a{};
  a.gt = gt; a.gt_is_bg = gt_is_bg; a.g = g; a.width = width; a.height = height; a.rw = rw; a.rh = rh; a.fw = fw; a.fh = fh;
  a.ns = ns; a.nr = nr; a.stride = rpn_stride; a.max_overlap = max_overlap;
  for (int i = 0; i < ns; ++i) a.sizes[i] = anchor_sizes_host[i];
  for (int i = 0; i < nr; ++i) { a.ratios[i][0] = anchor_ratios_host[2 * i]; a.ratios[i][1] = anchor_ratios_host[2 * i + 1]; }
  const int A = ns * nr;
  const size_t n = (size_t)A * fw * fh;
  if (g == 0) {              // no ground truth: nothing is valid (utils.py:723-738 never runs)
    RADNET_CHECK_HIP(ctx, hipMemsetAsync(valid, 0, n, ctx->stream));
    RADNET_CHECK_HIP(ctx, hipMemsetAsync(overlap, 0, n, ctx->stream));
    RADNET_CHECK_HIP(ctx, hipMemsetAsync(regr, 0, n * 4 * sizeof(double), ctx->stream));
    return RADNET_OK;
  }
  hipLaunchKernelGGL(anchor_clear_kernel, dim3(1), dim3(256), 0, ctx->stream, (unsigned long long*)scratch, n_for_gt, g);
  RADNET_CHECK_LAUNCH(ctx, "anchor_clear");
  hipLaunchKernelGGL(anchor_targets_kernel, dim3(radnet_cdiv(n, 256)), dim3(256), 0, ctx->stream, a, valid, overlap, regr,
                     (unsigned long long*)scratch, n_for_gt);
  RADNET_CHECK_LAUNCH(ctx, "anchor_targets");
  hipLaunchKernelGGL(anchor_fallback_kernel, dim3(1), dim3(64), 0, ctx->stream, a, valid, overlap, regr, (const unsigned long long*)scratch,
                     (const int*)n_for_gt, best_anchor);
  RADNET_CHECK_LAUNCH(ctx, "anchor_fallback");
  return RADNET_OK;
}

extern "C" int radnet_anchor_targets_pack(radnet_ctx* ctx, const uint8_t* valid, const uint8_t* overlap, const double* regr, int32_t fw,
                                          int32_t fh, int32_t a, double std_scaling, float* y_cls, float* y_regr) {
  if (!ctx || !valid || !overlap || !regr || !y_cls || !y_regr) return RADNET_ERR_ARG;
  const int total = fw * fh * 8 * a;
  hipLaunchKernelGGL(anchor_pack_kernel, dim3(radnet_cdiv(total, 256)), dim3(256), 0, ctx->stream, valid, overlap, regr, fw, fh, a,
                     std_scaling, y_cls, y_regr);
  RADNET_CHECK_LAUNCH(ctx, "anchor_pack");
  return RADNET_OK;
}

extern "C" int radnet_roi_targets(radnet_ctx* ctx, const int64_t* rois, int32_t n, const double* gt, const int32_t* gt_cls, int32_t g,
                                  int32_t width, int32_t height, int32_t rw, int32_t rh, double rpn_stride, double min_overlap,
                                  double max_overlap, const double* regr_std_host4, int32_t bg_class, uint8_t* keep, int32_t* cls,
                                  int32_t* box, double* t, double* iou, const int32_t* n_dev) {
  if (!ctx || !rois || !keep || !cls || !box || !t || !iou || !regr_std_host4) return RADNET_ERR_ARG;
  if (n <= 0) return RADNET_OK;
  if (g > 0 && (!gt || !gt_cls)) return RADNET_ERR_ARG;
  RoiArgs a{};
  a.n_dev = n_dev;
  a.rois = (const long long*)rois; a.gt = gt; a.gt_cls = gt_cls; a.n = n; a.g = g; a.width = width; a.height = height; a.rw = rw; a.rh = rh;
  a.bg = bg_class; a.stride = rpn_stride; a.min_ov = min_overlap; a.max_ov = max_overlap;
  for (int i = 0; i < 4; ++i) a.std[i] = regr_std_host4[i];
  hipLaunchKernelGGL(roi_targets_kernel, dim3(radnet_cdiv(n, 256)), dim3(256), 0, ctx->stream, a, keep, cls, box, t, iou);
  RADNET_CHECK_LAUNCH(ctx, "roi_targets");
  return RADNET_OK;
}

extern "C" int radnet_roi_batch_pack(radnet_ctx* ctx, const int32_t* sel, int32_t r, const int32_t* cls, const int32_t* box, const double* t,
                                     int32_t nc, int32_t bg_class, float* rois_out, float* y1, float* y2) {
  if (!ctx || !sel || !cls || !box || !t || !rois_out || !y1 || !y2) return RADNET_ERR_ARG;
  hipLaunchKernelGGL(roi_batch_pack_kernel, dim3(r), dim3(64), 0, ctx->stream, sel, r, cls, box, t, nc, bg_class, rois_out, y1, y2);
  RADNET_CHECK_LAUNCH(ctx, "roi_batch_pack");
  return RADNET_OK;
}
