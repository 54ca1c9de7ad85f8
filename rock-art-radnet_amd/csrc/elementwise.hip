// HBM-bound and small kernels of the Faster R-CNN hot path: pooling, RoI crop-resize, classifier
// dense heads, losses (forward value + gradient in one pass), bias-gradient column sums, Adam.
// Each stands in for a TensorFlow op the reference graph invokes implicitly; citations inline.
#include "radnet_internal.h"
#include "radnet_wino4.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- MaxPooling2D k x k / stride s, 'valid' (resnet50.py:188; VGG16 block pools) ------------------
__global__ void __launch_bounds__(256) maxpool_kernel(const float* __restrict__ x, float* __restrict__ y, int nb, int h, int w,
                                                      int c4, int oh, int ow, int k, int s) {
  const long long total = (long long)nb * oh * ow * c4;
  // 32-bit index arithmetic (the launcher rejects >= 2^31 work items): a 64-bit division is a long software routine
  for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < (unsigned)total; idx += gridDim.x * blockDim.x) {
    unsigned p = idx / (unsigned)c4;
    int cc = (int)(idx - p * (unsigned)c4);
    unsigned q = p / (unsigned)ow;
    int ox = (int)(p - q * (unsigned)ow);
    int b = (int)(q / (unsigned)oh);
    int oy = (int)(q - (unsigned)b * (unsigned)oh);
    const float4* src = reinterpret_cast<const float4*>(x) + ((long long)(b * h + oy * s) * w + ox * s) * c4 + cc;
    float4 m = src[0];
    for (int i = 0; i < k; ++i)
      for (int j = 0; j < k; ++j) {
        float4 v = src[((long long)i * w + j) * c4];
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    reinterpret_cast<float4*>(y)[idx] = m;
  }
}

// ---- RoiPoolingConv.call (RoiPoolingConv.py:48-88): crop + TF1 legacy bilinear resize ---------------
struct RoiGeom {
  int x0, y0, cw, ch;
};
__device__ __forceinline__ RoiGeom roi_geom(const float* roi, int H, int W) {
  int x = (int)roi[0], y = (int)roi[1], w = (int)roi[2], h = (int)roi[3];   // K.cast(..., 'int32'): truncation
  int y0 = min(max(y, 0), H), y1 = min(max(y + h, 0), H);                   // slice clamping
  int x0 = min(max(x, 0), W), x1 = min(max(x + w, 0), W);
  return RoiGeom{x0, y0, x1 - x0, y1 - y0};
}

// one workgroup per output pixel (roi, oy, ox); threads stride over channel quads: every global access
// is a run of consecutive 16-byte words (1 KiB per wave instruction), the 4 taps come from L2/L1.
__global__ void __launch_bounds__(256) roi_resize_fwd_kernel(const float* __restrict__ fmap, int H, int W, int c4,
                                                             const float* __restrict__ rois, int ps, float* __restrict__ y) {
  const int o = blockIdx.x;
  const int ox = o % ps, oy = (o / ps) % ps, r = o / (ps * ps);
  const RoiGeom g = roi_geom(rois + 4 * r, H, W);
  float4* dst = reinterpret_cast<float4*>(y) + (long long)o * c4;
  if (g.cw <= 0 || g.ch <= 0) {
    for (int c = threadIdx.x; c < c4; c += blockDim.x) dst[c] = make_float4(0, 0, 0, 0);
    return;
  }
  const float hs = (float)g.ch / (float)ps, ws = (float)g.cw / (float)ps;
  const float sy = (float)oy * hs, sx = (float)ox * ws;
  const int ylo = (int)floorf(sy), xlo = (int)floorf(sx);
  const int yhi = min(ylo + 1, g.ch - 1), xhi = min(xlo + 1, g.cw - 1);
  const float ly = sy - (float)ylo, lx = sx - (float)xlo;
  const float4* f = reinterpret_cast<const float4*>(fmap);
  const float4* tl = f + ((long long)(g.y0 + ylo) * W + g.x0 + xlo) * c4;
  const float4* tr = f + ((long long)(g.y0 + ylo) * W + g.x0 + xhi) * c4;
  const float4* bl = f + ((long long)(g.y0 + yhi) * W + g.x0 + xlo) * c4;
  const float4* br = f + ((long long)(g.y0 + yhi) * W + g.x0 + xhi) * c4;
  for (int c = threadIdx.x; c < c4; c += blockDim.x) {
    float4 a = tl[c], b = tr[c], d = bl[c], e = br[c], out;
#define LERP2(q)                                         \
  {                                                      \
    float top = a.q + (b.q - a.q) * lx;                  \
    float bot = d.q + (e.q - d.q) * lx;                  \
    out.q = top + (bot - top) * ly;                      \
  }
    LERP2(x) LERP2(y) LERP2(z) LERP2(w)
#undef LERP2
    dst[c] = out;
  }
}

__global__ void __launch_bounds__(256) roi_resize_bwd_kernel(const float* __restrict__ dy, int H, int W, int C,
                                                             const float* __restrict__ rois, int ps, float* __restrict__ dfmap) {
  const int o = blockIdx.x;
  const int ox = o % ps, oy = (o / ps) % ps, r = o / (ps * ps);
  const RoiGeom g = roi_geom(rois + 4 * r, H, W);
  if (g.cw <= 0 || g.ch <= 0) return;
  const float hs = (float)g.ch / (float)ps, ws = (float)g.cw / (float)ps;
  const float sy = (float)oy * hs, sx = (float)ox * ws;
  const int ylo = (int)floorf(sy), xlo = (int)floorf(sx);
  const int yhi = min(ylo + 1, g.ch - 1), xhi = min(xlo + 1, g.cw - 1);
  const float ly = sy - (float)ylo, lx = sx - (float)xlo;
  float* tl = dfmap + ((long long)(g.y0 + ylo) * W + g.x0 + xlo) * C;
  float* tr = dfmap + ((long long)(g.y0 + ylo) * W + g.x0 + xhi) * C;
  float* bl = dfmap + ((long long)(g.y0 + yhi) * W + g.x0 + xlo) * C;
  float* br = dfmap + ((long long)(g.y0 + yhi) * W + g.x0 + xhi) * C;
  const float* src = dy + (long long)o * C;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float v = src[c];
    atomicAdd(tl + c, v * (1.f - ly) * (1.f - lx));
    atomicAdd(tr + c, v * (1.f - ly) * lx);
    atomicAdd(bl + c, v * ly * (1.f - lx));
    atomicAdd(br + c, v * ly * lx);
  }
}

// Ordered form of the same gradient (radnet_ctx::deterministic): overlapping RoIs add into the same feature-map pixel, and
// the atomics above add them in whatever order the workgroups run.  Here the sum is GATHERED: a workgroup owns one feature-map pixel
// (all channels, four per thread) and walks the RoIs, their output rows and output columns in index order; a tap that lands on its pixel
// is added to a register.  Every decision (does RoI r cover the pixel, is output row oy / column ox one of its two neighbours) is the
// forward kernel's own float arithmetic on wave-uniform values, so the loops are scalar branches; only taps that hit cost a load.  The
// order of the additions is a function of the RoI list alone; products are rounded before they are added (no contraction into an FMA:
// the result does not depend on how the compiler schedules the chain).
// (Round 4.  The form before this one gave a workgroup a feature-map ROW in LDS and walked the RoIs with one wave: ~40 dependent memory
// round trips per row, then 63 dependent read-modify-writes of the row -- 100-165 us for 20 RoIs on the 38x63 map against 50 us for the
// atomics; return-less LDS adds and four waves per row changed nothing, tools/roi_bwd_timing.py.)
__global__ void __launch_bounds__(256) roi_resize_bwd_ordered_kernel(const float* __restrict__ dy, int H, int W, int C4,
                                                                     const float* __restrict__ rois, int R, int ps, float* __restrict__ dfmap) {
  const int y = blockIdx.x / W, x = blockIdx.x - y * W;
  const float4* src = reinterpret_cast<const float4*>(dy);
  for (int c4 = threadIdx.x; c4 < C4; c4 += 256) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    bool touched = false;
    auto add = [&](const float4& v, float wy, float wx) {
      acc.x = __fadd_rn(acc.x, __fmul_rn(__fmul_rn(v.x, wy), wx));
      acc.y = __fadd_rn(acc.y, __fmul_rn(__fmul_rn(v.y, wy), wx));
      acc.z = __fadd_rn(acc.z, __fmul_rn(__fmul_rn(v.z, wy), wx));
      acc.w = __fadd_rn(acc.w, __fmul_rn(__fmul_rn(v.w, wy), wx));
    };
    for (int r = 0; r < R; ++r) {
      const RoiGeom g = roi_geom(rois + 4 * r, H, W);
      if (g.cw <= 0 || g.ch <= 0 || y < g.y0 || y >= g.y0 + g.ch || x < g.x0 || x >= g.x0 + g.cw) continue;
      const float hs = (float)g.ch / (float)ps, ws = (float)g.cw / (float)ps;
      // output columns whose low / high neighbour is this pixel's column (bit ox)
      unsigned lo_mask = 0u, hi_mask = 0u;
      for (int ox = 0; ox < ps; ++ox) {
        const float sx = (float)ox * ws;
        const int xlo = (int)floorf(sx), xhi = min(xlo + 1, g.cw - 1);
        lo_mask |= (g.x0 + xlo == x ? 1u : 0u) << ox;
        hi_mask |= (g.x0 + xhi == x ? 1u : 0u) << ox;
      }
      const unsigned any_mask = lo_mask | hi_mask;
      if (any_mask == 0u) continue;
      for (int oy = 0; oy < ps; ++oy) {
        const float sy = (float)oy * hs;
        const int ylo = (int)floorf(sy), yhi = min(ylo + 1, g.ch - 1);
        const float ly = sy - (float)ylo;
        const bool top = g.y0 + ylo == y, bot = g.y0 + yhi == y;
        if (!top && !bot) continue;
        const float4* rowp = src + ((long long)(r * ps + oy) * ps) * C4 + c4;
        for (unsigned m = any_mask; m != 0u; m &= m - 1u) {
          const int ox = __builtin_ctz(m);
          const float sx = (float)ox * ws;
          const float lx = sx - (float)(int)floorf(sx);
          const bool lo = (lo_mask >> ox) & 1u, hi = (hi_mask >> ox) & 1u;
          const float4 v = rowp[(long long)ox * C4];
          touched = true;
          // the order the scatter forms use for one pixel: top-low, top-high, bottom-low, bottom-high
          if (top && lo) add(v, 1.f - ly, 1.f - lx);
          if (top && hi) add(v, 1.f - ly, lx);
          if (bot && lo) add(v, ly, 1.f - lx);
          if (bot && hi) add(v, ly, lx);
        }
      }
    }
    if (touched) {
      float4* dst = reinterpret_cast<float4*>(dfmap) + (long long)blockIdx.x * C4 + c4;
      float4 d = *dst;
      d.x += acc.x; d.y += acc.y; d.z += acc.z; d.w += acc.w;
      *dst = d;
    }
  }
}

// ---- AveragePooling2D((7,7)) + Flatten over the RoI axis (resnet50.py:260-261) ----------------------
__global__ void __launch_bounds__(256) avgpool_fwd_kernel(const float* __restrict__ x, int r, int hw, int c4, float* __restrict__ y) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= r * c4) return;
  const int rr = idx / c4, cc = idx - rr * c4;
  const float4* src = reinterpret_cast<const float4*>(x) + (long long)rr * hw * c4 + cc;
  float4 s = make_float4(0, 0, 0, 0);
  for (int p = 0; p < hw; ++p) {
    float4 v = src[(long long)p * c4];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  const float inv = (float)hw;
  reinterpret_cast<float4*>(y)[idx] = make_float4(s.x / inv, s.y / inv, s.z / inv, s.w / inv);
}

__global__ void __launch_bounds__(256) avgpool_bwd_relu_kernel(const float* __restrict__ dfeat, const float* __restrict__ yact,
                                                               int r, int hw, int c4, float* __restrict__ dx) {
  const long long total = (long long)r * hw * c4;
  const float inv = (float)hw;
  for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < (unsigned)total; idx += gridDim.x * blockDim.x) {      // total < 2^31 (launcher)
    int cc = (int)(idx % (unsigned)c4);
    int rr = (int)(idx / ((unsigned)hw * (unsigned)c4));
    float4 g = reinterpret_cast<const float4*>(dfeat)[(long long)rr * c4 + cc];
    float4 a = reinterpret_cast<const float4*>(yact)[idx];
    float4 o;
    o.x = a.x > 0.f ? g.x / inv : 0.f;
    o.y = a.y > 0.f ? g.y / inv : 0.f;
    o.z = a.z > 0.f ? g.z / inv : 0.f;
    o.w = a.w > 0.f ? g.w / inv : 0.f;
    reinterpret_cast<float4*>(dx)[idx] = o;
  }
}

// ---- classifier dense heads (resnet50.py:263-279): one workgroup per RoI --------------------------------
template <int NP>
__global__ void __launch_bounds__(256) dense_heads_fwd_kernel(const float* __restrict__ feat, int k, const float* __restrict__ w,
                                                              const float* __restrict__ b, int nc, int nreg,
                                                              float* __restrict__ out_cls, float* __restrict__ out_regr) {
  __shared__ float red[4][NP];
  __shared__ float z[NP];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float acc[NP];
#pragma unroll
  for (int n = 0; n < NP; ++n) acc[n] = 0.f;
  const float* f = feat + (long long)r * k;
  for (int kk = tid; kk < k; kk += 256) {
    const float xv = f[kk];
    const float4* wr = reinterpret_cast<const float4*>(w + (long long)kk * NP);
#pragma unroll
    for (int q = 0; q < NP / 4; ++q) {
      float4 v = wr[q];
      acc[4 * q + 0] += xv * v.x; acc[4 * q + 1] += xv * v.y; acc[4 * q + 2] += xv * v.z; acc[4 * q + 3] += xv * v.w;
    }
  }
#pragma unroll
  for (int n = 0; n < NP; ++n) {
    float s = wave_sum(acc[n]);
    if (lane == 0) red[wave][n] = s;
  }
  __syncthreads();
  const int nout = nc + nreg;
  if (tid < nout) z[tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid] + b[tid];
  __syncthreads();
  if (tid < nout) {
    if (tid < nc) {
      float mx = z[0];
      for (int i = 1; i < nc; ++i) mx = fmaxf(mx, z[i]);
      float s = 0.f;
      for (int i = 0; i < nc; ++i) s += expf(z[i] - mx);
      out_cls[(long long)r * nc + tid] = expf(z[tid] - mx) / s;
    } else {
      out_regr[(long long)r * nreg + (tid - nc)] = z[tid];
    }
  }
}

// dw[k][n] = sum_r feat[r][k] dz[r][n];  dfeat[r][k] = sum_n dz[r][n] w[k][n];  db[n] = sum_r dz[r][n]
__global__ void __launch_bounds__(256) dense_heads_bwd_kernel(const float* __restrict__ feat, const float* __restrict__ dz, int r, int k,
                                                              const float* __restrict__ w, int np, int nout, float* __restrict__ dw,
                                                              float* __restrict__ db, float* __restrict__ dfeat, int acc) {
  extern __shared__ float sdz[];            // [r][np]
  for (int i = threadIdx.x; i < r * np; i += blockDim.x) {
    int rr = i / np, n = i - rr * np;
    sdz[i] = n < nout ? dz[(long long)rr * nout + n] : 0.f;
  }
  __syncthreads();
  const int kk = blockIdx.x * 8 + (threadIdx.x >> 5);      // 8 k rows per block, 32 threads per row
  const int t = threadIdx.x & 31;
  if (kk < k) {
    for (int n = t; n < np; n += 32) {
      float s = 0.f;
      for (int rr = 0; rr < r; ++rr) s += feat[(long long)rr * k + kk] * sdz[rr * np + n];
      const float v = n < nout ? s : 0.f;
      if (acc) dw[(long long)kk * np + n] += v;
      else dw[(long long)kk * np + n] = v;
    }
    for (int rr = t; rr < r; rr += 32) {
      float s = 0.f;
      const float* wr = w + (long long)kk * np;
      for (int n = 0; n < nout; ++n) s += sdz[rr * np + n] * wr[n];
      dfeat[(long long)rr * k + kk] = s;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < np) {
    float s = 0.f;
    for (int rr = 0; rr < r; ++rr) s += sdz[rr * np + threadIdx.x];
    const float v = threadIdx.x < nout ? s : 0.f;
    if (acc) db[threadIdx.x] += v;
    else db[threadIdx.x] = v;
  }
}

// ---- column sums (bias gradients) ----------------------------------------------------------------------
// `partials` != null (radnet_ctx::deterministic): the row blocks of a column block hand their partial sums to the last one
// to arrive, which adds them in a fixed shape (sc1 stores / relaxed agent-scope ticket / sc1 loads, as the split-K
// reduction of conv_mfma.hip) -- the same bits on every run.  partials == null: one fp32 atomic per block and column.
__global__ void __launch_bounds__(256) colsum_kernel(const float* __restrict__ g, int m, int n, int ld, const float* __restrict__ gscale,
                                                     float* __restrict__ out, int rows_per_block, float* __restrict__ partials,
                                                     unsigned* __restrict__ counters) {
  __shared__ float red[4][64];
  __shared__ int s_last;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + tx;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(m, r0 + rows_per_block);
  float s = 0.f;
  if (col < n)
    for (int rr = r0 + ty; rr < r1; rr += 4) s += g[(long long)rr * ld + col];
  red[ty][tx] = s;
  __syncthreads();
  float v = red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx];
  if (partials == nullptr) {
    if (ty == 0 && col < n) {
      if (gscale) v *= gscale[col];
      atomicAdd(out + col, v);
    }
    return;
  }
  if (ty == 0) __hip_atomic_store(partials + (size_t)blockIdx.y * kAuxColsumCols + blockIdx.x * 64 + tx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned ticket = __hip_atomic_fetch_add(counters + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = ticket == gridDim.y - 1;
    if (s_last) __hip_atomic_store(counters + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!s_last) return;                     // uniform for the workgroup
  // fixed shape: wave ty adds row blocks ty, ty + 4, ... in order, then the four waves in order
  float t = 0.f;
  if (col < n)
    for (unsigned b = ty; b < gridDim.y; b += 4)
      t += __hip_atomic_load(partials + (size_t)b * kAuxColsumCols + blockIdx.x * 64 + tx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  red[ty][tx] = t;                         // every thread read red[] before the barriers above
  __syncthreads();
  if (ty != 0 || col >= n) return;
  t = ((red[0][tx] + red[1][tx]) + red[2][tx]) + red[3][tx];
  if (gscale) t *= gscale[col];
  out[col] += t;                   // this workgroup is the only writer of its 64 columns in the launch
}

// ---- keras.optimizers.Adam (Keras 2 update rule) over a flat arena ------------------------------------------
// aff_*: optionally (radnet_adam_step_affine) the folded epilogue shifts of the convs whose biases live in [aff_off4, aff_off4 + aff_n4)
// float4 chunks of the arena are refreshed from the just-updated biases in the same pass: shift = scale * bias + t0
// (FixedBatchNormalization.py:59-85 folded; one launch fewer on the classifier lane per step).
// wz: optionally (radnet_adam_step_fused) 3x3 kernels [3][3][C][N] inside the arena whose Winograd F(4x4,3x3) transform U = G g G^T
// [36][C][N] is rewritten in the same launch.  Those kernels are taken out of the flat sweep and given to workgroups of their own, behind
// the sweep's in the grid: a workgroup owns 64 consecutive float4 chunks (c, 4 n) of a layer with all nine taps -- phase 1, every thread:
// the Adam update of its share of the 9 x 64 chunks, coalesced, new weights also into LDS; phase 2, one thread per chunk: the transform
// of its nine float4 values, wino4_filter_kernel's code, 36 coalesced 16-byte stores.  The transformed filters cost their own
// bytes (4x the kernels') and no launch (three launches cost the classifier lane as much as the Winograd forward gives: DESIGN.md 4).
constexpr int kAdamWinoMax = 12;
struct AdamWino {
  long long off4[kAdamWinoMax];   // first float4 of the layer's kernel in the arena
  int cn4[kAdamWinoMax];          // C * N / 4: float4 chunks per tap (a multiple of 64)
  int unit0[kAdamWinoMax + 1];    // first workgroup (relative to the first Winograd workgroup) of each layer; [n] = their total
  float* u[kAdamWinoMax];
  int n;
  unsigned sweep_blocks; // workgroups of the flat sweep (the Winograd workgroups follow)
};
__device__ __forceinline__ void adam_one(float4& pp, const float4& gg, float4& mm, float4& vv, float lr_t, float b1, float b2, float eps, float gs) {
#define ADAM1(q)                                         \
  {                                                      \
    float gq = gg.q * gs;                                \
    mm.q = b1 * mm.q + (1.f - b1) * gq;                  \
    vv.q = b2 * vv.q + (1.f - b2) * gq * gq;             \
    pp.q = pp.q - lr_t * mm.q / (sqrtf(vv.q) + eps);     \
  }
  ADAM1(x) ADAM1(y) ADAM1(z) ADAM1(w)
#undef ADAM1
}
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long long n4, float lr_t, float b1, float b2, float eps,
                                                   float gs, int zero_grad, long long aff_off4, long long aff_n4,
                                                   const float* __restrict__ aff_scale, const float* __restrict__ aff_t0, float* __restrict__ aff_shift,
                                                   AdamWino wz) {
  float4* p4 = reinterpret_cast<float4*>(p);
  float4* g4 = reinterpret_cast<float4*>(g);
  float4* m4 = reinterpret_cast<float4*>(m);
  float4* v4 = reinterpret_cast<float4*>(v);
  if (wz.n > 0 && blockIdx.x >= wz.sweep_blocks) {
    __shared__ float4 taps[9][64];
    const int unit = (int)(blockIdx.x - wz.sweep_blocks);
    int layer = 0;
    for (int l = 1; l < wz.n; ++l)
      if (unit >= wz.unit0[l]) layer = l;
    const int cn4 = wz.cn4[layer];
    const long long j0 = (long long)(unit - wz.unit0[layer]) * 64;
    for (int it = threadIdx.x; it < 9 * 64; it += 256) {
      const int tap = it >> 6, jj = it & 63;
      const long long k = wz.off4[layer] + (long long)tap * cn4 + j0 + jj;
      float4 pp = p4[k], mm = m4[k], vv = v4[k];
      adam_one(pp, g4[k], mm, vv, lr_t, b1, b2, eps, gs);
      p4[k] = pp; m4[k] = mm; v4[k] = vv;
      if (zero_grad) g4[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      taps[tap][jj] = pp;
    }
    __syncthreads();
    // phase 2: wino4_filter_kernel's own code on the same vector type (one thread per float4 chunk), so that both produce the same bits
    // (dealing the six output rows to three waves changed nothing measurable -- 84 against 85 us -- and the compiler's FMA choices with it)
    if (threadIdx.x < 64) {
      float4 t[6][3];
#pragma unroll
      for (int bb = 0; bb < 3; ++bb) {
        float4 col[3], o[6];
#pragma unroll
        for (int aa = 0; aa < 3; ++aa) col[aa] = taps[aa * 3 + bb][threadIdx.x];
        g6(col, o);
#pragma unroll
        for (int aa = 0; aa < 6; ++aa) t[aa][bb] = o[aa];
      }
      float4* dst = reinterpret_cast<float4*>(wz.u[layer]) + j0 + threadIdx.x;
#pragma unroll
      for (int aa = 0; aa < 6; ++aa) {
        float4 o[6];
        g6(t[aa], o);
#pragma unroll
        for (int bb = 0; bb < 6; ++bb) dst[(long long)(6 * aa + bb) * cn4] = o[bb];
      }
    }
    return;
  }
  const long long stride = (long long)(wz.n > 0 ? wz.sweep_blocks : gridDim.x) * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    bool skip = false;
    for (int l = 0; l < wz.n; ++l) skip |= i >= wz.off4[l] && i < wz.off4[l] + 9ll * wz.cn4[l];
    if (skip) continue;                          // a Winograd layer's kernel: updated by its own workgroups
    float4 pp = p4[i], mm = m4[i], vv = v4[i];
    adam_one(pp, g4[i], mm, vv, lr_t, b1, b2, eps, gs);
    p4[i] = pp;
    m4[i] = mm;
    v4[i] = vv;
    if (zero_grad) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (aff_shift != nullptr && i >= aff_off4 && i < aff_off4 + aff_n4) {
      const long long j = i - aff_off4;
      const float4 a = reinterpret_cast<const float4*>(aff_scale)[j], c = reinterpret_cast<const float4*>(aff_t0)[j];
      reinterpret_cast<float4*>(aff_shift)[j] = make_float4(a.x * pp.x + c.x, a.y * pp.y + c.y, a.z * pp.z + c.z, a.w * pp.w + c.w);
    }
  }
}

// ---- RPN losses (losses.py:16-66) ------------------------------------------------------------------------------
// scratch (double): [0] sum valid, [1] sum valid*ce, [2] sum mask, [3] sum mask*smoothL1
__device__ __forceinline__ float bce_swapped_logit(float t) {
  // Keras-2 K.binary_crossentropy(target=y_pred, output=y_true): logit of the clipped *label*
  const float lo = 1e-7f, hi = 1.0f - 1e-7f;
  float o = fminf(fmaxf(t, lo), hi);
  return logf(o / (1.0f - o));
}

__global__ void __launch_bounds__(256) rpn_loss_sums_kernel(const float* __restrict__ pred, int ld_pred, const float* __restrict__ ycls,
                                                            const float* __restrict__ yregr, int m, int a, int bce_mode,
                                                            double* __restrict__ scratch, double* __restrict__ partials) {
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  const long long total = (long long)m * 5 * a;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int row = (int)(idx / (5 * a)), col = (int)(idx - (long long)row * 5 * a);
    const float p = pred[(long long)row * ld_pred + col];
    if (col < a) {
      const float valid = ycls[(long long)row * 2 * a + col], t = ycls[(long long)row * 2 * a + a + col];
      float ce;
      if (bce_mode == 0) {
        const float l = bce_swapped_logit(t);
        ce = fmaxf(l, 0.f) - l * p + log1pf(expf(-fabsf(l)));
      } else {
        const float pc = fminf(fmaxf(p, 1e-7f), 1.0f - 1e-7f);
        const float z = logf(pc / (1.0f - pc));
        ce = fmaxf(z, 0.f) - z * t + log1pf(expf(-fabsf(z)));
      }
      s0 += valid;
      s1 += valid * ce;
    } else {
      const int j = col - a;
      const float mask = yregr[(long long)row * 8 * a + j], tgt = yregr[(long long)row * 8 * a + 4 * a + j];
      const float x = tgt - p, ax = fabsf(x);
      const float sl = ax <= 1.0f ? 0.5f * x * x : ax - 0.5f;
      s2 += mask;
      s3 += mask * sl;
    }
  }
  // wave shuffle -> LDS across the 4 waves -> ONE atomic per sum per block (contended fp64 atomics are ~10 ns each)
  __shared__ double red[4][4];
  s0 = wave_sum_d(s0); s1 = wave_sum_d(s1); s2 = wave_sum_d(s2); s3 = wave_sum_d(s3);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[wave][0] = s0; red[wave][1] = s1; red[wave][2] = s2; red[wave][3] = s3; }
  __syncthreads();
  if (partials == nullptr) {
    if (threadIdx.x < 4) atomicAdd(scratch + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
    return;
  }
  // ordered form: the last block to arrive adds the blocks' sums in a fixed shape (thread t takes blocks t, t + 256, ...; then the
  // wave shuffle tree; then the four waves in order) -- the same bits on every run, and one memory round trip instead of one per
  // block.  scratch[4] holds the arrival counter, zeroed with the rest of scratch by the launcher's memset.
  __shared__ int s_last;
  if (threadIdx.x < 4)
    __hip_atomic_store(partials + 4 * blockIdx.x + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x],
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned ticket = __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(scratch + 4), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = ticket == gridDim.x - 1;
  }
  __syncthreads();
  if (!s_last) return;                     // uniform for the workgroup
  double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
  for (unsigned b = threadIdx.x; b < gridDim.x; b += blockDim.x) {
    t0 += __hip_atomic_load(partials + 4 * b + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t1 += __hip_atomic_load(partials + 4 * b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t2 += __hip_atomic_load(partials + 4 * b + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t3 += __hip_atomic_load(partials + 4 * b + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  t0 = wave_sum_d(t0); t1 = wave_sum_d(t1); t2 = wave_sum_d(t2); t3 = wave_sum_d(t3);
  // red[] was last read before the two barriers above
  if (lane == 0) { red[wave][0] = t0; red[wave][1] = t1; red[wave][2] = t2; red[wave][3] = t3; }
  __syncthreads();
  if (threadIdx.x < 4) scratch[threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

__global__ void __launch_bounds__(256) rpn_loss_grad_kernel(const float* __restrict__ pred, int ld_pred, const float* __restrict__ ycls,
                                                            const float* __restrict__ yregr, int m, int a, int bce_mode,
                                                            const double* __restrict__ scratch, float* __restrict__ dz, int ld_dz,
                                                            float* __restrict__ losses) {
  const float den_c = (float)(1e-4 * (double)m * a + scratch[0]);
  const float den_r = (float)(1e-4 * (double)m * 4 * a + scratch[2]);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    losses[0] = (float)(scratch[1] / (double)den_c);
    losses[1] = (float)(scratch[3] / (double)den_r);
  }
  const long long total = (long long)m * ld_dz;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int row = (int)(idx / ld_dz), col = (int)(idx - (long long)row * ld_dz);
    float out = 0.f;
    if (col < a) {
      const float p = pred[(long long)row * ld_pred + col];
      const float valid = ycls[(long long)row * 2 * a + col], t = ycls[(long long)row * 2 * a + a + col];
      float dce;
      if (bce_mode == 0) {
        dce = -bce_swapped_logit(t);
      } else {
        const float lo = 1e-7f, hi = 1.0f - 1e-7f;
        const float pc = fminf(fmaxf(p, lo), hi);
        dce = (p >= lo && p <= hi) ? (pc - t) / (pc * (1.0f - pc)) : 0.f;
      }
      out = valid * dce / den_c * p * (1.0f - p);          // sigmoid' folded in
    } else if (col < 5 * a) {
      const int j = col - a;
      const float p = pred[(long long)row * ld_pred + col];
      const float mask = yregr[(long long)row * 8 * a + j], tgt = yregr[(long long)row * 8 * a + 4 * a + j];
      const float x = tgt - p, ax = fabsf(x);
      const float d = ax <= 1.0f ? x : (x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f));
      out = -(mask * d) / den_r;
    }
    dz[idx] = out;
  }
}

// ---- detector losses (losses.py:69-95), single workgroup ---------------------------------------------------------
__global__ void __launch_bounds__(256) det_loss_kernel(const float* __restrict__ pcls, const float* __restrict__ pregr,
                                                       const float* __restrict__ y1, const float* __restrict__ y2, int r, int nc,
                                                       int nreg, float* __restrict__ dz, float* __restrict__ losses) {
  __shared__ double red[4][4];
  __shared__ float s_den;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float lo = 1e-7f, hi = 1.0f - 1e-7f;
  double ce_sum = 0, acc_sum = 0, mask_sum = 0, sl_sum = 0;
  for (int rr = tid; rr < r; rr += blockDim.x) {
    const float* q = pcls + (long long)rr * nc;
    const float* t = y1 + (long long)rr * nc;
    float S = 0.f;
    for (int i = 0; i < nc; ++i) S += q[i];
    float ce = 0.f;
    int am_t = 0, am_q = 0;
    for (int i = 0; i < nc; ++i) {
      float oc = fminf(fmaxf(q[i] / S, lo), hi);
      ce -= t[i] * logf(oc);
      if (t[i] > t[am_t]) am_t = i;
      if (q[i] > q[am_q]) am_q = i;
    }
    ce_sum += ce;
    acc_sum += (am_t == am_q) ? 1.0 : 0.0;
    for (int j = 0; j < nreg; ++j) {
      const float mask = y2[(long long)rr * 2 * nreg + j], tgt = y2[(long long)rr * 2 * nreg + nreg + j];
      const float x = tgt - pregr[(long long)rr * nreg + j], ax = fabsf(x);
      mask_sum += mask;
      sl_sum += mask * (ax <= 1.0f ? 0.5f * x * x : ax - 0.5f);
    }
  }
  ce_sum = wave_sum_d(ce_sum); acc_sum = wave_sum_d(acc_sum); mask_sum = wave_sum_d(mask_sum); sl_sum = wave_sum_d(sl_sum);
  if (lane == 0) { red[wave][0] = ce_sum; red[wave][1] = acc_sum; red[wave][2] = mask_sum; red[wave][3] = sl_sum; }
  __syncthreads();
  if (tid == 0) {
    double c = 0, a = 0, ms = 0, sl = 0;
    for (int wv = 0; wv < 4; ++wv) { c += red[wv][0]; a += red[wv][1]; ms += red[wv][2]; sl += red[wv][3]; }
    const float den = (float)(1e-4 * (double)r * nreg + ms);
    s_den = den;
    losses[0] = (float)(c / r);
    losses[1] = (float)(sl / (double)den);
    losses[2] = (float)(a / r);
  }
  __syncthreads();
  const float den = s_den;
  const int nout = nc + nreg;
  for (int rr = tid; rr < r; rr += blockDim.x) {
    const float* q = pcls + (long long)rr * nc;
    const float* t = y1 + (long long)rr * nc;
    float S = 0.f;
    for (int i = 0; i < nc; ++i) S += q[i];
    // a_k = -t_k*inrange_k/oc_k ; dq_j = (a_j - sum_k a_k o_k)/S/R ; dlogit_i = q_i (dq_i - sum_j dq_j q_j)
    float sum_ao = 0.f;
    for (int i = 0; i < nc; ++i) {
      float o = q[i] / S, oc = fminf(fmaxf(o, lo), hi);
      float ak = (o >= lo && o <= hi) ? -t[i] / oc : 0.f;
      sum_ao += ak * o;
    }
    float sum_dqq = 0.f;
    for (int i = 0; i < nc; ++i) {
      float o = q[i] / S, oc = fminf(fmaxf(o, lo), hi);
      float ak = (o >= lo && o <= hi) ? -t[i] / oc : 0.f;
      sum_dqq += (ak - sum_ao) / S / (float)r * q[i];
    }
    for (int i = 0; i < nc; ++i) {
      float o = q[i] / S, oc = fminf(fmaxf(o, lo), hi);
      float ak = (o >= lo && o <= hi) ? -t[i] / oc : 0.f;
      float dq = (ak - sum_ao) / S / (float)r;
      dz[(long long)rr * nout + i] = q[i] * (dq - sum_dqq);
    }
    for (int j = 0; j < nreg; ++j) {
      const float mask = y2[(long long)rr * 2 * nreg + j], tgt = y2[(long long)rr * 2 * nreg + nreg + j];
      const float x = tgt - pregr[(long long)rr * nreg + j], ax = fabsf(x);
      const float d = ax <= 1.0f ? x : (x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f));
      dz[(long long)rr * nout + nc + j] = -(mask * d) / den;
    }
  }
}

// ---- misc --------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) preprocess_kernel(const uint8_t* __restrict__ img, long long npix, int cpad, float* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long long)gridDim.x * blockDim.x) {
    const float b = (float)img[3 * i + 0] - 103.939f, g = (float)img[3 * i + 1] - 116.779f, r = (float)img[3 * i + 2] - 123.68f;
    float* o = out + i * cpad;
    o[0] = b; o[1] = g; o[2] = r;
    for (int c = 3; c < cpad; ++c) o[c] = 0.f;
  }
}

__global__ void __launch_bounds__(256) scale_kernel(float* __restrict__ x, long long n, float alpha) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) x[i] *= alpha;
}

__global__ void __launch_bounds__(256) affine_vec_kernel(float* __restrict__ out, const float* __restrict__ a, const float* __restrict__ b,
                                                         const float* __restrict__ c, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) out[i] = a[i] * b[i] + c[i];
}

inline int grid_for(long long total, int block = 256, int cap = 4096) {
  long long b = (total + block - 1) / block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

}  // namespace

extern "C" int radnet_maxpool_fwd(radnet_ctx* ctx, const float* x, float* y, int32_t nb, int32_t h, int32_t w, int32_t c, int32_t k, int32_t s) {
  if (!ctx || !x || !y) return RADNET_ERR_ARG;
  if (c % 4) RADNET_FAIL(ctx, RADNET_ERR_ARG, "maxpool: c=%d not a multiple of 4", c);
  const int oh = (h - k) / s + 1, ow = (w - k) / s + 1;
  if (oh <= 0 || ow <= 0) RADNET_FAIL(ctx, RADNET_ERR_ARG, "maxpool: empty output");
  const long long total = (long long)nb * oh * ow * (c / 4);
  if (total >= (1ll << 31)) RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "maxpool: %lld work items (32-bit index arithmetic)", total);
  hipLaunchKernelGGL(maxpool_kernel, dim3(grid_for(total)), dim3(256), 0, ctx->stream, x, y, nb, h, w, c / 4, oh, ow, k, s);
  RADNET_CHECK_LAUNCH(ctx, "maxpool");
  return RADNET_OK;
}

extern "C" int radnet_roi_resize_fwd(radnet_ctx* ctx, const float* fmap, int32_t h, int32_t w, int32_t c, const float* rois, int32_t r,
                                     int32_t ps, float* y) {
  if (!ctx || !fmap || !rois || !y) return RADNET_ERR_ARG;
  if (c % 4 || r <= 0 || ps <= 0) RADNET_FAIL(ctx, RADNET_ERR_ARG, "roi_resize: bad c=%d r=%d ps=%d", c, r, ps);
  const int threads = (c / 4) >= 256 ? 256 : (((c / 4) + 63) / 64) * 64;
  hipLaunchKernelGGL(roi_resize_fwd_kernel, dim3(r * ps * ps), dim3(threads), 0, ctx->stream, fmap, h, w, c / 4, rois, ps, y);
  RADNET_CHECK_LAUNCH(ctx, "roi_resize_fwd");
  return RADNET_OK;
}

extern "C" int radnet_roi_resize_bwd(radnet_ctx* ctx, const float* dy, int32_t h, int32_t w, int32_t c, const float* rois, int32_t r,
                                     int32_t ps, float* dfmap) {
  if (!ctx || !dy || !rois || !dfmap) return RADNET_ERR_ARG;
  if (ctx->deterministic && (c % 4) == 0 && ps <= 32 && (long long)h * w < (1ll << 31))
    hipLaunchKernelGGL(roi_resize_bwd_ordered_kernel, dim3(h * w), dim3(256), 0, ctx->stream, dy, h, w, c / 4, rois, r, ps, dfmap);
  else
    hipLaunchKernelGGL(roi_resize_bwd_kernel, dim3(r * ps * ps), dim3(256), 0, ctx->stream, dy, h, w, c, rois, ps, dfmap);
  RADNET_CHECK_LAUNCH(ctx, "roi_resize_bwd");
  return RADNET_OK;
}

extern "C" int radnet_avgpool_fwd(radnet_ctx* ctx, const float* x, int32_t r, int32_t hw, int32_t c, float* y) {
  if (!ctx || !x || !y) return RADNET_ERR_ARG;
  if (c % 4) RADNET_FAIL(ctx, RADNET_ERR_ARG, "avgpool: c %% 4");
  hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(radnet_cdiv((long long)r * (c / 4), 256)), dim3(256), 0, ctx->stream, x, r, hw, c / 4, y);
  RADNET_CHECK_LAUNCH(ctx, "avgpool_fwd");
  return RADNET_OK;
}

extern "C" int radnet_avgpool_bwd_relu(radnet_ctx* ctx, const float* dfeat, const float* y_act, int32_t r, int32_t hw, int32_t c, float* dx) {
  if (!ctx || !dfeat || !y_act || !dx) return RADNET_ERR_ARG;
  if (c % 4) RADNET_FAIL(ctx, RADNET_ERR_ARG, "avgpool_bwd: c %% 4");
  if ((long long)r * hw * (c / 4) >= (1ll << 31)) RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "avgpool_bwd: too many work items (32-bit index arithmetic)");
  hipLaunchKernelGGL(avgpool_bwd_relu_kernel, dim3(grid_for((long long)r * hw * (c / 4))), dim3(256), 0, ctx->stream, dfeat, y_act, r, hw,
                     c / 4, dx);
  RADNET_CHECK_LAUNCH(ctx, "avgpool_bwd_relu");
  return RADNET_OK;
}

extern "C" int radnet_dense_heads_fwd(radnet_ctx* ctx, const float* feat, int32_t r, int32_t k, const float* w, int32_t ldw, const float* b,
                                      int32_t nc, int32_t nreg, float* out_cls, float* out_regr) {
  if (!ctx || !feat || !w || !b || !out_cls || !out_regr) return RADNET_ERR_ARG;
  if (nc + nreg > ldw || (ldw != 32 && ldw != 64)) RADNET_FAIL(ctx, RADNET_ERR_ARG, "dense_heads: ldw=%d must be 32 or 64 and >= nc+nreg=%d", ldw, nc + nreg);
  if (ldw == 32) hipLaunchKernelGGL(dense_heads_fwd_kernel<32>, dim3(r), dim3(256), 0, ctx->stream, feat, k, w, b, nc, nreg, out_cls, out_regr);
  else hipLaunchKernelGGL(dense_heads_fwd_kernel<64>, dim3(r), dim3(256), 0, ctx->stream, feat, k, w, b, nc, nreg, out_cls, out_regr);
  RADNET_CHECK_LAUNCH(ctx, "dense_heads_fwd");
  return RADNET_OK;
}

extern "C" int radnet_dense_heads_bwd(radnet_ctx* ctx, const float* feat, const float* dz, int32_t r, int32_t k, const float* w, int32_t ldw,
                                      int32_t nout, float* dw, float* db, float* dfeat, int32_t accumulate) {
  if (!ctx || !feat || !dz || !w || !dw || !db || !dfeat) return RADNET_ERR_ARG;
  const size_t smem = (size_t)r * ldw * sizeof(float);
  if (smem > 64 * 1024) RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "dense_heads_bwd: r=%d too large", r);
  hipLaunchKernelGGL(dense_heads_bwd_kernel, dim3(radnet_cdiv(k, 8)), dim3(256), smem, ctx->stream, feat, dz, r, k, w, ldw, nout, dw, db, dfeat, accumulate);
  RADNET_CHECK_LAUNCH(ctx, "dense_heads_bwd");
  return RADNET_OK;
}

extern "C" int radnet_colsum(radnet_ctx* ctx, const float* g, int32_t m, int32_t n, int32_t ld, const float* gscale, float* out,
                             int32_t accumulate) {
  if (!ctx || !g || !out) return RADNET_ERR_ARG;
  if (!accumulate) RADNET_CHECK_HIP(ctx, hipMemsetAsync(out, 0, (size_t)n * sizeof(float), ctx->stream));
  int rows_per_block = 128;
  float* partials = nullptr;
  if (ctx->deterministic && m > rows_per_block) {
    if (n > (int)kAuxColsumCols || radnet_cdiv(n, 64) > (int)kAuxColsumCounterCount) RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "colsum: n=%d", n);
    while (radnet_cdiv(m, rows_per_block) > (int)kAuxColsumRows) rows_per_block *= 2;
    partials = reinterpret_cast<float*>(ctx->aux + kAuxColsumScratch);
  }
  hipLaunchKernelGGL(colsum_kernel, dim3(radnet_cdiv(n, 64), radnet_cdiv(m, rows_per_block)), dim3(256), 0, ctx->stream, g, m, n, ld, gscale,
                     out, rows_per_block, partials, reinterpret_cast<unsigned*>(ctx->aux + kAuxColsumCounters));
  RADNET_CHECK_LAUNCH(ctx, "colsum");
  return RADNET_OK;
}

extern "C" int radnet_adam_step(radnet_ctx* ctx, float* p, float* g, float* m, float* v, int64_t n, int32_t t, float lr, float beta1,
                                float beta2, float eps, float grad_scale, int32_t zero_grad) {
  if (!ctx || !p || !g || !m || !v) return RADNET_ERR_ARG;
  if (n % 4) RADNET_FAIL(ctx, RADNET_ERR_ARG, "adam: arena length must be a multiple of 4");
  if (t < 1) RADNET_FAIL(ctx, RADNET_ERR_ARG, "adam: step counter starts at 1");
  // lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t)   (keras.optimizers.Adam.get_updates)
  const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, (double)t)) / (1.0 - pow((double)beta1, (double)t));
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n / 4, 256, 8192)), dim3(256), 0, ctx->stream, p, g, m, v, (long long)(n / 4), (float)lr_t,
                     beta1, beta2, eps, grad_scale, (int)zero_grad, 0ll, 0ll, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, AdamWino{});
  RADNET_CHECK_LAUNCH(ctx, "adam");
  return RADNET_OK;
}

extern "C" int radnet_adam_step_affine(radnet_ctx* ctx, float* p, float* g, float* m, float* v, int64_t n, int32_t t, float lr, float beta1,
                                       float beta2, float eps, float grad_scale, int32_t zero_grad, int64_t bias_off, int64_t bias_len,
                                       const float* scale, const float* t0, float* shift) {
  if (!ctx || !p || !g || !m || !v || !scale || !t0 || !shift) return RADNET_ERR_ARG;
  if ((n % 4) || (bias_off % 4) || (bias_len % 4) || bias_off < 0 || bias_len < 0 || bias_off + bias_len > n)
    RADNET_FAIL(ctx, RADNET_ERR_ARG, "adam_affine: arena length %lld, bias range [%lld, +%lld) must be multiples of 4 inside the arena", (long long)n,
                (long long)bias_off, (long long)bias_len);
  if (((uintptr_t)scale | (uintptr_t)t0 | (uintptr_t)shift) & 15) RADNET_FAIL(ctx, RADNET_ERR_ARG, "adam_affine: scale / t0 / shift must be 16-byte aligned");
  if (t < 1) RADNET_FAIL(ctx, RADNET_ERR_ARG, "adam: step counter starts at 1");
  const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, (double)t)) / (1.0 - pow((double)beta1, (double)t));
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n / 4, 256, 8192)), dim3(256), 0, ctx->stream, p, g, m, v, (long long)(n / 4), (float)lr_t,
                     beta1, beta2, eps, grad_scale, (int)zero_grad, (long long)(bias_off / 4), (long long)(bias_len / 4), scale, t0, shift, AdamWino{});
  RADNET_CHECK_LAUNCH(ctx, "adam_affine");
  return RADNET_OK;
}

extern "C" int radnet_adam_step_fused(radnet_ctx* ctx, float* p, float* g, float* m, float* v, int64_t n, int32_t t, float lr, float beta1,
                                      float beta2, float eps, float grad_scale, int32_t zero_grad, int64_t bias_off, int64_t bias_len,
                                      const float* scale, const float* t0, float* shift, const radnet_adam_wino* layers, int32_t n_layers) {
  if (!ctx || !p || !g || !m || !v || n_layers < 0 || n_layers > kAdamWinoMax || (n_layers > 0 && !layers)) return RADNET_ERR_ARG;
  if (shift != nullptr && (!scale || !t0)) return RADNET_ERR_ARG;
  if ((n % 4) || (bias_off % 4) || (bias_len % 4) || bias_off < 0 || bias_len < 0 || bias_off + bias_len > n)
    RADNET_FAIL(ctx, RADNET_ERR_ARG, "adam_fused: arena length %lld, bias range [%lld, +%lld) must be multiples of 4 inside the arena", (long long)n,
                (long long)bias_off, (long long)bias_len);
  if (shift && (((uintptr_t)scale | (uintptr_t)t0 | (uintptr_t)shift) & 15)) RADNET_FAIL(ctx, RADNET_ERR_ARG, "adam_fused: scale / t0 / shift must be 16-byte aligned");
  if (t < 1) RADNET_FAIL(ctx, RADNET_ERR_ARG, "adam: step counter starts at 1");
  AdamWino wz{};
  wz.n = n_layers;
  long long in_layers = 0;
  for (int l = 0; l < n_layers; ++l) {
    const radnet_adam_wino& d = layers[l];
    const int64_t len = 9ll * d.c * d.n;
    if (!d.u || d.c <= 0 || d.n <= 0 || (d.n & 3) || (d.off & 3) || d.off < 0 || d.off + len > n || ((uintptr_t)d.u & 15) || (int64_t)d.c * d.n / 4 >= (1ll << 28) ||
        ((int64_t)d.c * d.n / 4) % 64)
      RADNET_FAIL(ctx, RADNET_ERR_ARG, "adam_fused: layer %d (offset %lld, c %d, n %d) does not describe a dense [3][3][c][n] kernel inside the arena with c*n a multiple of 256",
                  l, (long long)d.off, d.c, d.n);
    if (shift && d.off < bias_off + bias_len && bias_off < d.off + len) RADNET_FAIL(ctx, RADNET_ERR_ARG, "adam_fused: layer %d overlaps the bias range", l);
    for (int k = 0; k < l; ++k)
      if (d.off < layers[k].off + 9ll * layers[k].c * layers[k].n && layers[k].off < d.off + len) RADNET_FAIL(ctx, RADNET_ERR_ARG, "adam_fused: layers %d and %d overlap", k, l);
    wz.off4[l] = d.off / 4;
    wz.cn4[l] = (int)((int64_t)d.c * d.n / 4);
    wz.u[l] = d.u;
    wz.unit0[l] = l == 0 ? 0 : wz.unit0[l - 1] + wz.cn4[l - 1] / 64;
    in_layers += len / 4;
  }
  wz.unit0[n_layers] = n_layers ? wz.unit0[n_layers - 1] + wz.cn4[n_layers - 1] / 64 : 0;
  const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, (double)t)) / (1.0 - pow((double)beta1, (double)t));
  const unsigned sweep = (unsigned)grid_for(std::max<long long>(n / 4 - in_layers, 1), 256, 8192);
  wz.sweep_blocks = sweep;
  hipLaunchKernelGGL(adam_kernel, dim3(sweep + (unsigned)wz.unit0[n_layers]), dim3(256), 0, ctx->stream, p, g, m, v, (long long)(n / 4), (float)lr_t,
                     beta1, beta2, eps, grad_scale, (int)zero_grad, (long long)(bias_off / 4), (long long)(shift ? bias_len / 4 : 0), scale, t0, shift, wz);
  RADNET_CHECK_LAUNCH(ctx, "adam_fused");
  return RADNET_OK;
}

extern "C" int radnet_rpn_loss(radnet_ctx* ctx, const float* pred, int32_t ld_pred, const float* y_cls, const float* y_regr, int32_t m,
                               int32_t a, int32_t bce_mode, float* dz, int32_t ld_dz, float* losses, double* scratch8) {
  if (!ctx || !pred || !y_cls || !y_regr || !dz || !losses || !scratch8) return RADNET_ERR_ARG;
  if (ld_pred < 5 * a || ld_dz < 5 * a) RADNET_FAIL(ctx, RADNET_ERR_ARG, "rpn_loss: leading dims too small");
  RADNET_CHECK_HIP(ctx, hipMemsetAsync(scratch8, 0, 8 * sizeof(double), ctx->stream));
  hipLaunchKernelGGL(rpn_loss_sums_kernel, dim3(grid_for((long long)m * 5 * a, 256, 256)), dim3(256), 0, ctx->stream, pred, ld_pred, y_cls,
                     y_regr, m, a, bce_mode, scratch8, ctx->deterministic ? reinterpret_cast<double*>(ctx->aux + kAuxLossPartials) : nullptr);
  RADNET_CHECK_LAUNCH(ctx, "rpn_loss_sums");
  hipLaunchKernelGGL(rpn_loss_grad_kernel, dim3(grid_for((long long)m * ld_dz, 256, 2048)), dim3(256), 0, ctx->stream, pred, ld_pred, y_cls,
                     y_regr, m, a, bce_mode, scratch8, dz, ld_dz, losses);
  RADNET_CHECK_LAUNCH(ctx, "rpn_loss_grad");
  return RADNET_OK;
}

extern "C" int radnet_det_loss(radnet_ctx* ctx, const float* p_cls, const float* p_regr, const float* y1, const float* y2, int32_t r,
                               int32_t nc, int32_t nreg, float* dz, float* losses) {
  if (!ctx || !p_cls || !p_regr || !y1 || !y2 || !dz || !losses) return RADNET_ERR_ARG;
  hipLaunchKernelGGL(det_loss_kernel, dim3(1), dim3(256), 0, ctx->stream, p_cls, p_regr, y1, y2, r, nc, nreg, dz, losses);
  RADNET_CHECK_LAUNCH(ctx, "det_loss");
  return RADNET_OK;
}

extern "C" int radnet_preprocess_bgr(radnet_ctx* ctx, const uint8_t* img, int32_t h, int32_t w, int32_t cpad, float* out) {
  if (!ctx || !img || !out || cpad < 3) return RADNET_ERR_ARG;
  hipLaunchKernelGGL(preprocess_kernel, dim3(grid_for((long long)h * w)), dim3(256), 0, ctx->stream, img, (long long)h * w, cpad, out);
  RADNET_CHECK_LAUNCH(ctx, "preprocess");
  return RADNET_OK;
}

namespace {
// 16-byte lanes, then the tail byte by byte.  Either side may be pinned host memory (mapped into the device's address space).
__global__ void __launch_bounds__(256) copy_bytes_kernel(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, unsigned long long n16,
                                                         unsigned long long bytes) {
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride)
    reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
  for (unsigned long long i = n16 * 16 + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < bytes; i += stride) dst[i] = src[i];
}
}  // namespace

extern "C" int radnet_copy_bytes(radnet_ctx* ctx, void* dst, const void* src, uint64_t bytes) {
  if (!ctx || (bytes && (!dst || !src))) return RADNET_ERR_ARG;
  if (!bytes) return RADNET_OK;
  const bool aligned = (((uintptr_t)dst | (uintptr_t)src) & 15) == 0;
  const unsigned long long n16 = aligned ? bytes / 16 : 0;
  hipLaunchKernelGGL(copy_bytes_kernel, dim3(grid_for((long long)(n16 ? n16 : bytes), 256, 2048)), dim3(256), 0, ctx->stream, (uint8_t*)dst, (const uint8_t*)src,
                     n16, (unsigned long long)bytes);
  RADNET_CHECK_LAUNCH(ctx, "copy_bytes");
  return RADNET_OK;
}

extern "C" int radnet_fill_zero(radnet_ctx* ctx, void* p, uint64_t bytes) {
  if (!ctx || !p) return RADNET_ERR_ARG;
  RADNET_CHECK_HIP(ctx, hipMemsetAsync(p, 0, bytes, ctx->stream));
  return RADNET_OK;
}

extern "C" int radnet_scale(radnet_ctx* ctx, float* x, int64_t n, float alpha) {
  if (!ctx || !x) return RADNET_ERR_ARG;
  hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n)), dim3(256), 0, ctx->stream, x, (long long)n, alpha);
  RADNET_CHECK_LAUNCH(ctx, "scale");
  return RADNET_OK;
}

namespace {
__global__ void __launch_bounds__(256) relu_mask_kernel(float* __restrict__ g, const float* __restrict__ act, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    if (!(act[i] > 0.f)) g[i] = 0.f;
}
}  // namespace

namespace {
// dst pixel (ih, iw) of the full grid takes the compact source pixel (ih/s, iw/s) when both coordinates are multiples
// of s (and inside the compact grid), zero otherwise; then the optional ReLU mask of the producer.
__global__ void __launch_bounds__(256) scatter_strided_kernel(const float4* __restrict__ src, int nb, int oh, int ow, int c4, int s, int h, int w,
                                                              const float4* __restrict__ mask, float4* __restrict__ dst) {
  const long long total = (long long)nb * h * w * c4;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total; i += gridDim.x * blockDim.x) {      // total < 2^31 (launcher)
    const unsigned p = i / (unsigned)c4;
    const int cc = (int)(i - p * (unsigned)c4);
    const unsigned q = p / (unsigned)w;
    const int iw = (int)(p - q * (unsigned)w);
    const int img = (int)(q / (unsigned)h);
    const int ih = (int)(q - (unsigned)img * (unsigned)h);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const int qh = ih / s, qw = iw / s;
    if (qh * s == ih && qw * s == iw && qh < oh && qw < ow) v = src[(((long long)img * oh + qh) * ow + qw) * c4 + cc];
    if (mask != nullptr) {
      const float4 m = mask[i];
      v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
    }
    dst[i] = v;
  }
}
}  // namespace

extern "C" int radnet_scatter_strided(radnet_ctx* ctx, const float* src, int32_t nb, int32_t oh, int32_t ow, int32_t c, int32_t stride,
                                      int32_t h, int32_t w, const float* mask, float* dst) {
  if (!ctx || !src || !dst) return RADNET_ERR_ARG;
  if (c % 4 || stride < 1 || (oh - 1) * stride >= h || (ow - 1) * stride >= w)
    RADNET_FAIL(ctx, RADNET_ERR_ARG, "scatter_strided: c=%d stride=%d grid %dx%d into %dx%d", c, stride, oh, ow, h, w);
  const long long total = (long long)nb * h * w * (c / 4);
  if (total >= (1ll << 31)) RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "scatter_strided: %lld work items (32-bit index arithmetic)", total);
  hipLaunchKernelGGL(scatter_strided_kernel, dim3(grid_for(total, 256, 8192)), dim3(256), 0, ctx->stream, (const float4*)src, nb, oh, ow, c / 4,
                     stride, h, w, (const float4*)mask, (float4*)dst);
  RADNET_CHECK_LAUNCH(ctx, "scatter_strided");
  return RADNET_OK;
}

extern "C" int radnet_relu_mask(radnet_ctx* ctx, float* g, const float* act, int64_t n) {
  if (!ctx || !g || !act) return RADNET_ERR_ARG;
  long long b = (n + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  hipLaunchKernelGGL(relu_mask_kernel, dim3((int)b), dim3(256), 0, ctx->stream, g, act, (long long)n);
  RADNET_CHECK_LAUNCH(ctx, "relu_mask");
  return RADNET_OK;
}

extern "C" int radnet_affine_vec(radnet_ctx* ctx, float* out, const float* a, const float* b, const float* c, int64_t n) {
  if (!ctx || !out || !a || !b || !c) return RADNET_ERR_ARG;
  hipLaunchKernelGGL(affine_vec_kernel, dim3(grid_for(n)), dim3(256), 0, ctx->stream, out, a, b, c, (long long)n);
  RADNET_CHECK_LAUNCH(ctx, "affine_vec");
  return RADNET_OK;
}
