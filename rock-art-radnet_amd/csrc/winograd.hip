// Winograd F(2x2, 3x3) transforms for the stride-1 'same' 3x3 convolutions (resnet50.py:53,104: conv_block /
// identity_block 2b; rpn.py:41-48 rpn_conv1).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A      per 2x2 output tile, 4x4 input patch d, 3x3 filter g, per (c, n)
//
// turns 36 multiplications per output tile and channel pair into 16: the 16 element-wise products, summed over the
// input channels, are 16 independent GEMMs [tiles x C] x [C x N] on the fp32 matrix cores (radnet_gemm_batched), 2.25x
// fewer MFMA flops than the direct implicit GEMM at the price of three HBM-bound passes (input transform 1 read + 4x
// write, output transform 4x read + 1 write, filter transform once per weight update).  It pays where K = 9*C is large
// against the activation size: rpn_conv1 (C = 1024) and the stage-5 3x3 convs (C = 512).
// Arithmetic: the transforms are exact in the sense of using only additions and multiplications by 1/2 (B and A have
// entries 0, +-1; G has 1/2); results differ from the direct sum by fp32 rounding (different association), well inside
// the stated 1e-3 activation tolerance (tests compare against the oracle's direct convolution).
#include "radnet_internal.h"
#include "radnet_wino4.h"

namespace {

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4half(float4 a) { return make_float4(0.5f * a.x, 0.5f * a.y, 0.5f * a.z, 0.5f * a.w); }

// U[p][c][n] = (G g G^T)[p], p = 4*xi + nu.  g: [3][3][C][ldw] (Keras HWIO flattened), one thread per (c, 4 n).
__global__ void __launch_bounds__(256) wino_filter_kernel(const float* __restrict__ g, int C, int N, int ldw, float* __restrict__ U) {
  const int n4 = N >> 2;
  const long long total = (long long)C * n4;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total; i += gridDim.x * blockDim.x) {      // total < 2^31 (launcher)
    const int c = (int)(i / (unsigned)n4), nq = (int)(i - (unsigned)c * (unsigned)n4);
    float4 w[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) w[a][b] = *reinterpret_cast<const float4*>(g + ((long long)(a * 3 + b) * C + c) * ldw + nq * 4);
    // t = G w  (4x3), rows: w0, (w0+w1+w2)/2, (w0-w1+w2)/2, w2
    float4 t[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      t[0][b] = w[0][b];
      t[1][b] = f4half(f4add(f4add(w[0][b], w[1][b]), w[2][b]));
      t[2][b] = f4half(f4add(f4sub(w[0][b], w[1][b]), w[2][b]));
      t[3][b] = w[2][b];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float4 u0 = t[a][0], u1 = f4half(f4add(f4add(t[a][0], t[a][1]), t[a][2])), u2 = f4half(f4add(f4sub(t[a][0], t[a][1]), t[a][2])), u3 = t[a][2];
      float4* dst = reinterpret_cast<float4*>(U + ((long long)(a * 4) * C + c) * N + nq * 4);
      const long long ps = (long long)C * N / 4;      // float4 stride between consecutive p
      dst[0] = u0; dst[ps] = u1; dst[2 * ps] = u2; dst[3 * ps] = u3;
    }
  }
}

// V[p][tile][c] = (B^T d B)[p]; tile = (img, ti, tj) covers outputs (2ti..2ti+1, 2tj..2tj+1), input patch rows
// 2ti-1 .. 2ti+2 (pad 1).  One thread per (tile, 4 channels).
__global__ void __launch_bounds__(256) wino_input_kernel(const float* __restrict__ x, int nb, int H, int W, int C, int TH, int TW,
                                                         float* __restrict__ V) {
  const int c4 = C >> 2;
  const long long T = (long long)nb * TH * TW, total = T * c4;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total; i += gridDim.x * blockDim.x) {      // total < 2^31 (launcher)
    const unsigned tile = i / (unsigned)c4;                     // 32-bit index arithmetic: 64-bit divisions cost microseconds here
    const int cq = (int)(i - tile * (unsigned)c4);
    const unsigned trow = tile / (unsigned)TW;
    const int tj = (int)(tile - trow * (unsigned)TW);
    const int img = (int)(trow / (unsigned)TH);
    const int ti = (int)(trow - (unsigned)img * (unsigned)TH);
    float4 d[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int ih = 2 * ti - 1 + a;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int iw = 2 * tj - 1 + b;
        d[a][b] = ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W)
                      ? *reinterpret_cast<const float4*>(x + (((long long)img * H + ih) * W + iw) * C + cq * 4)
                      : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    // t = B^T d: rows d0-d2, d1+d2, d2-d1, d1-d3
    float4 t[4][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      t[0][b] = f4sub(d[0][b], d[2][b]);
      t[1][b] = f4add(d[1][b], d[2][b]);
      t[2][b] = f4sub(d[2][b], d[1][b]);
      t[3][b] = f4sub(d[1][b], d[3][b]);
    }
    const long long ps = T * c4;                        // float4 stride between consecutive p
    float4* dst = reinterpret_cast<float4*>(V) + tile * c4 + cq;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      dst[(4 * a + 0) * ps] = f4sub(t[a][0], t[a][2]);
      dst[(4 * a + 1) * ps] = f4add(t[a][1], t[a][2]);
      dst[(4 * a + 2) * ps] = f4sub(t[a][2], t[a][1]);
      dst[(4 * a + 3) * ps] = f4sub(t[a][1], t[a][3]);
    }
  }
}

// y = act( (A^T m A) * scale + shift ) for the 2x2 outputs of each tile; Mm: [16][T][N].  One thread per (tile, 4 n).
__global__ void __launch_bounds__(256) wino_output_kernel(const float* __restrict__ Mm, int nb, int OH, int OW, int N, int TH, int TW,
                                                          const float* __restrict__ scale, const float* __restrict__ shift, int act,
                                                          float* __restrict__ y, int ldy) {
  const int n4 = N >> 2;
  const long long T = (long long)nb * TH * TW, total = T * n4;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total; i += gridDim.x * blockDim.x) {      // total < 2^31 (launcher)
    const unsigned tile = i / (unsigned)n4;                     // 32-bit index arithmetic: 64-bit divisions cost microseconds here
    const int nq = (int)(i - tile * (unsigned)n4);
    const unsigned trow = tile / (unsigned)TW;
    const int tj = (int)(tile - trow * (unsigned)TW);
    const int img = (int)(trow / (unsigned)TH);
    const int ti = (int)(trow - (unsigned)img * (unsigned)TH);
    const long long ps = T * n4;
    const float4* src = reinterpret_cast<const float4*>(Mm) + tile * n4 + nq;
    float4 m[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) m[a][b] = src[(4 * a + b) * ps];
    // t = A^T m: rows m0+m1+m2, m1-m2-m3
    float4 t[2][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      t[0][b] = f4add(f4add(m[0][b], m[1][b]), m[2][b]);
      t[1][b] = f4sub(f4sub(m[1][b], m[2][b]), m[3][b]);
    }
    const float4 sc = scale ? *reinterpret_cast<const float4*>(scale + nq * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 sh = shift ? *reinterpret_cast<const float4*>(shift + nq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int oh = 2 * ti + a;
      if (oh >= OH) continue;
      const float4 o[2] = {f4add(f4add(t[a][0], t[a][1]), t[a][2]), f4sub(f4sub(t[a][1], t[a][2]), t[a][3])};
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int ow = 2 * tj + b;
        if (ow >= OW) continue;
        float4 v = make_float4(o[b].x * sc.x + sh.x, o[b].y * sc.y + sh.y, o[b].z * sc.z + sh.z, o[b].w * sc.w + sh.w);
        if (act == 1) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
        *reinterpret_cast<float4*>(y + (((long long)img * OH + oh) * OW + ow) * ldy + nq * 4) = v;
      }
    }
  }
}

// ---- weight gradient in the Winograd domain -------------------------------------------------------------------------
// Y = A^T Z A, Z[p] = sum_c U[p] V[p]  =>  dU[p][c][n] = sum_tiles V[p][tile][c] * dZ[p][tile][n] with dZ = A dY A^T (16
// reduction-over-tiles GEMMs, radnet_wgrad_batched, on the V the forward pass already produced), then dg = G^T dU G.
// dZ[p][tile][n] for the 2x2 output-gradient block of each tile (outputs past the edge contribute 0), dy scaled by the
// per-channel factor gscale (frozen-BN scale) when given.
__global__ void __launch_bounds__(256) wino_dy_kernel(const float* __restrict__ dy, int nb, int OH, int OW, int N, int ld_dy, int TH, int TW,
                                                      const float* __restrict__ gscale, float* __restrict__ dZ) {
  const int n4 = N >> 2;
  const long long T = (long long)nb * TH * TW, total = T * n4;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total; i += gridDim.x * blockDim.x) {      // total < 2^31 (launcher)
    const unsigned tile = i / (unsigned)n4;                     // 32-bit index arithmetic: 64-bit divisions cost microseconds here
    const int nq = (int)(i - tile * (unsigned)n4);
    const unsigned trow = tile / (unsigned)TW;
    const int tj = (int)(tile - trow * (unsigned)TW);
    const int img = (int)(trow / (unsigned)TH);
    const int ti = (int)(trow - (unsigned)img * (unsigned)TH);
    const float4 gs = gscale ? *reinterpret_cast<const float4*>(gscale + nq * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
    float4 d[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int oh = 2 * ti + a, ow = 2 * tj + b;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (oh < OH && ow < OW) v = *reinterpret_cast<const float4*>(dy + (((long long)img * OH + oh) * OW + ow) * ld_dy + nq * 4);
        d[a][b] = make_float4(v.x * gs.x, v.y * gs.y, v.z * gs.z, v.w * gs.w);
      }
    // r = A d (4x2): d0, d0+d1, d0-d1, -d1
    float4 r[4][2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      r[0][b] = d[0][b];
      r[1][b] = f4add(d[0][b], d[1][b]);
      r[2][b] = f4sub(d[0][b], d[1][b]);
      r[3][b] = f4sub(make_float4(0.f, 0.f, 0.f, 0.f), d[1][b]);
    }
    const long long ps = T * n4;
    float4* dst = reinterpret_cast<float4*>(dZ) + tile * n4 + nq;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      dst[(4 * a + 0) * ps] = r[a][0];
      dst[(4 * a + 1) * ps] = f4add(r[a][0], r[a][1]);
      dst[(4 * a + 2) * ps] = f4sub(r[a][0], r[a][1]);
      dst[(4 * a + 3) * ps] = f4sub(make_float4(0.f, 0.f, 0.f, 0.f), r[a][1]);
    }
  }
}

// dw[3][3][C][ldw] (+)= G^T dU G, dU: [16][C][N]
__global__ void __launch_bounds__(256) wino_filter_grad_kernel(const float* __restrict__ dU, int C, int N, int ldw, float* __restrict__ dw, int accumulate) {
  const int n4 = N >> 2;
  const long long total = (long long)C * n4;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total; i += gridDim.x * blockDim.x) {      // total < 2^31 (launcher)
    const int c = (int)(i / (unsigned)n4), nq = (int)(i - (unsigned)c * (unsigned)n4);
    const long long ps = (long long)C * n4;
    const float4* src = reinterpret_cast<const float4*>(dU) + (long long)c * n4 + nq;
    float4 u[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) u[a][b] = src[(4 * a + b) * ps];
    // t = G^T u (3x4): u0 + (u1+u2)/2, (u1-u2)/2, (u1+u2)/2 + u3
    float4 t[3][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const float4 hs = f4half(f4add(u[1][b], u[2][b])), hd = f4half(f4sub(u[1][b], u[2][b]));
      t[0][b] = f4add(u[0][b], hs);
      t[1][b] = hd;
      t[2][b] = f4add(hs, u[3][b]);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float4 hs = f4half(f4add(t[a][1], t[a][2])), hd = f4half(f4sub(t[a][1], t[a][2]));
      const float4 g3[3] = {f4add(t[a][0], hs), hd, f4add(hs, t[a][3])};
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        float4* p = reinterpret_cast<float4*>(dw + ((long long)(a * 3 + b) * C + c) * ldw + nq * 4);
        *p = accumulate ? f4add(*p, g3[b]) : g3[b];
      }
    }
  }
}

// ---- F(4x4, 3x3): 6x6 input patches, 36 products per 4x4 output tile and channel pair instead of 144 (4x fewer matrix-core
// flops than the direct form, 1.78x fewer than F(2x2)); interpolation points 0, +-1, +-2, inf (Lavin & Gray):
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
// The transforms multiply by up to 8 and by 1/24: against the fp64 direct sum the fp32 result is off by about 2e-5 of the
// largest activation at C = 1024 (F(2x2): 1e-6, direct fp32: 5e-7) -- inside the 2e-4 the kernel tests state.
// Layouts as above with 36 positions p = 6*xi + nu and tiles of 4x4 outputs (input rows 4ti-1 .. 4ti+4).
// Kernels are templates over the per-thread channel vector (instantiated for float4, see RADNET_WINO4_LAUNCH).
// (vector helpers and the 1-D transforms bt6 / at6 / a6 / g6 / gt6: radnet_wino4.h, shared with the chain kernel of conv_mfma.hip)
template <typename VT>
__global__ void __launch_bounds__(256) wino4_filter_kernel(const float* __restrict__ g, int C, int N, int ldw, float* __restrict__ U) {
  constexpr int W = sizeof(VT) / 4;
  const int nv = N / W;
  const long long total = (long long)C * nv, ps = total;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total; i += gridDim.x * blockDim.x) {      // total < 2^31 (launcher)
    const int c = (int)(i / (unsigned)nv), nq = (int)(i - (unsigned)c * (unsigned)nv);
    VT t[6][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      VT col[3], o[6];
#pragma unroll
      for (int a = 0; a < 3; ++a) col[a] = *reinterpret_cast<const VT*>(g + ((long long)(a * 3 + b) * C + c) * ldw + nq * W);
      g6(col, o);
#pragma unroll
      for (int a = 0; a < 6; ++a) t[a][b] = o[a];
    }
    VT* dst = reinterpret_cast<VT*>(U) + (long long)c * nv + nq;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      VT o[6];
      g6(t[a], o);
#pragma unroll
      for (int b = 0; b < 6; ++b) dst[(6 * a + b) * ps] = o[b];
    }
  }
}

template <typename VT>
__global__ void __launch_bounds__(256) wino4_input_kernel(const float* __restrict__ x, int nb, int H, int W_, int C, int TH, int TW,
                                                          float* __restrict__ V) {
  constexpr int W = sizeof(VT) / 4;
  const int cv = C / W;
  const long long T = (long long)nb * TH * TW, total = T * cv, ps = total;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total; i += gridDim.x * blockDim.x) {      // total < 2^31 (launcher)
    const unsigned tile = i / (unsigned)cv;                     // 32-bit index arithmetic: 64-bit divisions cost microseconds here
    const int cq = (int)(i - tile * (unsigned)cv);
    const unsigned trow = tile / (unsigned)TW;
    const int tj = (int)(tile - trow * (unsigned)TW);
    const int img = (int)(trow / (unsigned)TH);
    const int ti = (int)(trow - (unsigned)img * (unsigned)TH);
    VT t[6][6];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      const int iw = 4 * tj - 1 + b;
      VT col[6], o[6];
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        const int ih = 4 * ti - 1 + a;
        col[a] = ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W_)
                     ? *reinterpret_cast<const VT*>(x + (((long long)img * H + ih) * W_ + iw) * C + cq * W)
                     : vzero<VT>();
      }
      bt6(col, o);
#pragma unroll
      for (int a = 0; a < 6; ++a) t[a][b] = o[a];
    }
    VT* dst = reinterpret_cast<VT*>(V) + tile * cv + cq;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      VT o[6];
      bt6(t[a], o);
#pragma unroll
      for (int b = 0; b < 6; ++b) dst[(6 * a + b) * ps] = o[b];
    }
  }
}

template <typename VT>
__global__ void __launch_bounds__(256) wino4_output_kernel(const float* __restrict__ Mm, int nb, int OH, int OW, int N, int TH, int TW,
                                                           const float* __restrict__ scale, const float* __restrict__ shift, int act,
                                                           float* __restrict__ y, int ldy) {
  constexpr int W = sizeof(VT) / 4;
  const int nv = N / W;
  const long long T = (long long)nb * TH * TW, total = T * nv, ps = total;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total; i += gridDim.x * blockDim.x) {      // total < 2^31 (launcher)
    const unsigned tile = i / (unsigned)nv;                     // 32-bit index arithmetic: 64-bit divisions cost microseconds here
    const int nq = (int)(i - tile * (unsigned)nv);
    const unsigned trow = tile / (unsigned)TW;
    const int tj = (int)(tile - trow * (unsigned)TW);
    const int img = (int)(trow / (unsigned)TH);
    const int ti = (int)(trow - (unsigned)img * (unsigned)TH);
    const VT* src = reinterpret_cast<const VT*>(Mm) + tile * nv + nq;
    VT t[4][6];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      VT col[6], o[4];
#pragma unroll
      for (int a = 0; a < 6; ++a) col[a] = src[(6 * a + b) * ps];
      at6(col, o);
#pragma unroll
      for (int a = 0; a < 4; ++a) t[a][b] = o[a];
    }
    VT sc, sh = vzero<VT>();
    if (scale) sc = *reinterpret_cast<const VT*>(scale + nq * W);
    if (shift) sh = *reinterpret_cast<const VT*>(shift + nq * W);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int oh = 4 * ti + a;
      VT o[4];
      at6(t[a], o);
      if (oh >= OH) continue;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int ow = 4 * tj + b;
        if (ow >= OW) continue;
        VT v = scale ? o[b] * sc + sh : o[b] + sh;
        if (act == 1) v = vmax0(v);
        *reinterpret_cast<VT*>(y + (((long long)img * OH + oh) * OW + ow) * ldy + nq * W) = v;
      }
    }
  }
}

template <typename VT>
__global__ void __launch_bounds__(256) wino4_dy_kernel(const float* __restrict__ dy, int nb, int OH, int OW, int N, int ld_dy, int TH, int TW,
                                                       const float* __restrict__ gscale, float* __restrict__ dZ) {
  constexpr int W = sizeof(VT) / 4;
  const int nv = N / W;
  const long long T = (long long)nb * TH * TW, total = T * nv, ps = total;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total; i += gridDim.x * blockDim.x) {      // total < 2^31 (launcher)
    const unsigned tile = i / (unsigned)nv;                     // 32-bit index arithmetic: 64-bit divisions cost microseconds here
    const int nq = (int)(i - tile * (unsigned)nv);
    const unsigned trow = tile / (unsigned)TW;
    const int tj = (int)(tile - trow * (unsigned)TW);
    const int img = (int)(trow / (unsigned)TH);
    const int ti = (int)(trow - (unsigned)img * (unsigned)TH);
    VT gs;
    if (gscale) gs = *reinterpret_cast<const VT*>(gscale + nq * W);
    VT r[6][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int ow = 4 * tj + b;
      VT col[4], o[6];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int oh = 4 * ti + a;
        VT v = vzero<VT>();
        if (oh < OH && ow < OW) v = *reinterpret_cast<const VT*>(dy + (((long long)img * OH + oh) * OW + ow) * ld_dy + nq * W);
        col[a] = gscale ? v * gs : v;
      }
      a6(col, o);
#pragma unroll
      for (int a = 0; a < 6; ++a) r[a][b] = o[a];
    }
    VT* dst = reinterpret_cast<VT*>(dZ) + tile * nv + nq;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      VT o[6];
      a6(r[a], o);
#pragma unroll
      for (int b = 0; b < 6; ++b) dst[(6 * a + b) * ps] = o[b];
    }
  }
}

template <typename VT>
__global__ void __launch_bounds__(256) wino4_filter_grad_kernel(const float* __restrict__ dU, int C, int N, int ldw, float* __restrict__ dw, int accumulate) {
  constexpr int W = sizeof(VT) / 4;
  const int nv = N / W;
  const long long total = (long long)C * nv, ps = total;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total; i += gridDim.x * blockDim.x) {      // total < 2^31 (launcher)
    const int c = (int)(i / (unsigned)nv), nq = (int)(i - (unsigned)c * (unsigned)nv);
    const VT* src = reinterpret_cast<const VT*>(dU) + (long long)c * nv + nq;
    VT t[3][6];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      VT col[6], o[3];
#pragma unroll
      for (int a = 0; a < 6; ++a) col[a] = src[(6 * a + b) * ps];
      gt6(col, o);
#pragma unroll
      for (int a = 0; a < 3; ++a) t[a][b] = o[a];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      VT o[3];
      gt6(t[a], o);
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        VT* q = reinterpret_cast<VT*>(dw + ((long long)(a * 3 + b) * C + c) * ldw + nq * W);
        *q = accumulate ? *q + o[b] : o[b];
      }
    }
  }
}

// ---- round 4: the two per-layer transforms with SIX threads per (tile, 4 channels) ---------------------------------------------
// The one-thread-per-unit kernels above give a 160-tile stage-4 layer 40 workgroups, each thread walking 36 loads and 36 stores
// on its own.  Here a unit's six patch columns (input) / six position columns (output) go to six WAVES of a 384-thread workgroup
// (role = wave, unit = lane: every access of a wave is 64 consecutive channel quads = 1 KB): role r transforms column r along the
// rows (6 loads), the 6x6 (6x4) intermediate crosses through LDS, role r then transforms ROW r along the columns and stores it --
// 6 + 6 memory operations per thread, 4 times the workgroups.  Same operations in the same order as the kernels above: same bits.
__global__ void __launch_bounds__(384) wino4_input_v2_kernel(const float* __restrict__ x, int nb, int H, int W_, int C, int TH, int TW,
                                                             float* __restrict__ V) {
  __shared__ float4 lds[36 * 64];
  const int cv = C >> 2;
  const unsigned T = (unsigned)(nb * TH * TW), total = T * (unsigned)cv;
  const int lane = threadIdx.x & 63, role = threadIdx.x >> 6;
  const unsigned i = blockIdx.x * 64u + (unsigned)lane;
  const bool live = i < total;
  const unsigned ii = live ? i : 0u;
  const unsigned tile = ii / (unsigned)cv;
  const int cq = (int)(ii - tile * (unsigned)cv);
  const unsigned trow = tile / (unsigned)TW;
  const int tj = (int)(tile - trow * (unsigned)TW);
  const int img = (int)(trow / (unsigned)TH);
  const int ti = (int)(trow - (unsigned)img * (unsigned)TH);
  {
    const int iw = 4 * tj - 1 + role;
    float4 col[6], o[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      const int ih = 4 * ti - 1 + a;
      col[a] = (live && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W_)
                   ? *reinterpret_cast<const float4*>(x + (((long long)img * H + ih) * W_ + iw) * C + cq * 4)
                   : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    bt6(col, o);
#pragma unroll
    for (int a = 0; a < 6; ++a) lds[(a * 6 + role) * 64 + lane] = o[a];
  }
  __syncthreads();
  {
    float4 row[6], o[6];
#pragma unroll
    for (int b = 0; b < 6; ++b) row[b] = lds[(role * 6 + b) * 64 + lane];
    bt6(row, o);
    if (live) {
      float4* dst = reinterpret_cast<float4*>(V) + (size_t)tile * cv + cq;
      const size_t ps = total;
#pragma unroll
      for (int b = 0; b < 6; ++b) dst[(size_t)(6 * role + b) * ps] = o[b];
    }
  }
}

__global__ void __launch_bounds__(384) wino4_output_v2_kernel(const float* __restrict__ Mm, int nb, int OH, int OW, int N, int TH, int TW,
                                                              const float* __restrict__ scale, const float* __restrict__ shift, int act,
                                                              float* __restrict__ y, int ldy) {
  __shared__ float4 lds[24 * 64];
  const int nv = N >> 2;
  const unsigned T = (unsigned)(nb * TH * TW), total = T * (unsigned)nv;
  const int lane = threadIdx.x & 63, role = threadIdx.x >> 6;
  const unsigned i = blockIdx.x * 64u + (unsigned)lane;
  const bool live = i < total;
  const unsigned ii = live ? i : 0u;
  const unsigned tile = ii / (unsigned)nv;
  const int nq = (int)(ii - tile * (unsigned)nv);
  {
    const float4* src = reinterpret_cast<const float4*>(Mm) + (size_t)tile * nv + nq;
    const size_t ps = total;
    float4 col[6], o[4];
#pragma unroll
    for (int a = 0; a < 6; ++a) col[a] = live ? src[(size_t)(6 * a + role) * ps] : make_float4(0.f, 0.f, 0.f, 0.f);
    at6(col, o);
#pragma unroll
    for (int a = 0; a < 4; ++a) lds[(a * 6 + role) * 64 + lane] = o[a];
  }
  __syncthreads();
  if (role < 4 && live) {
    const unsigned trow = tile / (unsigned)TW;
    const int tj = (int)(tile - trow * (unsigned)TW);
    const int img = (int)(trow / (unsigned)TH);
    const int ti = (int)(trow - (unsigned)img * (unsigned)TH);
    float4 row[6], o[4];
#pragma unroll
    for (int b = 0; b < 6; ++b) row[b] = lds[(role * 6 + b) * 64 + lane];
    at6(row, o);
    const int oh = 4 * ti + role;
    if (oh < OH) {
      float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
      if (scale) sc = *reinterpret_cast<const float4*>(scale + nq * 4);
      if (shift) sh = *reinterpret_cast<const float4*>(shift + nq * 4);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int ow = 4 * tj + b;
        if (ow >= OW) continue;
        float4 v = scale ? o[b] * sc + sh : o[b] + sh;
        if (act == 1) v = vmax0(v);
        *reinterpret_cast<float4*>(y + (((long long)img * OH + oh) * OW + ow) * ldy + nq * 4) = v;
      }
    }
  }
}

// One thread per (tile | filter element, 4 channels).  Narrower per-thread vectors (float2 / float: 2x / 4x the workgroups for
// the 160-tile stage-4 layers) were measured and changed nothing -- input transform 6.3 us, output 5.9 us either way: at a few
// MB per launch these kernels last one launch + one memory round trip, not a bandwidth- or occupancy-limited time.
#define RADNET_WINO4_LAUNCH(kernel, units4, ...) \
  hipLaunchKernelGGL((kernel<float4>), dim3(grid_of(units4)), dim3(256), 0, ctx->stream, __VA_ARGS__)

inline int grid_of(long long total) {
  long long b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

}  // namespace

extern "C" int radnet_winograd_filter(radnet_ctx* ctx, const float* w, int32_t c, int32_t n, int32_t ldw, float* u) {
  if (!ctx || !w || !u) return RADNET_ERR_ARG;
  if ((n & 3) || (ldw & 3)) RADNET_FAIL(ctx, RADNET_ERR_ARG, "winograd_filter: n=%d, ldw=%d must be multiples of 4", n, ldw);
  if ((long long)c * (n / 4) >= (1ll << 31)) RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "wino_filter_kernel: %lld work items (32-bit index arithmetic)", (long long)((long long)c * (n / 4)));
  hipLaunchKernelGGL(wino_filter_kernel, dim3(grid_of((long long)c * (n / 4))), dim3(256), 0, ctx->stream, w, c, n, ldw, u);
  RADNET_CHECK_LAUNCH(ctx, "winograd_filter");
  return RADNET_OK;
}

extern "C" int radnet_winograd_input(radnet_ctx* ctx, const float* x, int32_t nb, int32_t h, int32_t w, int32_t c, float* v) {
  if (!ctx || !x || !v) return RADNET_ERR_ARG;
  if (c & 3) RADNET_FAIL(ctx, RADNET_ERR_ARG, "winograd_input: c=%d must be a multiple of 4", c);
  const int th = (h + 1) / 2, tw = (w + 1) / 2;
  if ((long long)nb * th * tw * (c / 4) >= (1ll << 31)) RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "wino_input_kernel: %lld work items (32-bit index arithmetic)", (long long)((long long)nb * th * tw * (c / 4)));
  hipLaunchKernelGGL(wino_input_kernel, dim3(grid_of((long long)nb * th * tw * (c / 4))), dim3(256), 0, ctx->stream, x, nb, h, w, c, th, tw, v);
  RADNET_CHECK_LAUNCH(ctx, "winograd_input");
  return RADNET_OK;
}

extern "C" int radnet_winograd_output(radnet_ctx* ctx, const float* m, int32_t nb, int32_t oh, int32_t ow, int32_t n, const float* scale,
                                      const float* shift, int32_t act, float* y, int32_t ldy) {
  if (!ctx || !m || !y) return RADNET_ERR_ARG;
  if ((n & 3) || (ldy & 3) || ldy < n) RADNET_FAIL(ctx, RADNET_ERR_ARG, "winograd_output: n=%d, ldy=%d", n, ldy);
  const int th = (oh + 1) / 2, tw = (ow + 1) / 2;
  if ((long long)nb * th * tw * (n / 4) >= (1ll << 31)) RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "wino_output_kernel: %lld work items (32-bit index arithmetic)", (long long)((long long)nb * th * tw * (n / 4)));
  hipLaunchKernelGGL(wino_output_kernel, dim3(grid_of((long long)nb * th * tw * (n / 4))), dim3(256), 0, ctx->stream, m, nb, oh, ow, n, th, tw, scale,
                     shift, act, y, ldy);
  RADNET_CHECK_LAUNCH(ctx, "winograd_output");
  return RADNET_OK;
}

extern "C" int radnet_winograd_dy(radnet_ctx* ctx, const float* dy, int32_t nb, int32_t oh, int32_t ow, int32_t n, int32_t ld_dy,
                                  const float* gscale, float* dz) {
  if (!ctx || !dy || !dz) return RADNET_ERR_ARG;
  if ((n & 3) || (ld_dy & 3) || ld_dy < n) RADNET_FAIL(ctx, RADNET_ERR_ARG, "winograd_dy: n=%d, ld_dy=%d", n, ld_dy);
  const int th = (oh + 1) / 2, tw = (ow + 1) / 2;
  if ((long long)nb * th * tw * (n / 4) >= (1ll << 31)) RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "wino_dy_kernel: %lld work items (32-bit index arithmetic)", (long long)((long long)nb * th * tw * (n / 4)));
  hipLaunchKernelGGL(wino_dy_kernel, dim3(grid_of((long long)nb * th * tw * (n / 4))), dim3(256), 0, ctx->stream, dy, nb, oh, ow, n, ld_dy, th, tw,
                     gscale, dz);
  RADNET_CHECK_LAUNCH(ctx, "winograd_dy");
  return RADNET_OK;
}

extern "C" int radnet_winograd_filter_grad(radnet_ctx* ctx, const float* du, int32_t c, int32_t n, int32_t ldw, float* dw, int32_t accumulate) {
  if (!ctx || !du || !dw) return RADNET_ERR_ARG;
  if ((n & 3) || (ldw & 3)) RADNET_FAIL(ctx, RADNET_ERR_ARG, "winograd_filter_grad: n=%d, ldw=%d must be multiples of 4", n, ldw);
  if ((long long)c * (n / 4) >= (1ll << 31)) RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "wino_filter_grad_kernel: %lld work items (32-bit index arithmetic)", (long long)((long long)c * (n / 4)));
  hipLaunchKernelGGL(wino_filter_grad_kernel, dim3(grid_of((long long)c * (n / 4))), dim3(256), 0, ctx->stream, du, c, n, ldw, dw, accumulate ? 1 : 0);
  RADNET_CHECK_LAUNCH(ctx, "winograd_filter_grad");
  return RADNET_OK;
}


// ---- F(4x4,3x3) entry points: same contracts with 36 positions and tiles = nb*ceil(h/4)*ceil(w/4) -----------------------
extern "C" int radnet_winograd4_filter(radnet_ctx* ctx, const float* w, int32_t c, int32_t n, int32_t ldw, float* u) {
  if (!ctx || !w || !u) return RADNET_ERR_ARG;
  if ((n & 3) || (ldw & 3)) RADNET_FAIL(ctx, RADNET_ERR_ARG, "winograd4_filter: n=%d, ldw=%d must be multiples of 4", n, ldw);
  if ((long long)c * (n / 4) >= (1ll << 31)) RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "wino4_filter_kernel: %lld work items (32-bit index arithmetic)", (long long)((long long)c * (n / 4)));
  RADNET_WINO4_LAUNCH(wino4_filter_kernel, (long long)c * (n / 4), w, c, n, ldw, u);
  RADNET_CHECK_LAUNCH(ctx, "winograd4_filter");
  return RADNET_OK;
}

extern "C" int radnet_winograd4_input(radnet_ctx* ctx, const float* x, int32_t nb, int32_t h, int32_t w, int32_t c, float* v) {
  if (!ctx || !x || !v) return RADNET_ERR_ARG;
  if (c & 3) RADNET_FAIL(ctx, RADNET_ERR_ARG, "winograd4_input: c=%d must be a multiple of 4", c);
  const int th = (h + 3) / 4, tw = (w + 3) / 4;
  if ((long long)nb * th * tw * (c / 4) >= (1ll << 31)) RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "wino4_input_kernel: %lld work items (32-bit index arithmetic)", (long long)((long long)nb * th * tw * (c / 4)));
  static const bool v1 = getenv("RADNET_WINO_V1") != nullptr;      // A/B: the one-thread-per-unit kernels
  const long long units = (long long)nb * th * tw * (c / 4);
  if (v1) RADNET_WINO4_LAUNCH(wino4_input_kernel, units, x, nb, h, w, c, th, tw, v);
  else hipLaunchKernelGGL(wino4_input_v2_kernel, dim3((unsigned)((units + 63) / 64)), dim3(384), 0, ctx->stream, x, nb, h, w, c, th, tw, v);
  RADNET_CHECK_LAUNCH(ctx, "winograd4_input");
  return RADNET_OK;
}

extern "C" int radnet_winograd4_output(radnet_ctx* ctx, const float* m, int32_t nb, int32_t oh, int32_t ow, int32_t n, const float* scale,
                                       const float* shift, int32_t act, float* y, int32_t ldy) {
  if (!ctx || !m || !y) return RADNET_ERR_ARG;
  if ((n & 3) || (ldy & 3) || ldy < n) RADNET_FAIL(ctx, RADNET_ERR_ARG, "winograd4_output: n=%d, ldy=%d", n, ldy);
  const int th = (oh + 3) / 4, tw = (ow + 3) / 4;
  if ((long long)nb * th * tw * (n / 4) >= (1ll << 31)) RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "wino4_output_kernel: %lld work items (32-bit index arithmetic)", (long long)((long long)nb * th * tw * (n / 4)));
  static const bool v1 = getenv("RADNET_WINO_V1") != nullptr;
  const long long units = (long long)nb * th * tw * (n / 4);
  if (v1) RADNET_WINO4_LAUNCH(wino4_output_kernel, units, m, nb, oh, ow, n, th, tw, scale, shift, act, y, ldy);
  else hipLaunchKernelGGL(wino4_output_v2_kernel, dim3((unsigned)((units + 63) / 64)), dim3(384), 0, ctx->stream, m, nb, oh, ow, n, th, tw, scale, shift, act, y, ldy);
  RADNET_CHECK_LAUNCH(ctx, "winograd4_output");
  return RADNET_OK;
}

extern "C" int radnet_winograd4_dy(radnet_ctx* ctx, const float* dy, int32_t nb, int32_t oh, int32_t ow, int32_t n, int32_t ld_dy,
                                   const float* gscale, float* dz) {
  if (!ctx || !dy || !dz) return RADNET_ERR_ARG;
  if ((n & 3) || (ld_dy & 3) || ld_dy < n) RADNET_FAIL(ctx, RADNET_ERR_ARG, "winograd4_dy: n=%d, ld_dy=%d", n, ld_dy);
  const int th = (oh + 3) / 4, tw = (ow + 3) / 4;
  if ((long long)nb * th * tw * (n / 4) >= (1ll << 31)) RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "wino4_dy_kernel: %lld work items (32-bit index arithmetic)", (long long)((long long)nb * th * tw * (n / 4)));
  RADNET_WINO4_LAUNCH(wino4_dy_kernel, (long long)nb * th * tw * (n / 4), dy, nb, oh, ow, n, ld_dy, th, tw, gscale, dz);
  RADNET_CHECK_LAUNCH(ctx, "winograd4_dy");
  return RADNET_OK;
}

extern "C" int radnet_winograd4_filter_grad(radnet_ctx* ctx, const float* du, int32_t c, int32_t n, int32_t ldw, float* dw, int32_t accumulate) {
  if (!ctx || !du || !dw) return RADNET_ERR_ARG;
  if ((n & 3) || (ldw & 3)) RADNET_FAIL(ctx, RADNET_ERR_ARG, "winograd4_filter_grad: n=%d, ldw=%d must be multiples of 4", n, ldw);
  if ((long long)c * (n / 4) >= (1ll << 31)) RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "wino4_filter_grad_kernel: %lld work items (32-bit index arithmetic)", (long long)((long long)c * (n / 4)));
  RADNET_WINO4_LAUNCH(wino4_filter_grad_kernel, (long long)c * (n / 4), du, c, n, ldw, dw, accumulate ? 1 : 0);
  RADNET_CHECK_LAUNCH(ctx, "winograd4_filter_grad");
  return RADNET_OK;
}
