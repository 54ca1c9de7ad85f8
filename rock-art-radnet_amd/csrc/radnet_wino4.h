// F(4x4, 3x3) Winograd building blocks shared by winograd.hip (one launch per transform) and the chain kernel of
// conv_mfma.hip (transforms as work items of a persistent launch).  See winograd.hip for the matrices and the layouts.
#pragma once
#include <hip/hip_runtime.h>

namespace {

template <typename VT> __device__ __forceinline__ VT vzero();
template <> __device__ __forceinline__ float vzero<float>() { return 0.f; }
template <> __device__ __forceinline__ float2 vzero<float2>() { return make_float2(0.f, 0.f); }
template <> __device__ __forceinline__ float4 vzero<float4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float vmax0(float a) { return fmaxf(a, 0.f); }
__device__ __forceinline__ float2 vmax0(float2 a) { return make_float2(fmaxf(a.x, 0.f), fmaxf(a.y, 0.f)); }
__device__ __forceinline__ float4 vmax0(float4 a) { return make_float4(fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(a.z, 0.f), fmaxf(a.w, 0.f)); }

template <typename VT> __device__ __forceinline__ void bt6(const VT (&d)[6], VT (&o)[6]) {          // o = B^T d
  o[0] = d[0] * 4.f + (d[4] - d[2] * 5.f);
  const VT p = d[4] - d[2] * 4.f, q = d[3] - d[1] * 4.f;
  o[1] = p + q;
  o[2] = p - q;
  const VT r = d[4] - d[2], t = (d[3] - d[1]) * 2.f;
  o[3] = r + t;
  o[4] = r - t;
  o[5] = d[1] * 4.f + (d[5] - d[3] * 5.f);
}
template <typename VT> __device__ __forceinline__ void at6(const VT (&m)[6], VT (&o)[4]) {          // o = A^T m
  const VT s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
  o[0] = (m[0] + s12) + s34;
  o[1] = d34 * 2.f + d12;
  o[2] = s34 * 4.f + s12;
  o[3] = (d34 * 8.f + d12) + m[5];
}
template <typename VT> __device__ __forceinline__ void a6(const VT (&y)[4], VT (&z)[6]) {           // z = A y (adjoint of at6)
  const VT e = y[0] + y[2], o = y[1] + y[3];
  const VT e4 = y[2] * 4.f + y[0], o2 = y[3] * 8.f + y[1] * 2.f;
  z[0] = y[0];
  z[1] = e + o;
  z[2] = e - o;
  z[3] = e4 + o2;
  z[4] = e4 - o2;
  z[5] = y[3];
}
template <typename VT> __device__ __forceinline__ void g6(const VT (&g)[3], VT (&u)[6]) {           // u = G g
  const VT s = g[0] + g[2];
  u[0] = g[0] * 0.25f;
  u[1] = (s + g[1]) * (-1.f / 6.f);
  u[2] = (s - g[1]) * (-1.f / 6.f);
  const VT a = g[0] * (1.f / 24.f) + g[2] * (1.f / 6.f), b = g[1] * (1.f / 12.f);
  u[3] = a + b;
  u[4] = a - b;
  u[5] = g[2];
}
template <typename VT> __device__ __forceinline__ void gt6(const VT (&u)[6], VT (&w)[3]) {          // w = G^T u (adjoint of g6)
  const VT s12 = u[1] + u[2], s34 = u[3] + u[4];
  w[0] = u[0] * 0.25f + (s12 * (-1.f / 6.f) + s34 * (1.f / 24.f));
  w[1] = (u[2] - u[1]) * (1.f / 6.f) + (u[3] - u[4]) * (1.f / 12.f);
  w[2] = (s12 * (-1.f / 6.f) + s34 * (1.f / 6.f)) + u[5];
}


}  // namespace
