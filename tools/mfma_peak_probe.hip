// Micro-probe: wall-clock fp32 MFMA throughput (v_mfma_f32_32x32x2_f32) the whole chip SUSTAINS, operands in registers,
// for 1 / 2 / 4 waves per SIMD and short / long kernels -- the practical ceiling (clock under load) next to the 157.3
// TFLOP/s datasheet peak (256 CUs x 4 SIMDs x 256 flop/cycle/CU-SIMD... at 2.4 GHz).
// hipcc --offload-arch=gfx950 -O3 tools/mfma_peak_probe.hip -o tools/bin/mfma_peak_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(256) peak(const float* __restrict__ in, float* __restrict__ out, int iters) {
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  const float a = in[threadIdx.x], b = in[threadIdx.x + 256];
  for (int t = 0; t < iters; ++t) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, a3, 0, 0, 0);
    }
  }
  float r = 0.f;
  for (int i = 0; i < 16; ++i) r += a0[i] + a1[i] + a2[i] + a3[i];
  if (r == 123.456f) out[threadIdx.x] = r;
}

int main() {
  float *in, *out;
  (void)hipMalloc(&in, 4096); (void)hipMalloc(&out, 4096);
  {  // non-trivial operands: zeros would understate the power drawn by the multipliers
    float h[1024];
    unsigned s = 12345u;
    for (int i = 0; i < 1024; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((int)(s >> 8) % 2001 - 1000) * 1e-6f; }
    (void)hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wps : {1, 2, 4}) {
    for (int iters : {200, 2000, 20000}) {
      const int wgs = 256 * wps;      // 256-thread workgroup = one wave on each SIMD of a CU
      hipLaunchKernelGGL(peak, dim3(wgs), dim3(256), 0, 0, in, out, iters);   // warm
      hipDeviceSynchronize();
      hipEventRecord(e0);
      const int reps = 5;
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(peak, dim3(wgs), dim3(256), 0, 0, in, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flops = (double)reps * wgs * 4 /*waves*/ * iters * 32.0 /*mfma*/ * (2.0 * 32 * 32 * 2);
      printf("waves/SIMD %d, %6d iters: %.3f ms per launch, %.1f TFLOP/s\n", wps, iters, ms / reps, flops / (ms * 1e9));
    }
  }
  return 0;
}
