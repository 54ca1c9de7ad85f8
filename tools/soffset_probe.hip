// Micro-probe: is the SGPR offset of a raw buffer load part of the hardware range check on gfx950?
// hipcc --offload-arch=gfx950 -O3 tools/soffset_probe.hip -o tools/bin/soffset_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float* buf, unsigned bytes, const unsigned* voffs, const unsigned* soffs, int n, float* out) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(buf), 0, (int)bytes, 0x00020000);
  for (int c = 0; c < n; ++c) {
    const unsigned so = __builtin_amdgcn_readfirstlane(soffs[c]);
    f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voffs[c], (int)so, 0));
    if (threadIdx.x == 0) { out[4 * c] = v.x; out[4 * c + 1] = v.y; out[4 * c + 2] = v.z; out[4 * c + 3] = v.w; }
  }
}
int main() {
  const int nfl = 2048;              // allocation: 8192 bytes; the descriptor covers the first 4096
  float h[nfl];
  for (int i = 0; i < nfl; ++i) h[i] = (float)i;
  float *d, *out; unsigned *dv, *ds;
  const unsigned vo[] = {16, 0x80000000u, 16, 4080, 4080, 0, 4096, 2048, 16, 0, 0x80000010u, 0xFFFFFF00u, 0x80000000u};
  const unsigned so[] = {32, 32, 0x80000000u, 0, 16, 4096, 0, 2048, 0x7fffff00u, 4080, 0x80000000u, 0x200u, 0x80000000u};
  const int n = sizeof(vo) / 4;
  (void)hipMalloc(&d, sizeof(h)); (void)hipMalloc(&out, n * 16); (void)hipMalloc(&dv, sizeof(vo)); (void)hipMalloc(&ds, sizeof(so));
  (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice); (void)hipMemcpy(dv, vo, sizeof(vo), hipMemcpyHostToDevice); (void)hipMemcpy(ds, so, sizeof(so), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 4096u, dv, ds, n, out);
  float r[128];
  (void)hipMemcpy(r, out, n * 16, hipMemcpyDeviceToHost);
  for (int c = 0; c < n; ++c)
    printf("voffset %10u soffset %10u -> %6.0f %6.0f %6.0f %6.0f   (in-range element would be %u)\n", vo[c], so[c], r[4 * c], r[4 * c + 1], r[4 * c + 2], r[4 * c + 3],
           (vo[c] + so[c]) / 4);
  return 0;
}
