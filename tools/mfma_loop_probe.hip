// Micro-probe: what does one wave per SIMD sustain on v_mfma_f32_32x32x2_f32 under the ingredients of the conv K loop?
// Standalone (hipcc --offload-arch=gfx950 -O3 tools/mfma_loop_probe.hip -o gpurun_out/mfma_probe); not part of the library.
// Each variant runs 256 threads per workgroup, `wgs_per_cu` x 256 workgroups, and reports cycles per MFMA (ideal 64)
// from s_memtime around the loop (median over workgroups).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTiles = 64;   // 16 MFMA steps each

template <int VARIANT>
__global__ void __launch_bounds__(256) probe(const float* __restrict__ in, float* __restrict__ out, unsigned long long* cyc) {
  __shared__ float lds[2 * 32 * 133];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 2 * 32 * 133; i += 256) lds[i] = in[i & 4095];
  __syncthreads();
  f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
  float a = in[tid], b = in[tid + 256];
  unsigned junk = tid;
  float fx[4] = {a, b, a + 1.f, b + 1.f};
  float4 ld[8] = {};
  const float* sA = lds + (lane >> 5) * 65 + (lane & 31);
  const float* sB = lds + 32 * 65 + (lane >> 5) * 68 + (lane & 31);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int t = 0; t < kTiles; ++t) {
    if (VARIANT == 0) {          // one dependent chain, operands in registers
#pragma unroll
      for (int s = 0; s < 16; ++s) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
    } else if (VARIANT == 1) {   // two independent chains
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
      }
    } else if (VARIANT == 2) {   // four independent chains
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, acc3, 0, 0, 0);
      }
    } else if (VARIANT == 3 || VARIANT == 4 || VARIANT == 5) {
      // operands from LDS (ds_read_b32 x2 per MFMA, prefetched one step ahead), one chain (3) / two chains (4, 5);
      // 5 adds the barrier per tile
      float fa[2], fb[2];
      fa[0] = sA[0];
      fb[0] = sB[0];
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        if (s + 1 < 16) {
          fa[(s + 1) & 1] = sA[(2 * s + 2) * 65];
          fb[(s + 1) & 1] = sB[(2 * s + 2) * 68];
        }
        if (VARIANT == 3 || (s & 1) == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s & 1], fb[s & 1], acc0, 0, 0, 0);
        else acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s & 1], fb[s & 1], acc1, 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
      if (VARIANT == 5) __syncthreads();
    } else if (VARIANT == 6) {   // four chains from LDS (the 128x128 tile's shape: 4 reads, 4 MFMAs per step), barrier
      float fa[2][2], fb[2][2];
      fa[0][0] = sA[0]; fa[0][1] = sA[32]; fb[0][0] = sB[0]; fb[0][1] = sB[32];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (s + 1 < 4) {
          fa[(s + 1) & 1][0] = sA[(2 * s + 2) * 65]; fa[(s + 1) & 1][1] = sA[(2 * s + 2) * 65 + 32];
          fb[(s + 1) & 1][0] = sB[(2 * s + 2) * 68]; fb[(s + 1) & 1][1] = sB[(2 * s + 2) * 68 + 32];
        }
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s & 1][0], fb[s & 1][0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s & 1][0], fb[s & 1][1], acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s & 1][1], fb[s & 1][0], acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s & 1][1], fb[s & 1][1], acc3, 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      }
      __syncthreads();
    } else if (VARIANT >= 10 && VARIANT < 40) {
      // LDS operands, 2 chains, barrier, plus (VARIANT - 10) independent integer VALU ops after every MFMA
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const float fa = sA[(2 * s) * 65], fb = sB[(2 * s) * 68];
        if ((s & 1) == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc0, 0, 0, 0);
        else acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc1, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < VARIANT - 10; ++v) junk = junk * 1664525u + (unsigned)(s + v + t);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    } else if (VARIANT >= 100 && VARIANT < 200) {
      // conv loop shape plus (VARIANT - 100) independent v_fma_f32 per MFMA step (4 separate dependency chains)
      float fa[3], fb[3];
      fa[0] = sA[0]; fb[0] = sB[0]; fa[1] = sA[2 * 65]; fb[1] = sB[2 * 68];
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        if (s + 2 < 16) {
          fa[(s + 2) % 3] = sA[(2 * s + 4) * 65];
          fb[(s + 2) % 3] = sB[(2 * s + 4) * 68];
        }
        if ((s & 1) == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s % 3], fb[s % 3], acc0, 0, 0, 0);
        else acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s % 3], fb[s % 3], acc1, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < VARIANT - 100; ++v) fx[v & 3] = __builtin_fmaf(fx[v & 3], a, b);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    } else if (VARIANT >= 50 && VARIANT < 80) {
      // the conv K loop's shape: one chain pair, fragments prefetched two steps ahead, barrier per tile, and per tile
      //   50+W: W LDS stores (ds_write_b32 pairs, transposed-store pattern) from registers, one per step from step 4
      //   60+W: the same with ds_write_b128
      //   70+L: L 16-byte buffer-style global loads (results only summed after the loop), one per step from step 0
      float fa[3], fb[3];
      fa[0] = sA[0]; fb[0] = sB[0]; fa[1] = sA[2 * 65]; fb[1] = sB[2 * 68];
      float* wbase = lds + 32 * 133 + (t & 1) * 0 + (tid >> 3) + (tid & 7) * 4 * 65;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        if (s + 2 < 16) {
          fa[(s + 2) % 3] = sA[(2 * s + 4) * 65];
          fb[(s + 2) % 3] = sB[(2 * s + 4) * 68];
        }
        if ((s & 1) == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s % 3], fb[s % 3], acc0, 0, 0, 0);
        else acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s % 3], fb[s % 3], acc1, 0, 0, 0);
        if (VARIANT >= 50 && VARIANT < 60 && s >= 4 && s < 4 + (VARIANT - 50)) {
          wbase[(s & 3) * 65] = a;
          wbase[(s & 3) * 65 + 32] = b;
        }
        if (VARIANT >= 60 && VARIANT < 70 && s >= 4 && s < 4 + (VARIANT - 60))
          reinterpret_cast<float4*>(lds + 32 * 133)[tid + 256 * (s & 3)] = make_float4(a, b, a, b);
        if (VARIANT >= 70 && s < VARIANT - 70) {
          const float4 nv = reinterpret_cast<const float4*>(in)[(tid + 256 * s + 64 * t) & 1023];
          ld[s].x += nv.x;
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    } else if (VARIANT >= 40 && VARIANT < 50) {
      // same with (VARIANT - 40) 16-byte global loads per tile issued after the first MFMAs, consumed next tile,
      // and as many ds_write_b128 of the previous tile's values into the other half of the LDS array
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const float fa = sA[(2 * s) * 65], fb = sB[(2 * s) * 68];
        if ((s & 1) == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc0, 0, 0, 0);
        else acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc1, 0, 0, 0);
        if (s < VARIANT - 40) {
          const float4 nv = reinterpret_cast<const float4*>(in)[(tid + 256 * s + 64 * t) & 1023];
          reinterpret_cast<float4*>(lds + 32 * 133)[tid + 256 * (s & 3)] = ld[s];
          ld[s] = nv;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  (void)0;
  f32x16 r = acc0 + acc1 + acc2 + acc3;
  float sum = 0.f;
  for (int i = 0; i < 16; ++i) sum += r[i];
  for (int i = 0; i < 8; ++i) sum += ld[i].x;
  out[blockIdx.x * 256 + tid] = sum + (float)junk + fx[0] + fx[1] + fx[2] + fx[3];
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V>
void run(const char* name, int wgs_per_cu, const float* in, float* out, unsigned long long* cyc) {
  const int n = 256 * wgs_per_cu;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(probe<V>, dim3(n), dim3(256), 0, 0, in, out, cyc);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(n);
  hipMemcpy(h.data(), cyc, n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  printf("%-58s wgs/CU %d : %7.1f cycles per MFMA (median), %7.1f (p90)\n", name, wgs_per_cu, (double)h[n / 2] / (kTiles * 16),
         (double)h[n * 9 / 10] / (kTiles * 16));
}

int main() {
  float *in, *out;
  unsigned long long* cyc;
  hipMalloc(&in, 8192 * sizeof(float));
  hipMalloc(&out, 1024 * 256 * sizeof(float));
  hipMalloc(&cyc, 1024 * sizeof(unsigned long long));
  std::vector<float> h(8192);
  for (int i = 0; i < 8192; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
  hipMemcpy(in, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
  for (int w = 1; w <= 2; ++w) {
    run<0>("registers, 1 dependent chain", w, in, out, cyc);
    run<1>("registers, 2 chains", w, in, out, cyc);
    run<2>("registers, 4 chains", w, in, out, cyc);
    run<3>("LDS operands (2 ds_read_b32 / MFMA), 1 chain", w, in, out, cyc);
    run<4>("LDS operands, 2 chains", w, in, out, cyc);
    run<5>("LDS operands, 2 chains, barrier per 16", w, in, out, cyc);
    run<6>("LDS operands, 4 chains (2x2 frags), barrier per 16", w, in, out, cyc);
    run<10>("LDS, 2 chains, barrier, +0 VALU per MFMA", w, in, out, cyc);
    run<14>("LDS, 2 chains, barrier, +4 VALU per MFMA", w, in, out, cyc);
    run<18>("LDS, 2 chains, barrier, +8 VALU per MFMA", w, in, out, cyc);
    run<22>("LDS, 2 chains, barrier, +12 VALU per MFMA", w, in, out, cyc);
    run<26>("LDS, 2 chains, barrier, +16 VALU per MFMA", w, in, out, cyc);
    run<34>("LDS, 2 chains, barrier, +24 VALU per MFMA", w, in, out, cyc);
    run<50>("conv loop shape, no staging", w, in, out, cyc);
    run<104>("conv loop shape, +4 v_fma_f32 per MFMA", w, in, out, cyc);
    run<108>("conv loop shape, +8 v_fma_f32 per MFMA", w, in, out, cyc);
    run<112>("conv loop shape, +12 v_fma_f32 per MFMA", w, in, out, cyc);
    run<116>("conv loop shape, +16 v_fma_f32 per MFMA", w, in, out, cyc);
    run<124>("conv loop shape, +24 v_fma_f32 per MFMA", w, in, out, cyc);
    run<52>("conv loop shape, 2 ds_write2_b32 per tile", w, in, out, cyc);
    run<54>("conv loop shape, 4 ds_write2_b32 per tile", w, in, out, cyc);
    run<58>("conv loop shape, 8 ds_write2_b32 per tile", w, in, out, cyc);
    run<62>("conv loop shape, 2 ds_write_b128 per tile", w, in, out, cyc);
    run<64>("conv loop shape, 4 ds_write_b128 per tile", w, in, out, cyc);
    run<72>("conv loop shape, 2 global_load_dwordx4 per tile", w, in, out, cyc);
    run<74>("conv loop shape, 4 global_load_dwordx4 per tile", w, in, out, cyc);
    run<78>("conv loop shape, 8 global_load_dwordx4 per tile", w, in, out, cyc);
    run<42>("LDS, 2 chains, barrier, 2 global loads + 2 ds_write_b128 per tile", w, in, out, cyc);
    run<44>("LDS, 2 chains, barrier, 4 global loads + 4 ds_write_b128 per tile", w, in, out, cyc);
    run<48>("LDS, 2 chains, barrier, 8 global loads + 8 ds_write_b128 per tile", w, in, out, cyc);
  }
  return 0;
}
