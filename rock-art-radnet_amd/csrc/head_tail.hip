// The tail of classifier_layer (resnet50.py:260-279) and its losses (losses.py:69-95) as TWO launches instead of five.
//
//   radnet_head_tail_fwd   AveragePooling2D((7,7)) + Flatten + the two Dense heads (+ softmax) per RoI and, when targets are
//                          given, the detector losses and the gradient w.r.t. the logits / regression outputs in the same
//                          launch (one workgroup per RoI; the last one to finish sums the per-RoI loss terms in RoI order).
//   radnet_head_tail_bwd   Dense backward (dw, db, dfeat) + average-pool backward fused with the ReLU mask of res5c.
//
// They replace radnet_avgpool_fwd + radnet_dense_heads_fwd + radnet_det_loss and radnet_dense_heads_bwd +
// radnet_avgpool_bwd_relu on the training step's head lane (20 x 31 outputs: five latency-bound launches of 5-19 us each);
// the separate entry points stay for the Keras-style test_on_batch / predict calls and the VGG16 head.
// RoIs come in `groups` of r / groups rows (per-GPU mini-batch): every group is its own reference step with its own
// normalisers; a group flagged idle gets zero gradient rows and no loss.
#include "radnet_internal.h"

namespace {

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

struct TailArgs {
  const float* y5;      // [r][hw][c] res5c output (post-ReLU)
  const float* w;       // [c][ldw] class columns first, then regression columns
  const float* b;       // [ldw]
  float* feat;          // [r][c]
  float* pcls;          // [r][nc]
  float* pregr;         // [r][nreg]
  const float* y1;      // [r][nc] or null (inference)
  const float* y2;      // [r][2*nreg]
  float* dz;            // [r][nc+nreg]
  float* losses;        // [groups][3]: cls, regr, accuracy
  const int* group_live;// [groups] or null (all live)
  double* partial;      // [r][4] scratch: ce, acc, smooth-L1 sum, unused
  unsigned* ticket;     // zero between launches
  int r, hw, c, ldw, nc, nreg, groups;
};

template <int NP>
__global__ void __launch_bounds__(256) head_tail_fwd_kernel(TailArgs g) {
  __shared__ float red[4][NP];
  __shared__ float z[NP];
  __shared__ float s_q[NP];
  __shared__ double dred[4];
  __shared__ int s_last;
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c4 = g.c >> 2;
  // ---- average pool over the hw positions: this thread's float4 columns tid, tid + 256, ...
  float acc[NP];
#pragma unroll
  for (int n = 0; n < NP; ++n) acc[n] = 0.f;
  const float4* src = reinterpret_cast<const float4*>(g.y5) + (long long)r * g.hw * c4;
  const float inv = (float)g.hw;
  for (int col = tid; col < c4; col += 256) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = 0; p < g.hw; ++p) {
      const float4 v = src[(long long)p * c4 + col];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const float4 f = make_float4(s.x / inv, s.y / inv, s.z / inv, s.w / inv);
    reinterpret_cast<float4*>(g.feat)[(long long)r * c4 + col] = f;
    // ---- dense heads: this thread's 4 channels against their weight rows
    const float fv[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float4* wr = reinterpret_cast<const float4*>(g.w + (long long)(col * 4 + e) * NP);
#pragma unroll
      for (int q = 0; q < NP / 4; ++q) {
        const float4 v = wr[q];
        acc[4 * q + 0] += fv[e] * v.x; acc[4 * q + 1] += fv[e] * v.y; acc[4 * q + 2] += fv[e] * v.z; acc[4 * q + 3] += fv[e] * v.w;
      }
    }
  }
#pragma unroll
  for (int n = 0; n < NP; ++n) {
    const float s = wave_sum_f(acc[n]);
    if (lane == 0) red[wave][n] = s;
  }
  __syncthreads();
  const int nout = g.nc + g.nreg;
  if (tid < nout) z[tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid] + g.b[tid];
  __syncthreads();
  if (tid < nout) {
    if (tid < g.nc) {
      float mx = z[0];
      for (int i = 1; i < g.nc; ++i) mx = fmaxf(mx, z[i]);
      float s = 0.f;
      for (int i = 0; i < g.nc; ++i) s += expf(z[i] - mx);
      const float q = expf(z[tid] - mx) / s;
      s_q[tid] = q;
      g.pcls[(long long)r * g.nc + tid] = q;
    } else {
      s_q[tid] = z[tid];
      g.pregr[(long long)r * g.nreg + (tid - g.nc)] = z[tid];
    }
  }
  if (g.y1 == nullptr) return;          // inference: uniform for the whole grid
  __syncthreads();
  // ---- losses.py:69-95 for this RoI.  Normaliser of the regression loss: sum over the GROUP's label entries (exact:
  //      they are 0/1) + 1e-4 per element
  const int rg = g.r / g.groups, grp = r / rg, r0 = grp * rg;
  const bool live = g.group_live == nullptr || g.group_live[grp] != 0;
  double ms = 0.0;
  for (int i = tid; i < rg * g.nreg; i += 256) {
    const int rr = r0 + i / g.nreg, j = i - (i / g.nreg) * g.nreg;
    ms += (double)g.y2[(long long)rr * 2 * g.nreg + j];
  }
  for (int o = 32; o > 0; o >>= 1) ms += __shfl_xor(ms, o, 64);
  if (lane == 0) dred[wave] = ms;
  __syncthreads();
  const float den = (float)(1e-4 * (double)rg * g.nreg + (dred[0] + dred[1] + dred[2] + dred[3]));
  const float lo = 1e-7f, hi = 1.0f - 1e-7f;
  const float* t = g.y1 + (long long)r * g.nc;
  float* dzr = g.dz + (long long)r * nout;
  if (tid == 0) {
    // categorical cross-entropy on the re-normalised, clipped softmax (Keras 2) + categorical accuracy
    float S = 0.f;
    for (int i = 0; i < g.nc; ++i) S += s_q[i];
    float ce = 0.f;
    int am_t = 0, am_q = 0;
    float sum_ao = 0.f;
    for (int i = 0; i < g.nc; ++i) {
      const float o = s_q[i] / S, oc = fminf(fmaxf(o, lo), hi);
      ce -= t[i] * logf(oc);
      if (t[i] > t[am_t]) am_t = i;
      if (s_q[i] > s_q[am_q]) am_q = i;
      const float ak = (o >= lo && o <= hi) ? -t[i] / oc : 0.f;
      sum_ao += ak * o;
    }
    float sum_dqq = 0.f;
    for (int i = 0; i < g.nc; ++i) {
      const float o = s_q[i] / S, oc = fminf(fmaxf(o, lo), hi);
      const float ak = (o >= lo && o <= hi) ? -t[i] / oc : 0.f;
      sum_dqq += (ak - sum_ao) / S / (float)rg * s_q[i];
    }
    for (int i = 0; i < g.nc; ++i) {
      const float o = s_q[i] / S, oc = fminf(fmaxf(o, lo), hi);
      const float ak = (o >= lo && o <= hi) ? -t[i] / oc : 0.f;
      const float dq = (ak - sum_ao) / S / (float)rg;
      dzr[i] = live ? s_q[i] * (dq - sum_dqq) : 0.f;
    }
    double sl = 0.0;
    for (int j = 0; j < g.nreg; ++j) {
      const float mask = g.y2[(long long)r * 2 * g.nreg + j], tgt = g.y2[(long long)r * 2 * g.nreg + g.nreg + j];
      const float x = tgt - s_q[g.nc + j], ax = fabsf(x);
      sl += mask * (ax <= 1.0f ? 0.5f * x * x : ax - 0.5f);
      const float d = ax <= 1.0f ? x : (x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f));
      dzr[g.nc + j] = live ? -(mask * d) / den : 0.f;
    }
    // Hand-off to whichever workgroup finishes last (cdna_hip_programming.md 6 Guideline 16, write-through form, as the
    // split-K reduction of conv_mfma.hip): every handed-off value is an agent-scope (sc1, write-through) store, drained by
    // this lane before it takes its ticket; the reader uses agent-scope loads.  No release / acquire cache maintenance.
    double* pr = g.partial + 4ll * r;
    __hip_atomic_store(&pr[0], (double)ce, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&pr[1], (am_t == am_q) ? 1.0 : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&pr[2], sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&pr[3], (double)den, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned tk = __hip_atomic_fetch_add(g.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = tk == (unsigned)(g.r - 1);
    if (s_last) __hip_atomic_store(g.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // ready for the next launch
  }
  __syncthreads();
  if (!s_last || tid >= g.groups) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");          // compiler-only: keeps the loads below the ticket
  // the last workgroup: per group, the loss terms summed in RoI order (deterministic)
  {
    const int gi = tid;
    const bool glive = g.group_live == nullptr || g.group_live[gi] != 0;
    double c = 0.0, a = 0.0, sl = 0.0, dn = 1.0;
    for (int rr = gi * rg; rr < (gi + 1) * rg; ++rr) {
      const double* pr = g.partial + 4ll * rr;
      c += __hip_atomic_load(&pr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      a += __hip_atomic_load(&pr[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      sl += __hip_atomic_load(&pr[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      dn = __hip_atomic_load(&pr[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (glive) {
      g.losses[3 * gi + 0] = (float)(c / rg);
      g.losses[3 * gi + 1] = (float)(sl / dn);
      g.losses[3 * gi + 2] = (float)(a / rg);
    }
  }
}

// Dense backward + average-pool backward for KB channels per workgroup:
//   dw[k][n] (+)= sum_r feat[r][k] dz[r][n];  db[n] (+)= sum_r dz[r][n] (workgroup 0);
//   dfeat[r][k] = sum_n dz[r][n] w[k][n];  g_last[r][p][k] = y5[r][p][k] > 0 ? dfeat[r][k] / hw : 0
constexpr int KB = 32;
__global__ void __launch_bounds__(256) head_tail_bwd_kernel(const float* __restrict__ feat, const float* __restrict__ dz, const float* __restrict__ y5,
                                                            int r, int hw, int c, const float* __restrict__ w, int np, int nout,
                                                            float* __restrict__ dw, float* __restrict__ db, float* __restrict__ dfeat,
                                                            float* __restrict__ g_last, int acc) {
  extern __shared__ float sm[];
  float* sdz = sm;                  // [r][np]
  float* sdf = sm + r * np;         // [r][KB]
  float* sw = sdf + r * KB;         // [KB][np]
  const int tid = threadIdx.x, k0 = blockIdx.x * KB;
  for (int i = tid; i < r * np; i += 256) {
    const int rr = i / np, n = i - rr * np;
    sdz[i] = n < nout ? dz[(long long)rr * nout + n] : 0.f;
  }
  for (int i = tid; i < KB * np; i += 256) sw[i] = w[(long long)k0 * np + i];
  __syncthreads();
  // dw: thread -> (k = tid / 8, 4 consecutive n starting at (tid % 8) * 4 [+ 32 per pass for np = 64])
  {
    const int k = tid >> 3;
    for (int n0 = (tid & 7) * 4; n0 < np; n0 += 32) {
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int rr = 0; rr < r; ++rr) {
        const float f = feat[(long long)rr * c + k0 + k];
        const float4 d = *reinterpret_cast<const float4*>(sdz + rr * np + n0);
        s.x += f * d.x; s.y += f * d.y; s.z += f * d.z; s.w += f * d.w;
      }
      float4* dst = reinterpret_cast<float4*>(dw + (long long)(k0 + k) * np + n0);
      if (n0 + 0 >= nout) s.x = 0.f;
      if (n0 + 1 >= nout) s.y = 0.f;
      if (n0 + 2 >= nout) s.z = 0.f;
      if (n0 + 3 >= nout) s.w = 0.f;
      if (acc) {
        const float4 o = *dst;
        s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
      }
      *dst = s;
    }
  }
  // dfeat for this workgroup's channels
  for (int i = tid; i < r * KB; i += 256) {
    const int rr = i / KB, k = i - rr * KB;
    float s = 0.f;
    for (int n = 0; n < nout; ++n) s += sdz[rr * np + n] * sw[k * np + n];
    sdf[i] = s;
    dfeat[(long long)rr * c + k0 + k] = s;
  }
  if (blockIdx.x == 0 && tid < np) {
    float s = 0.f;
    for (int rr = 0; rr < r; ++rr) s += sdz[rr * np + tid];
    const float v = tid < nout ? s : 0.f;
    db[tid] = acc ? db[tid] + v : v;
  }
  __syncthreads();
  // average-pool backward with res5c's ReLU mask: rows (roi, position), 32 channels = 128 bytes per row
  const float inv = (float)hw;
  const int k = tid & 31;
  for (int row = tid >> 5; row < r * hw; row += 8) {
    const int rr = row / hw;
    const long long idx = (long long)row * c + k0 + k;
    g_last[idx] = y5[idx] > 0.f ? sdf[rr * KB + k] / inv : 0.f;
  }
}

}  // namespace

extern "C" uint64_t radnet_head_tail_scratch_bytes(int32_t r) { return (uint64_t)r * 4 * sizeof(double) + 256; }

extern "C" int radnet_head_tail_fwd(radnet_ctx* ctx, const float* y5, int32_t r, int32_t hw, int32_t c, const float* w, int32_t ldw, const float* b,
                                    int32_t nc, int32_t nreg, float* feat, float* p_cls, float* p_regr, const float* y1, const float* y2,
                                    float* dz, float* losses, int32_t groups, const int32_t* group_live, void* scratch) {
  if (!ctx || !y5 || !w || !b || !feat || !p_cls || !p_regr) return RADNET_ERR_ARG;
  if (nc + nreg > ldw || (ldw != 32 && ldw != 64)) RADNET_FAIL(ctx, RADNET_ERR_ARG, "head_tail: ldw=%d must be 32 or 64 and >= nc+nreg=%d", ldw, nc + nreg);
  if (c % 4) RADNET_FAIL(ctx, RADNET_ERR_ARG, "head_tail: c %% 4");
  if (groups < 1 || r % groups || groups > 256) RADNET_FAIL(ctx, RADNET_ERR_ARG, "head_tail: %d RoIs in %d groups", r, groups);
  if (y1 && (!y2 || !dz || !losses || !scratch)) RADNET_FAIL(ctx, RADNET_ERR_ARG, "head_tail: targets without y2 / dz / losses / scratch");
  TailArgs g{};
  g.y5 = y5; g.w = w; g.b = b; g.feat = feat; g.pcls = p_cls; g.pregr = p_regr; g.y1 = y1; g.y2 = y2; g.dz = dz; g.losses = losses;
  g.group_live = group_live;
  g.ticket = (unsigned*)scratch;                          // first 256 bytes: the arrival counter (zeroed once by the caller)
  g.partial = (double*)((char*)scratch + 256);
  g.r = r; g.hw = hw; g.c = c; g.ldw = ldw; g.nc = nc; g.nreg = nreg; g.groups = groups;
  if (ldw == 32) hipLaunchKernelGGL(head_tail_fwd_kernel<32>, dim3(r), dim3(256), 0, ctx->stream, g);
  else hipLaunchKernelGGL(head_tail_fwd_kernel<64>, dim3(r), dim3(256), 0, ctx->stream, g);
  RADNET_CHECK_LAUNCH(ctx, "head_tail_fwd");
  return RADNET_OK;
}

extern "C" int radnet_head_tail_bwd(radnet_ctx* ctx, const float* feat, const float* dz, const float* y5, int32_t r, int32_t hw, int32_t c,
                                    const float* w, int32_t ldw, int32_t nout, float* dw, float* db, float* dfeat, float* g_last,
                                    int32_t accumulate) {
  if (!ctx || !feat || !dz || !y5 || !w || !dw || !db || !dfeat || !g_last) return RADNET_ERR_ARG;
  if (c % KB || (ldw != 32 && ldw != 64) || nout > ldw) RADNET_FAIL(ctx, RADNET_ERR_ARG, "head_tail_bwd: c=%d ldw=%d nout=%d", c, ldw, nout);
  const size_t smem = ((size_t)r * ldw + (size_t)r * KB + (size_t)KB * ldw) * sizeof(float);
  if (smem > 64 * 1024) RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "head_tail_bwd: r=%d too large", r);
  hipLaunchKernelGGL(head_tail_bwd_kernel, dim3(c / KB), dim3(256), smem, ctx->stream, feat, dz, y5, r, hw, c, w, ldw, nout, dw, db, dfeat,
                     g_last, accumulate);
  RADNET_CHECK_LAUNCH(ctx, "head_tail_bwd");
  return RADNET_OK;
}
