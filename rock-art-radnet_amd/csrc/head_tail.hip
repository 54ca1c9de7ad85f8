// The tail of classifier_layer (resnet50.py:260-279) and its losses (losses.py:69-95) in ONE launch.
//
//   radnet_head_tail_fwd   AveragePooling2D((7,7)) + Flatten + the two Dense heads (+ softmax) per RoI and, when targets are
//                          given, the detector losses and the gradient w.r.t. the logits / regression outputs in the same
//                          launch (r x 8 workgroups stream the activations; the last channel slice of a RoI finishes it, the
//                          last RoI sums the loss terms in RoI order).
//
// It replaces radnet_avgpool_fwd + radnet_dense_heads_fwd (+ radnet_det_loss) on the training step's head lane and on the
// predict path.  Measured alone on the chip, 20 RoIs (tools/head_tail_timing.py): 12.6 us against 6.8 + 16.1 without targets,
// 28.2 us against 6.8 + 16.1 + 15.0 with them.  (A fused dense-backward + average-pool-backward twin was built and measured
// too: 26 us against 5.6 + 6.2 for the two separate launches -- every workgroup repeated the small dense part -- and dropped.)
// The separate entry points stay for the Keras-style test_on_batch calls and the VGG16 head.
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
  float* zpart;         // [r][slices][ldw] partial dense sums of the channel slices
  double* partial;      // [r][4] per-RoI loss terms: ce, acc, smooth-L1 sum, normaliser
  unsigned* tickets;    // [1 + r]: [0] RoIs finished, [1 + roi] slices finished; all zero between launches
  int r, hw, c, ldw, nc, nreg, groups;
};

constexpr int kSlices = 8;       // channel slices per RoI: r x 8 workgroups stream the 8 MB of res5c activations
constexpr int kPosGroups = 4;    // position groups per workgroup (threads 64 x 4)

// grid (r, kSlices) x 256 threads.  Thread (col = tid & 63, pg = tid >> 6) sums positions pg, pg + 4, ... of float4 column
// slice * 64 + col -- all of its loads are issued before the first add --, the four groups meet in LDS, wave 0 multiplies
// the pooled 256 channels of the slice into the dense heads; the LAST slice of a RoI to finish adds the slices' partial
// sums in slice order (deterministic), applies softmax and, with targets, the detector losses of that RoI; the LAST RoI sums
// the loss terms in RoI order.  Hand-offs: agent-scope (sc1, write-through) stores drained before a relaxed ticket,
// agent-scope loads by the reader (cdna_hip_programming.md 6 Guideline 16, write-through form; as conv_mfma.hip's split-K).
template <int NP>
__global__ void __launch_bounds__(256) head_tail_fwd_kernel(TailArgs g) {
  __shared__ float4 pool[kPosGroups][64];
  __shared__ float z[NP];
  __shared__ float s_q[NP];
  __shared__ double dred[4];
  __shared__ int s_last;
  const int r = blockIdx.x, slice = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c4 = g.c >> 2, cols_per_slice = c4 / kSlices;             // 64 for c = 2048
  const int col = slice * cols_per_slice + lane;
  const bool col_ok = lane < cols_per_slice;
  // ---- average pool: positions wave, wave + 4, ... of this thread's column
  constexpr int kMaxPos = 16;
  float4 v[kMaxPos];
  const float4* src = reinterpret_cast<const float4*>(g.y5) + (long long)r * g.hw * c4 + col;
#pragma unroll
  for (int q = 0; q < kMaxPos; ++q) {
    const int p = wave + kPosGroups * q;
    v[q] = (col_ok && p < g.hw) ? src[(long long)p * c4] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float4 s = v[0];
#pragma unroll
  for (int q = 1; q < kMaxPos; ++q) { s.x += v[q].x; s.y += v[q].y; s.z += v[q].z; s.w += v[q].w; }
  for (int p = wave + kPosGroups * kMaxPos; p < g.hw; p += kPosGroups) {      // hw > 64: the rest, one by one
    const float4 u = col_ok ? src[(long long)p * c4] : make_float4(0.f, 0.f, 0.f, 0.f);
    s.x += u.x; s.y += u.y; s.z += u.z; s.w += u.w;
  }
  pool[wave][lane] = s;
  __syncthreads();
  if (wave == 0) {
    const float inv = (float)g.hw;
    float4 t = pool[0][lane];
#pragma unroll
    for (int q = 1; q < kPosGroups; ++q) { const float4 u = pool[q][lane]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
    const float4 f = make_float4(t.x / inv, t.y / inv, t.z / inv, t.w / inv);
    if (col_ok) reinterpret_cast<float4*>(g.feat)[(long long)r * c4 + col] = f;
    // ---- dense heads, this slice's share: 4 channels per lane against their weight rows, summed over the wave
    float acc[NP];
#pragma unroll
    for (int n = 0; n < NP; ++n) acc[n] = 0.f;
    if (col_ok) {
      const float fv[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float4* wr = reinterpret_cast<const float4*>(g.w + (long long)(col * 4 + e) * NP);
#pragma unroll
        for (int q = 0; q < NP / 4; ++q) {
          const float4 u = wr[q];
          acc[4 * q + 0] += fv[e] * u.x; acc[4 * q + 1] += fv[e] * u.y; acc[4 * q + 2] += fv[e] * u.z; acc[4 * q + 3] += fv[e] * u.w;
        }
      }
    }
    float* zp = g.zpart + ((long long)r * kSlices + slice) * NP;
#pragma unroll
    for (int n = 0; n < NP; ++n) {
      const float t2 = wave_sum_f(acc[n]);
      if (lane == 0) __hip_atomic_store(zp + n, t2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned tk = __hip_atomic_fetch_add(g.tickets + 1 + r, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = tk == (unsigned)(kSlices - 1);
      if (s_last) __hip_atomic_store(g.tickets + 1 + r, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  if (!s_last) return;                                   // uniform per workgroup
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // compiler-only: keeps the loads below the ticket
  // ---- the RoI's last slice: logits = slices in order + bias, softmax, outputs
  const int nout = g.nc + g.nreg;
  if (tid < nout) {
    float zz = 0.f;
    for (int sl = 0; sl < kSlices; ++sl)
      zz += __hip_atomic_load(g.zpart + ((long long)r * kSlices + sl) * NP + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    z[tid] = zz + g.b[tid];
  }
  __syncthreads();
  if (tid < nout) {
    if (tid < g.nc) {
      float mx = z[0];
      for (int i = 1; i < g.nc; ++i) mx = fmaxf(mx, z[i]);
      float sm = 0.f;
      for (int i = 0; i < g.nc; ++i) sm += expf(z[i] - mx);
      const float q = expf(z[tid] - mx) / sm;
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
  int last_roi = 0;
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
    double* pr = g.partial + 4ll * r;
    __hip_atomic_store(&pr[0], (double)ce, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&pr[1], (am_t == am_q) ? 1.0 : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&pr[2], sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&pr[3], (double)den, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned tk = __hip_atomic_fetch_add(g.tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_roi = tk == (unsigned)(g.r - 1);
    if (last_roi) __hip_atomic_store(g.tickets, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // ready for the next launch
    s_last = last_roi;
  }
  __syncthreads();
  if (!s_last || tid >= g.groups) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // the last RoI: per group, the loss terms summed in RoI order (deterministic)
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

}  // namespace

// scratch: [tickets: (1 + r) u32, padded to 256-byte multiple][partial: r x 4 doubles][zpart: r x kSlices x 64 floats]
static size_t tail_ticket_bytes(int r) { return ((size_t)(1 + r) * 4 + 255) / 256 * 256; }
extern "C" uint64_t radnet_head_tail_scratch_bytes(int32_t r) {
  return (uint64_t)(tail_ticket_bytes(r) + (size_t)r * 4 * sizeof(double) + (size_t)r * kSlices * 64 * sizeof(float));
}

extern "C" int radnet_head_tail_fwd(radnet_ctx* ctx, const float* y5, int32_t r, int32_t hw, int32_t c, const float* w, int32_t ldw, const float* b,
                                    int32_t nc, int32_t nreg, float* feat, float* p_cls, float* p_regr, const float* y1, const float* y2,
                                    float* dz, float* losses, int32_t groups, const int32_t* group_live, void* scratch) {
  if (!ctx || !y5 || !w || !b || !feat || !p_cls || !p_regr) return RADNET_ERR_ARG;
  if (nc + nreg > ldw || (ldw != 32 && ldw != 64)) RADNET_FAIL(ctx, RADNET_ERR_ARG, "head_tail: ldw=%d must be 32 or 64 and >= nc+nreg=%d", ldw, nc + nreg);
  if (c % 4) RADNET_FAIL(ctx, RADNET_ERR_ARG, "head_tail: c %% 4");
  if (groups < 1 || r % groups || groups > 256) RADNET_FAIL(ctx, RADNET_ERR_ARG, "head_tail: %d RoIs in %d groups", r, groups);
  if (y1 && (!y2 || !dz || !losses)) RADNET_FAIL(ctx, RADNET_ERR_ARG, "head_tail: targets without y2 / dz / losses");
  TailArgs g{};
  g.y5 = y5; g.w = w; g.b = b; g.feat = feat; g.pcls = p_cls; g.pregr = p_regr; g.y1 = y1; g.y2 = y2; g.dz = dz; g.losses = losses;
  g.group_live = group_live;
  if (!scratch) RADNET_FAIL(ctx, RADNET_ERR_ARG, "head_tail: scratch is required (arrival counters, slice partial sums)");
  if ((c / 4) % kSlices || (c / 4) / kSlices > 64) RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "head_tail: c=%d (needs c / 32 <= 64 float4 columns per slice)", c);
  g.tickets = (unsigned*)scratch;                         // arrival counters (zeroed once by the caller, left at zero by every launch)
  g.partial = (double*)((char*)scratch + tail_ticket_bytes(r));
  g.zpart = (float*)((char*)g.partial + (size_t)r * 4 * sizeof(double));
  g.r = r; g.hw = hw; g.c = c; g.ldw = ldw; g.nc = nc; g.nreg = nreg; g.groups = groups;
  if (ldw == 32) hipLaunchKernelGGL(head_tail_fwd_kernel<32>, dim3(r, kSlices), dim3(256), 0, ctx->stream, g);
  else hipLaunchKernelGGL(head_tail_fwd_kernel<64>, dim3(r, kSlices), dim3(256), 0, ctx->stream, g);
  RADNET_CHECK_LAUNCH(ctx, "head_tail_fwd");
  return RADNET_OK;
}
