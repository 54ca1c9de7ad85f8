// Internal definitions shared by the gfx950 kernel translation units of libradnet_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <map>
#include <memory>
#include <tuple>
#include <utility>
#include <vector>

#include "radnet_hip.h"

struct radnet_timing_slot {
  double ms = 0.0;
  double flops = 0.0;
  int64_t launches = 0;
};

// Measured tile / split choice per GEMM problem shape (filled by the launchers when autotuning is on).
struct radnet_shape_key {
  int kind, m, n, k, c, npos, stride;
  bool operator<(const radnet_shape_key& o) const {
    return std::tie(kind, m, n, k, c, npos, stride) < std::tie(o.kind, o.m, o.n, o.k, o.c, o.npos, o.stride);
  }
};
struct radnet_tuned {
  int a, b, splits;
  float ms;
  int waves = 4;
};

// Autotune mode 2 (radnet_set_autotune): a problem shape not measured yet adopts the choice of the measured shape that
// differs from it in M only, by at most M/4 (training on tiles whose size changes from sample to sample: every layer's M
// moves with the image, N / K / taps do not).  Nearest M wins; nullptr when there is none.
inline const radnet_tuned* radnet_tuned_neighbour(const std::map<radnet_shape_key, radnet_tuned>& table, const radnet_shape_key& key) {
  const radnet_tuned* best = nullptr;
  int best_d = key.m / 4 + 1;
  for (const auto& kv : table) {
    const radnet_shape_key& o = kv.first;
    if (o.kind != key.kind || o.n != key.n || o.k != key.k || o.c != key.c || o.npos != key.npos || o.stride != key.stride) continue;
    const int d = o.m > key.m ? o.m - key.m : key.m - o.m;
    if (d < best_d) { best_d = d; best = &kv.second; }
  }
  return best;
}

// Device-resident work-unit / fix-up tables of a K-split GEMM launch (conv_mfma.hip: get_unit_table).
struct radnet_unit_table {
  int* d_units = nullptr;
  unsigned* d_counters = nullptr;      // one arrival counter per K-split output tile; zero between launches
  int n_units = 0, n_split_tiles = 0, n_slots = 0;
};

struct radnet_ctx {
  int autotune = 0;
  int force_a = 0, force_b = 0, force_splits = 0, force_waves = 0;      // radnet_force_config (tests): overrides tuned / heuristic choices
  // measured launch choices; contexts of one engine share ONE table (radnet_share_tuning), calls come from one host thread
  std::shared_ptr<std::map<radnet_shape_key, radnet_tuned>> tuned = std::make_shared<std::map<radnet_shape_key, radnet_tuned>>();
  std::map<std::array<int, 6>, radnet_unit_table> unit_tables;
  // conv geometry -> device row table (conv_mfma.hip: get_row_table): read-only once built, so the contexts of one engine share
  // them like the tuning table (radnet_share_tuning) -- a context that first meets a geometry inside a graph capture must not
  // have to build (allocate + copy) its table there.  Freed by the last context that holds the map.
  struct RowTables {
    std::map<std::array<int, 11>, void*> m;
    ~RowTables();
  };
  std::shared_ptr<RowTables> row_tables = std::make_shared<RowTables>();
  hipEvent_t tune_ev0 = nullptr, tune_ev1 = nullptr;
  int device = 0;
  hipStream_t stream = nullptr;
  char err[512] = {0};
  void* ws = nullptr;
  uint64_t ws_bytes = 0;
  // timing of GEMM-class launches with HIP events on `stream` (bench roofline leg)
  int timing = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  radnet_timing_slot slots[5];      // 0 fwd, 1 dgrad, 2 wgrad, 3 Winograd layers of a program, 4 dgrad + wgrad of a layer in one launch
  hipEvent_t arm0 = nullptr, arm1 = nullptr;      // event pair of the launch being timed (radnet_timing_arm), or null
  void* pair_capture = nullptr;     // conv_mfma.hip: radnet_conv_bwd collects the two launches of a layer here instead of issuing them
  // pending (not yet resolved) event pairs are resolved lazily to avoid a sync per launch
  static constexpr int kMaxPending = 4096;
  hipEvent_t pend0[kMaxPending];
  hipEvent_t pend1[kMaxPending];
  int pend_cls[kMaxPending];
  double pend_flops[kMaxPending];
  int n_pending = 0;
  int n_events_alloc = 0;
  // written only by the diagnostic build (make diag, -DRADNET_DIAG_STAMPS): per-workgroup s_memtime stamps
  unsigned long long* diag_stamps = nullptr;
  // data-parallel exchange (program.hip): RCCL communicator of this context, bound at run time
  void* comm = nullptr;
  int comm_world = 0;
  int64_t comm_calls = 0, comm_elems = 0;      // radnet_allreduce_grads issued on this context (radnet_comm_stats)
  // Ordered (run-to-run reproducible) reductions: arrival counters and partial-sum scratch of the kernels that hand a
  // reduction to their last-arriving workgroup (split weight gradients, column sums, the RPN loss sums).  One block of
  // device memory per context (radnet_create), zeroed once; every launch leaves its counters at zero again.
  // deterministic = 1 (default; RADNET_DETERMINISTIC=0 turns it off): no floating-point atomics anywhere in a training step.
  int deterministic = 1;
  char* aux = nullptr;
};

// layout of radnet_ctx::aux (bytes)
constexpr size_t kAuxWgradCounters = 0;                          // 65536 x u32: one per (problem, k tile, n tile) of a split wgrad launch
constexpr size_t kAuxWgradCounterCount = 65536;
constexpr size_t kAuxColsumCounters = kAuxWgradCounters + kAuxWgradCounterCount * 4;      // 1024 x u32: one per 64-column block
constexpr size_t kAuxColsumCounterCount = 1024;
constexpr size_t kAuxLossPartials = kAuxColsumCounters + kAuxColsumCounterCount * 4;      // 1024 x 4 doubles (rpn_loss_sums blocks)
constexpr size_t kAuxLossBlocks = 1024;
constexpr size_t kAuxColsumScratch = kAuxLossPartials + kAuxLossBlocks * 4 * 8;           // kAuxColsumRows x 65536 floats
constexpr size_t kAuxColsumRows = 32;
constexpr size_t kAuxColsumCols = 65536;
constexpr size_t kAuxBytes = kAuxColsumScratch + kAuxColsumRows * kAuxColsumCols * 4;

#define RADNET_FAIL(ctx, code, ...)                         \
  do {                                                      \
    snprintf((ctx)->err, sizeof((ctx)->err), __VA_ARGS__);  \
    return (code);                                          \
  } while (0)

#define RADNET_CHECK_HIP(ctx, expr)                                                              \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess) RADNET_FAIL(ctx, RADNET_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

#define RADNET_CHECK_LAUNCH(ctx, what)                                                              \
  do {                                                                                              \
    hipError_t _e = hipGetLastError();                                                              \
    if (_e != hipSuccess) RADNET_FAIL(ctx, RADNET_ERR_HIP, "launch %s: %s", what, hipGetErrorString(_e)); \
  } while (0)

// timing helpers (api.cpp)
void radnet_timing_begin(radnet_ctx* ctx);
void radnet_timing_end(radnet_ctx* ctx, int cls, double flops);
// Kernel-exact form for a single launch: arm() hands out the event pair, the launch site passes it to hipExtLaunchKernelGGL
// (start / stop taken from the dispatch itself, as rocprofv3 reads them), end_armed() books it.  The marker-event form above
// brackets several launches (a Winograd layer) and includes the dispatch latency of its two markers (~5 us per bracket).
void radnet_timing_arm(radnet_ctx* ctx);
void radnet_timing_end_armed(radnet_ctx* ctx, int cls, double flops);

static inline int radnet_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
// environment switch: set and neither empty nor "0"
static inline bool radnet_env_flag(const char* name) {
  const char* v = getenv(name);
  return v != nullptr && v[0] != '\0' && !(v[0] == '0' && v[1] == '\0');
}

// Autotune helper: one warm-up launch, then `iters` launches bracketed by HIP events on the ctx stream.
template <typename F>
static inline int radnet_time_launches(radnet_ctx* ctx, F&& launch, int iters, float* ms_out) {
  if (!ctx->tune_ev0) {
    RADNET_CHECK_HIP(ctx, hipEventCreate(&ctx->tune_ev0));
    RADNET_CHECK_HIP(ctx, hipEventCreate(&ctx->tune_ev1));
  }
  int rc = launch();
  if (rc != RADNET_OK) return rc;
  RADNET_CHECK_HIP(ctx, hipEventRecord(ctx->tune_ev0, ctx->stream));
  for (int i = 0; i < iters; ++i) {
    rc = launch();
    if (rc != RADNET_OK) return rc;
  }
  RADNET_CHECK_HIP(ctx, hipEventRecord(ctx->tune_ev1, ctx->stream));
  RADNET_CHECK_HIP(ctx, hipEventSynchronize(ctx->tune_ev1));
  float ms = 0.f;
  RADNET_CHECK_HIP(ctx, hipEventElapsedTime(&ms, ctx->tune_ev0, ctx->tune_ev1));
  *ms_out = ms / iters;
  return RADNET_OK;
}

// exact floor(m / d) for m, d < 2^20 as (m * magic) >> 40  (m*d < 2^40, see conv_mfma.hip)
static inline uint64_t radnet_div_magic(uint32_t d) { return ((1ull << 40) + d - 1) / d; }
