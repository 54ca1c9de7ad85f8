// Context management, error reporting and GEMM-launch timing for libradnet_hip.so.
#include "radnet_internal.h"

extern "C" int radnet_version(void) { return 100; }

extern "C" int radnet_create(int device, void* hip_stream, radnet_ctx** out) {
  if (!out) return RADNET_ERR_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return RADNET_ERR_HIP;   // no GPU: fail loudly
  if (device < 0 || device >= ndev) return RADNET_ERR_ARG;
  if (hipSetDevice(device) != hipSuccess) return RADNET_ERR_HIP;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return RADNET_ERR_HIP;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return RADNET_ERR_UNSUPPORTED;   // gfx950-only code objects
  radnet_ctx* c = new radnet_ctx();
  c->device = device;
  c->stream = (hipStream_t)hip_stream;
  if (const char* e = getenv("RADNET_DETERMINISTIC")) c->deterministic = strcmp(e, "0") != 0;
  if (hipMalloc((void**)&c->aux, kAuxBytes) != hipSuccess || hipMemset(c->aux, 0, kAuxBytes) != hipSuccess) {
    delete c;
    return RADNET_ERR_HIP;
  }
  *out = c;
  return RADNET_OK;
}

extern "C" void radnet_destroy(radnet_ctx* ctx) {
  if (!ctx) return;
  for (int i = 0; i < ctx->n_events_alloc; ++i) {
    (void)hipEventDestroy(ctx->pend0[i]);
    (void)hipEventDestroy(ctx->pend1[i]);
  }
  for (auto& kv : ctx->unit_tables) {
    if (kv.second.d_units) (void)hipFree(kv.second.d_units);
    if (kv.second.d_counters) (void)hipFree(kv.second.d_counters);
  }
  if (ctx->tune_ev0) (void)hipEventDestroy(ctx->tune_ev0);
  if (ctx->tune_ev1) (void)hipEventDestroy(ctx->tune_ev1);
  if (ctx->aux) (void)hipFree(ctx->aux);
  delete ctx;
}

extern "C" const char* radnet_last_error(radnet_ctx* ctx) { return ctx ? ctx->err : "null context"; }

extern "C" int radnet_sync(radnet_ctx* ctx) {
  if (!ctx) return RADNET_ERR_ARG;
  RADNET_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return RADNET_OK;
}

extern "C" int radnet_set_stream(radnet_ctx* ctx, void* hip_stream) {
  if (!ctx) return RADNET_ERR_ARG;
  ctx->stream = (hipStream_t)hip_stream;
  return RADNET_OK;
}

extern "C" int radnet_set_autotune(radnet_ctx* ctx, int enable) {
  if (!ctx) return RADNET_ERR_ARG;
  ctx->autotune = enable == 2 ? 2 : (enable ? 1 : 0);
  return RADNET_OK;
}

radnet_ctx::RowTables::~RowTables() {
  for (auto& kv : m)
    if (kv.second) (void)hipFree(kv.second);
}

extern "C" int radnet_set_deterministic(radnet_ctx* ctx, int enable) {
  if (!ctx) return RADNET_ERR_ARG;
  ctx->deterministic = enable ? 1 : 0;
  return RADNET_OK;
}

extern "C" int radnet_get_deterministic(radnet_ctx* ctx) { return ctx ? (ctx->deterministic ? 1 : 0) : RADNET_ERR_ARG; }

extern "C" int radnet_share_tuning(radnet_ctx* ctx, radnet_ctx* owner) {
  if (!ctx || !owner) return RADNET_ERR_ARG;
  ctx->tuned = owner->tuned;
  ctx->row_tables = owner->row_tables;      // read-only device tables per conv geometry (tables this context built itself are freed here)
  return RADNET_OK;
}

extern "C" int radnet_tuned_shapes(radnet_ctx* ctx) { return ctx ? (int)ctx->tuned->size() : -1; }

extern "C" int radnet_tune_save(radnet_ctx* ctx, const char* path) {
  if (!ctx || !path) return RADNET_ERR_ARG;
  FILE* f = fopen(path, "w");
  if (!f) RADNET_FAIL(ctx, RADNET_ERR_ARG, "tune_save: cannot open %s", path);
  fprintf(f, "# radnet tuned GEMM launch shapes v2: kind m n k c npos stride | tile_a tile_b slices ms waves\n");
  for (const auto& kv : *ctx->tuned)
    fprintf(f, "%d %d %d %d %d %d %d %d %d %d %.6f %d\n", kv.first.kind, kv.first.m, kv.first.n, kv.first.k, kv.first.c, kv.first.npos,
            kv.first.stride, kv.second.a, kv.second.b, kv.second.splits, (double)kv.second.ms, kv.second.waves);
  fclose(f);
  return RADNET_OK;
}

extern "C" int radnet_tune_load(radnet_ctx* ctx, const char* path) {
  if (!ctx || !path) return RADNET_ERR_ARG;
  FILE* f = fopen(path, "r");
  if (!f) RADNET_FAIL(ctx, RADNET_ERR_ARG, "tune_load: cannot open %s", path);
  char line[256];
  int n = 0;
  while (fgets(line, sizeof(line), f)) {
    if (line[0] == '#') continue;
    radnet_shape_key k{};
    radnet_tuned t{};
    double ms = 0.0;
    int waves = 4;
    if (sscanf(line, "%d %d %d %d %d %d %d %d %d %d %lf %d", &k.kind, &k.m, &k.n, &k.k, &k.c, &k.npos, &k.stride, &t.a, &t.b, &t.splits, &ms,
               &waves) < 11) continue;
    t.waves = waves == 8 ? 8 : 4;
    // tiles: 64 / 128; the forward / data-gradient kernel (kinds 0, 1, 8) also has 32x64 and 32x32 in its 4-wave form
    const bool small_ok = (k.kind & 2) == 0 && t.waves == 4 && t.a == 32 && (t.b == 32 || t.b == 64);
    if (!small_ok && ((t.a != 64 && t.a != 128) || (t.b != 64 && t.b != 128))) continue;
    if (t.splits == 0 || t.splits > 64 || t.splits < -64) continue;
    t.ms = (float)ms;
    (*ctx->tuned)[k] = t;
    ++n;
  }
  fclose(f);
  return n >= 0 ? RADNET_OK : RADNET_ERR_ARG;
}

extern "C" int radnet_force_config(radnet_ctx* ctx, int tile_a, int tile_b, int slices) {
  if (!ctx) return RADNET_ERR_ARG;
  const bool small_ok = tile_a == 32 && (tile_b == 32 || tile_b == 64);      // forward / data-gradient launches only
  if (tile_a != 0 && !small_ok && ((tile_a != 64 && tile_a != 128) || (tile_b != 64 && tile_b != 128)))
    RADNET_FAIL(ctx, RADNET_ERR_ARG, "force_config: tiles must be 64 or 128 (or 32x64 / 32x32)");
  ctx->force_a = tile_a;
  ctx->force_b = tile_b;
  ctx->force_splits = slices;
  return RADNET_OK;
}

extern "C" int radnet_force_waves(radnet_ctx* ctx, int waves) {
  if (!ctx || (waves != 0 && waves != 4 && waves != 8)) return RADNET_ERR_ARG;
  ctx->force_waves = waves;
  return RADNET_OK;
}

extern "C" int radnet_set_workspace(radnet_ctx* ctx, void* ws, uint64_t bytes) {
  if (!ctx) return RADNET_ERR_ARG;
  ctx->ws = ws;
  ctx->ws_bytes = bytes;
  return RADNET_OK;
}

// ---- timing -----------------------------------------------------------------------------------------
static void resolve_pending(radnet_ctx* ctx) {
  for (int i = 0; i < ctx->n_pending; ++i) {
    float ms = 0.f;
    if (hipEventSynchronize(ctx->pend1[i]) == hipSuccess && hipEventElapsedTime(&ms, ctx->pend0[i], ctx->pend1[i]) == hipSuccess) {
      radnet_timing_slot& s = ctx->slots[ctx->pend_cls[i]];
      s.ms += ms;
      s.flops += ctx->pend_flops[i];
      s.launches += 1;
    }
  }
  ctx->n_pending = 0;
}

void radnet_timing_begin(radnet_ctx* ctx) {
  if (!ctx->timing) return;
  if (ctx->n_pending >= radnet_ctx::kMaxPending) resolve_pending(ctx);
  int i = ctx->n_pending;
  if (i >= ctx->n_events_alloc) {
    (void)hipEventCreate(&ctx->pend0[i]);
    (void)hipEventCreate(&ctx->pend1[i]);
    ctx->n_events_alloc = i + 1;
  }
  (void)hipEventRecord(ctx->pend0[i], ctx->stream);
}

void radnet_timing_end(radnet_ctx* ctx, int cls, double flops) {
  if (!ctx->timing) return;
  int i = ctx->n_pending;
  (void)hipEventRecord(ctx->pend1[i], ctx->stream);
  ctx->pend_cls[i] = cls;
  ctx->pend_flops[i] = flops;
  ctx->n_pending = i + 1;
}

void radnet_timing_arm(radnet_ctx* ctx) {
  ctx->arm0 = ctx->arm1 = nullptr;
  if (!ctx->timing) return;
  if (ctx->n_pending >= radnet_ctx::kMaxPending) resolve_pending(ctx);
  int i = ctx->n_pending;
  if (i >= ctx->n_events_alloc) {
    (void)hipEventCreate(&ctx->pend0[i]);
    (void)hipEventCreate(&ctx->pend1[i]);
    ctx->n_events_alloc = i + 1;
  }
  ctx->arm0 = ctx->pend0[i];
  ctx->arm1 = ctx->pend1[i];
}

void radnet_timing_end_armed(radnet_ctx* ctx, int cls, double flops) {
  if (!ctx->timing || !ctx->arm0) return;
  int i = ctx->n_pending;
  ctx->pend_cls[i] = cls;
  ctx->pend_flops[i] = flops;
  ctx->n_pending = i + 1;
  ctx->arm0 = ctx->arm1 = nullptr;
}

extern "C" int radnet_timing_enable(radnet_ctx* ctx, int enable) {
  if (!ctx) return RADNET_ERR_ARG;
  if (!enable) resolve_pending(ctx);
  ctx->timing = enable;
  return RADNET_OK;
}

extern "C" int radnet_timing_reset(radnet_ctx* ctx) {
  if (!ctx) return RADNET_ERR_ARG;
  resolve_pending(ctx);
  for (auto& s : ctx->slots) s = radnet_timing_slot();
  return RADNET_OK;
}

extern "C" int radnet_timing_read(radnet_ctx* ctx, int cls, double* ms, int64_t* launches, double* flops) {
  if (!ctx || cls < 0 || cls > 4) return RADNET_ERR_ARG;
  resolve_pending(ctx);
  if (ms) *ms = ctx->slots[cls].ms;
  if (launches) *launches = ctx->slots[cls].launches;
  if (flops) *flops = ctx->slots[cls].flops;
  return RADNET_OK;
}
