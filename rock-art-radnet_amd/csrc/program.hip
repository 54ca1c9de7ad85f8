// Layer programs and the composed entry points of the C ABI (SURVEY.md 8b "minimum exports": rpn_forward, train_step,
// predict_tile, allreduce_grads).
//
//   radnet_program_run     executes a static list of launches (radnet_op[]) on the context's stream: the layer programs of
//                          nn_base / rpn_layer / classifier_layer (resnet50.py:150-281, rpn.py:12-66) and their backward
//                          programs, as the host scheduler lays them out once per input size.  Nothing is allocated, nothing
//                          synchronises: a program can be recorded into a hipGraph by running it on a capturing stream.
//   radnet_rpn_forward     model_rpn.predict (RADNet.py:552): base program + RPN program.
//   radnet_predict_tile    one tile of RADNet.predict (RADNet.py:520-600) up to the classifier outputs: preprocess, base, RPN,
//                          decode + NMS (rpn.rpn_to_roi), RoI crop-resize, stage 5, dense heads.
//   radnet_train_step      one reference training iteration (train.py:288-402) on one image, one stream: the two places where
//                          the reference draws from NumPy's global RNG (utils.py:785-813, train.py:93-129) are host callbacks.
//   radnet_allreduce_grads sum-all-reduce of a flat fp32 gradient arena over RCCL; RCCL is bound at run time
//                          (dlopen), so the library loads on machines without it.
#include <dlfcn.h>

#include "radnet_internal.h"

namespace {

int run_wino(radnet_ctx* ctx, const radnet_op& o, bool reuse) {
  const float* x = (const float*)o.p[0];
  float* V = (float*)o.p[1];
  const float* U = (const float*)o.p[2];
  float* M = (float*)o.p[3];
  const float* scale = (const float*)o.p[4];
  const float* shift = (const float*)o.p[5];
  float* y = (float*)o.p[6];
  const int nb = o.i[0], h = o.i[1], w = o.i[2], c = o.i[3], n = o.i[4], T = o.i[5], act = o.i[6], ldy = o.i[7];
  const bool f4 = o.i[8] == 4;                 // Winograd output tile: 4 = F(4x4,3x3) (36 positions), else F(2x2,3x3) (16)
  // roofline leg: the LAYER is timed (three kernels) and credited its algorithmic flops (class 3)
  const int timed = ctx->timing;
  if (timed) {
    radnet_timing_begin(ctx);
    ctx->timing = 0;
  }
  int rc = reuse ? RADNET_OK : (f4 ? radnet_winograd4_input(ctx, x, nb, h, w, c, V) : radnet_winograd_input(ctx, x, nb, h, w, c, V));
  if (rc == RADNET_OK) rc = radnet_gemm_batched(ctx, V, U, M, f4 ? 36 : 16, T, n, c);
  if (rc == RADNET_OK)
    rc = f4 ? radnet_winograd4_output(ctx, M, nb, h, w, n, scale, shift, act, y, ldy) : radnet_winograd_output(ctx, M, nb, h, w, n, scale, shift, act, y, ldy);
  if (timed) {
    ctx->timing = timed;
    radnet_timing_end(ctx, 3, 2.0 * nb * h * w * (double)n * 9.0 * c);
  }
  return rc;
}

int run_wino_wgrad(radnet_ctx* ctx, const radnet_op& o) {
  const float* dy = (const float*)o.p[0];
  const float* V = (const float*)o.p[1];
  float* dZ = (float*)o.p[2];
  float* dU = (float*)o.p[3];
  float* dw = (float*)o.p[4];
  const float* gscale = (const float*)o.p[5];       // per-output-channel factor on dy (frozen BN scale of a res block conv) or null
  const int nb = o.i[0], h = o.i[1], w = o.i[2], c = o.i[3], n = o.i[4], ld_dy = o.i[5], T = o.i[6], ldw = o.i[7], mode = o.i[8];
  const int timed = ctx->timing;
  if (timed) {
    radnet_timing_begin(ctx);
    ctx->timing = 0;
  }
  const bool f4 = o.i[9] == 4;
  int rc = f4 ? radnet_winograd4_dy(ctx, dy, nb, h, w, n, ld_dy, gscale, dZ) : radnet_winograd_dy(ctx, dy, nb, h, w, n, ld_dy, gscale, dZ);
  if (rc == RADNET_OK) rc = radnet_wgrad_batched(ctx, V, dZ, dU, f4 ? 36 : 16, T, c, n, 0);
  if (rc == RADNET_OK)
    rc = f4 ? radnet_winograd4_filter_grad(ctx, dU, c, n, ldw, dw, mode == 1 ? 1 : 0) : radnet_winograd_filter_grad(ctx, dU, c, n, ldw, dw, mode == 1 ? 1 : 0);
  if (timed) {
    ctx->timing = timed;
    radnet_timing_end(ctx, 3, 2.0 * nb * h * w * (double)n * 9.0 * c);
  }
  return rc;
}

}  // namespace

extern "C" int radnet_program_run(radnet_ctx* ctx, const radnet_op* ops, int32_t n_ops) {
  if (!ctx || (n_ops > 0 && !ops) || n_ops < 0) return RADNET_ERR_ARG;
  for (int k = 0; k < n_ops; ++k) {
    const radnet_op& o = ops[k];
    int rc;
    switch (o.kind) {
      case RADNET_OP_CONV_FWD: rc = radnet_conv_fwd(ctx, &o.conv); break;
      case RADNET_OP_CONV_DGRAD: rc = radnet_conv_dgrad(ctx, &o.conv); break;
      case RADNET_OP_CONV_WGRAD: rc = radnet_conv_wgrad(ctx, &o.conv); break;
      case RADNET_OP_CONV_BWD: rc = radnet_conv_bwd(ctx, &o.conv); break;
      case RADNET_OP_CONV_FWD_PAIR:
        if (k + 1 >= n_ops || ops[k + 1].kind != RADNET_OP_NOP) RADNET_FAIL(ctx, RADNET_ERR_ARG, "program: CONV_FWD_PAIR at %d without its second descriptor (a NOP slot)", k);
        rc = radnet_conv_fwd_pair(ctx, &o.conv, &ops[k + 1].conv);
        break;
      case RADNET_OP_CONV_BNECK: {
        const int extra = o.i[0] ? 2 : 1;
        for (int e = 1; e <= extra; ++e)
          if (k + e >= n_ops || ops[k + e].kind != RADNET_OP_NOP) RADNET_FAIL(ctx, RADNET_ERR_ARG, "program: CONV_BNECK at %d without its %d following descriptors (NOP slots)", k, extra);
        rc = radnet_conv_bottleneck(ctx, &o.conv, &ops[k + 1].conv, o.i[0] ? &ops[k + 2].conv : nullptr);
        break;
      }
      case RADNET_OP_NOP: rc = RADNET_OK; break;
      case RADNET_OP_MAXPOOL:
        rc = radnet_maxpool_fwd(ctx, (const float*)o.p[0], (float*)o.p[1], o.i[0], o.i[1], o.i[2], o.i[3], o.i[4], o.i[5]);
        break;
      case RADNET_OP_COLSUM:
        rc = radnet_colsum(ctx, (const float*)o.p[0], o.i[0], o.i[1], o.i[2], (const float*)o.p[1], (float*)o.p[2], o.i[3]);
        break;
      case RADNET_OP_WINO: rc = run_wino(ctx, o, false); break;
      case RADNET_OP_WINO_REUSE: rc = run_wino(ctx, o, true); break;
      case RADNET_OP_WINO_WGRAD: rc = run_wino_wgrad(ctx, o); break;
      case RADNET_OP_SCATTER:
        rc = radnet_scatter_strided(ctx, (const float*)o.p[0], o.i[0], o.i[1], o.i[2], o.i[3], o.i[4], o.i[5], o.i[6], (const float*)o.p[1],
                                    (float*)o.p[2]);
        break;
      case RADNET_OP_FILL0: rc = radnet_fill_zero(ctx, const_cast<void*>(o.p[0]), ((uint64_t)(uint32_t)o.i[1] << 32) | (uint32_t)o.i[0]); break;
      case RADNET_OP_RELU_MASK:
        rc = radnet_relu_mask(ctx, (float*)const_cast<void*>(o.p[0]), (const float*)o.p[1], ((int64_t)o.i[1] << 32) | (uint32_t)o.i[0]);
        break;
      case RADNET_OP_CHAIN: rc = radnet_chain_run(ctx, (radnet_chain*)const_cast<void*>(o.p[0])); break;
      case RADNET_OP_ROI_BWD:
        rc = radnet_roi_resize_bwd(ctx, (const float*)o.p[0], o.i[0], o.i[1], o.i[2], (const float*)o.p[1], o.i[3], o.i[4], (float*)o.p[2]);
        break;
      default: RADNET_FAIL(ctx, RADNET_ERR_ARG, "program: unknown op kind %d at position %d", o.kind, k);
    }
    if (rc != RADNET_OK) return rc;
  }
  return RADNET_OK;
}

extern "C" int radnet_rpn_forward(radnet_ctx* ctx, const radnet_op* base_ops, int32_t n_base, const radnet_op* rpn_ops, int32_t n_rpn) {
  int rc = radnet_program_run(ctx, base_ops, n_base);
  if (rc == RADNET_OK) rc = radnet_program_run(ctx, rpn_ops, n_rpn);
  return rc;
}

namespace {

// (x1, y1, x2, y2) int64 proposals -> (x, y, w, h) fp32 RoIs for RoiPoolingConv (RADNet.py:566-567); rows past *n repeat
// row 0 (RADNet.py:115-122 pads the last chunk with copies of its first RoI)
__global__ void rois_xywh_kernel(const long long* __restrict__ R, const int* __restrict__ n_dev, int rows, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  const int n = *n_dev;
  const int j = i < n ? i : 0;
  const long long x1 = R[4 * j], y1 = R[4 * j + 1], x2 = R[4 * j + 2], y2 = R[4 * j + 3];
  out[4 * i] = (float)x1;
  out[4 * i + 1] = (float)y1;
  out[4 * i + 2] = (float)(x2 - x1);
  out[4 * i + 3] = (float)(y2 - y1);
}

// classifier_layer forward; with targets (y1 != 0) the detector losses and dz come out of the tail's launch (head_tail.hip)
int head_forward(radnet_ctx* ctx, const radnet_head_desc& h, const float* y1 = nullptr, const float* y2 = nullptr, float* dz = nullptr,
                 float* losses = nullptr) {
  int rc = radnet_roi_resize_fwd(ctx, h.fmap, h.fh, h.fw, h.fc, h.rois, h.n_rois, h.pool, h.pooled);
  if (rc == RADNET_OK) rc = radnet_program_run(ctx, h.fwd_ops, h.n_fwd);
  if (rc == RADNET_OK)
    rc = radnet_head_tail_fwd(ctx, h.y5, h.n_rois, h.hw, h.feat_c, h.dense_w, h.dense_ld, h.dense_b, h.nc, h.nreg, h.feat, h.p_cls, h.p_regr, y1,
                              y2, dz, losses, 1, nullptr, h.tail_scratch);
  return rc;
}

}  // namespace

extern "C" int radnet_predict_tile(radnet_ctx* ctx, const radnet_tile_desc* t) {
  if (!ctx || !t) return RADNET_ERR_ARG;
  int rc = RADNET_OK;
  if (t->img_u8) rc = radnet_preprocess_bgr(ctx, t->img_u8, t->h, t->w, 4, t->x);
  if (rc == RADNET_OK) rc = radnet_rpn_forward(ctx, t->base_ops, t->n_base, t->rpn_ops, t->n_rpn);
  if (rc == RADNET_OK)
    rc = radnet_rpn_to_roi(ctx, t->pred, t->ld_pred, t->fh, t->fw, t->a, t->anchor_wh_host, t->std_scaling, 1, t->overlap_thresh, t->max_boxes,
                           t->R, t->Rp, t->Rn, t->prop_ws);
  if (rc != RADNET_OK || t->head == nullptr) return rc;
  const radnet_head_desc& h = *t->head;
  hipLaunchKernelGGL(rois_xywh_kernel, dim3(radnet_cdiv(h.n_rois, 256)), dim3(256), 0, ctx->stream, (const long long*)t->R, t->Rn, h.n_rois,
                     const_cast<float*>(h.rois));
  RADNET_CHECK_LAUNCH(ctx, "rois_xywh");
  return head_forward(ctx, h);
}

extern "C" int radnet_train_step(radnet_ctx* ctx, const radnet_train_desc* d, const radnet_host_hooks* hooks, float* losses5,
                                 int32_t* took_head_step) {
  if (!ctx || !d || !hooks || !hooks->subsample_anchors || !hooks->select_rois || !losses5 || !took_head_step) return RADNET_ERR_ARG;
  *took_head_step = 0;
  for (int i = 0; i < 5; ++i) losses5[i] = 0.f;
  const int M = d->fh * d->fw, A = d->a;
  hipStream_t st = ctx->stream;
  // ---- phase A: anchor targets (utils.calc_region_props), device half; label maps -> pinned host
  int rc = radnet_anchor_targets(ctx, d->gt, d->gt_is_bg, d->g, d->width, d->height, d->w, d->h, d->fw, d->fh, d->anchor_sizes_host, d->ns,
                                 d->anchor_ratios_host, d->nr, d->rpn_stride, d->rpn_max_overlap, d->valid, d->overlap, d->regr, d->best_anchor,
                                 d->n_for_gt, d->at_scratch);
  if (rc != RADNET_OK) return rc;
  RADNET_CHECK_HIP(ctx, hipMemcpyAsync(d->h_valid, d->valid, (size_t)A * M, hipMemcpyDeviceToHost, st));
  RADNET_CHECK_HIP(ctx, hipMemcpyAsync(d->h_overlap, d->overlap, (size_t)A * M, hipMemcpyDeviceToHost, st));
  // ---- phase B: base forward (once: the base is frozen, train.py:288,291,393 recompute the same values) + RPN forward
  if (d->img_u8) {
    rc = radnet_preprocess_bgr(ctx, d->img_u8, d->h, d->w, 4, d->x);
    if (rc != RADNET_OK) return rc;
  }
  rc = radnet_rpn_forward(ctx, d->base_ops, d->n_base, d->rpn_fwd_ops, d->n_rpn_fwd);
  if (rc != RADNET_OK) return rc;
  RADNET_CHECK_HIP(ctx, hipStreamSynchronize(st));
  // host half of the labeller: random subsampling on the caller's RNG (utils.py:785-813)
  const int n_pos = hooks->subsample_anchors(hooks->user, d->h_valid, d->h_overlap, A, d->fh, d->fw);
  // Data parallel (world > 1): the collectives must stay symmetric.  A rank whose image is dropped, or whose proposals
  // overlap no box, still joins BOTH exchanges with its (zero) gradient arena and applies the same optimizer steps as its
  // peers -- otherwise they block in ncclAllReduce for a rank that never comes (radnet_hip/trainer.py does the same).
  auto head_update = [&]() -> int {
    int r = RADNET_OK;
    if (d->world > 1) r = radnet_allreduce_grads(ctx, d->head_opt.g, d->head_opt.n);
    if (r == RADNET_OK && d->n_head_wino > 0)
      r = radnet_adam_step_fused(ctx, d->head_opt.p, d->head_opt.g, d->head_opt.m, d->head_opt.v, d->head_opt.n, d->head_opt.t, d->head_opt.lr, 0.9f,
                                 0.999f, 1e-7f, 1.0f / (float)(d->world > 0 ? d->world : 1), 1, 0, 0, nullptr, nullptr, nullptr, d->head_wino, d->n_head_wino);
    else if (r == RADNET_OK)
      r = radnet_adam_step(ctx, d->head_opt.p, d->head_opt.g, d->head_opt.m, d->head_opt.v, d->head_opt.n, d->head_opt.t, d->head_opt.lr, 0.9f,
                           0.999f, 1e-7f, 1.0f / (float)(d->world > 0 ? d->world : 1), 1);
    if (r == RADNET_OK && d->head_shift) r = radnet_affine_vec(ctx, d->head_shift, d->head_scale, d->head_bias, d->head_t0, d->head_bias_len);
    return r;
  };
  auto rpn_update = [&]() -> int {
    int r = RADNET_OK;
    if (d->world > 1) r = radnet_allreduce_grads(ctx, d->rpn_opt.g, d->rpn_opt.n);
    if (r == RADNET_OK)
      r = radnet_adam_step(ctx, d->rpn_opt.p, d->rpn_opt.g, d->rpn_opt.m, d->rpn_opt.v, d->rpn_opt.n, d->rpn_opt.t, d->rpn_opt.lr, 0.9f, 0.999f,
                           1e-7f, 1.0f / (float)(d->world > 0 ? d->world : 1), 1);
    if (r == RADNET_OK && d->wino_w)
      r = d->wino_form == 4 ? radnet_winograd4_filter(ctx, d->wino_w, d->wino_c, d->wino_n, d->wino_ldw, d->wino_u)
                            : radnet_winograd_filter(ctx, d->wino_w, d->wino_c, d->wino_n, d->wino_ldw, d->wino_u);
    return r;
  };
  if (n_pos < 0) {                          // labeller failure: the reference's generator skips the sample (utils.py:461-465)
    *took_head_step = -1;
    if (d->world > 1) {                     // nothing of this image touched a gradient: both arenas hold zeros
      rc = rpn_update();
      if (rc == RADNET_OK) rc = head_update();
      if (rc == RADNET_OK) RADNET_CHECK_HIP(ctx, hipStreamSynchronize(st));
    }
    return rc;
  }
  RADNET_CHECK_HIP(ctx, hipMemcpyAsync(d->valid, d->h_valid, (size_t)A * M, hipMemcpyHostToDevice, st));
  rc = radnet_anchor_targets_pack(ctx, d->valid, d->overlap, d->regr, d->fw, d->fh, A, d->std_scaling, d->y_cls, d->y_regr);
  // ---- phase C: model_rpn.train_on_batch (train.py:288)
  if (rc == RADNET_OK)
    rc = radnet_rpn_loss(ctx, d->pred, d->ld_pred, d->y_cls, d->y_regr, M, A, d->bce_mode, d->dz, d->ld_pred, d->rpn_losses, d->loss_scratch8);
  if (rc == RADNET_OK) rc = radnet_program_run(ctx, d->rpn_bwd_ops, d->n_rpn_bwd);
  if (rc == RADNET_OK) rc = rpn_update();
  // ---- phase D: re-predict with the updated RPN (train.py:291), proposals, RoI labelling
  if (rc == RADNET_OK) rc = radnet_program_run(ctx, d->rpn_refwd_ops, d->n_rpn_refwd);
  if (rc == RADNET_OK)
    rc = radnet_rpn_to_roi(ctx, d->pred, d->ld_pred, d->fh, d->fw, A, d->anchor_wh_host, d->std_scaling, 1, d->overlap_thresh, d->max_boxes, d->R,
                           d->Rp, d->Rn, d->prop_ws);
  if (rc == RADNET_OK)
    rc = radnet_roi_targets(ctx, d->R, d->max_boxes, d->gt, d->gt_cls, d->g, d->width, d->height, d->rw, d->rh, d->rpn_stride, d->min_overlap,
                            d->max_overlap, d->regr_std_host4, d->bg_class, d->keep, d->roi_cls, d->roi_box, d->roi_t, d->roi_iou, d->Rn);
  if (rc != RADNET_OK) return rc;
  RADNET_CHECK_HIP(ctx, hipMemcpyAsync(d->h_roi_cls, d->roi_cls, (size_t)d->max_boxes * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  RADNET_CHECK_HIP(ctx, hipMemcpyAsync(d->h_n, d->Rn, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  RADNET_CHECK_HIP(ctx, hipMemcpyAsync(losses5, d->rpn_losses, 2 * sizeof(float), hipMemcpyDeviceToHost, st));
  RADNET_CHECK_HIP(ctx, hipStreamSynchronize(st));
  const int n = *d->h_n < d->max_boxes ? *d->h_n : d->max_boxes;
  const radnet_head_desc& h = *d->head;
  // train.get_selected_samples on the caller's RNG (train.py:93-129); 0 = calc_iou kept nothing, the head step is skipped
  const int k = n > 0 ? hooks->select_rois(hooks->user, d->h_roi_cls, n, d->h_sel, h.n_rois) : 0;
  if (k <= 0) {
    if (d->world > 1) {                     // the head arena holds zeros: join the peers' exchange and update (see above)
      rc = head_update();
      if (rc == RADNET_OK) RADNET_CHECK_HIP(ctx, hipStreamSynchronize(st));
    }
    return rc;
  }
  if (k != h.n_rois) RADNET_FAIL(ctx, RADNET_ERR_ARG, "train_step: select_rois returned %d indices, the head plan holds %d", k, h.n_rois);
  RADNET_CHECK_HIP(ctx, hipMemcpyAsync(d->sel, d->h_sel, (size_t)k * sizeof(int32_t), hipMemcpyHostToDevice, st));
  rc = radnet_roi_batch_pack(ctx, d->sel, k, d->roi_cls, d->roi_box, d->roi_t, h.nc, d->bg_class, const_cast<float*>(h.rois), d->y1, d->y2);
  // ---- model_classifier.train_on_batch (train.py:393)
  if (rc == RADNET_OK) rc = head_forward(ctx, h, d->y1, d->y2, d->head_dz, d->det_losses);
  if (rc == RADNET_OK)
    rc = radnet_dense_heads_bwd(ctx, h.feat, d->head_dz, h.n_rois, h.feat_c, h.dense_w, h.dense_ld, h.nc + h.nreg, d->dense_dw, d->dense_db, d->dfeat, 1);
  if (rc == RADNET_OK) rc = radnet_avgpool_bwd_relu(ctx, d->dfeat, h.y5, h.n_rois, h.hw, h.feat_c, d->g_last);
  if (rc == RADNET_OK) rc = radnet_program_run(ctx, d->head_bwd_ops, d->n_head_bwd);
  if (rc == RADNET_OK) rc = head_update();
  if (rc != RADNET_OK) return rc;
  RADNET_CHECK_HIP(ctx, hipMemcpyAsync(losses5 + 2, d->det_losses, 3 * sizeof(float), hipMemcpyDeviceToHost, st));
  RADNET_CHECK_HIP(ctx, hipStreamSynchronize(st));
  *took_head_step = 1;
  return RADNET_OK;
}

// ---- RCCL, bound at run time --------------------------------------------------------------------------------------------
namespace {

struct NcclId {             // ncclUniqueId: 128 opaque bytes, passed BY VALUE to ncclCommInitRank
  char b[128];
};

struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, NcclId, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};

Rccl* rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return r.lib ? &r : nullptr;
  tried = true;
  // the copy the process already holds (PyTorch-ROCm ships its own) first: two RCCLs in one process do not share state
  for (const char* name : {"librccl.so.1", "librccl.so"}) {
    r.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
    if (r.lib) break;
  }
  if (!r.lib)
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
  if (!r.lib) return nullptr;
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
  r.AllReduce = (decltype(r.AllReduce))dlsym(r.lib, "ncclAllReduce");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
  if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy) {
    r.lib = nullptr;
    return nullptr;
  }
  return &r;
}

}  // namespace

extern "C" int radnet_comm_unique_id(char out128[128]) {
  Rccl* r = rccl();
  if (!r || !out128) return RADNET_ERR_UNSUPPORTED;
  return r->GetUniqueId(out128) == 0 ? RADNET_OK : RADNET_ERR_HIP;
}

extern "C" int radnet_comm_init(radnet_ctx* ctx, int32_t world, int32_t rank, const char id128[128]) {
  if (!ctx || !id128 || world < 1 || rank < 0 || rank >= world) return RADNET_ERR_ARG;
  Rccl* r = rccl();
  if (!r) RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "comm_init: librccl not found");
  if (ctx->comm) RADNET_FAIL(ctx, RADNET_ERR_ARG, "comm_init: the context already holds a communicator");
  NcclId id;
  memcpy(id.b, id128, 128);
  RADNET_CHECK_HIP(ctx, hipSetDevice(ctx->device));
  void* comm = nullptr;
  const int e = r->CommInitRank(&comm, world, id, rank);
  if (e != 0) RADNET_FAIL(ctx, RADNET_ERR_HIP, "ncclCommInitRank: %s", r->GetErrorString ? r->GetErrorString(e) : "error");
  ctx->comm = comm;
  ctx->comm_world = world;
  return RADNET_OK;
}

extern "C" int radnet_comm_destroy(radnet_ctx* ctx) {
  if (!ctx) return RADNET_ERR_ARG;
  if (ctx->comm) {
    Rccl* r = rccl();
    if (r) (void)r->CommDestroy(ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_world = 0;
  }
  return RADNET_OK;
}

extern "C" int radnet_allreduce_grads(radnet_ctx* ctx, float* grads, int64_t count) {
  if (!ctx || !grads || count < 0) return RADNET_ERR_ARG;
  if (!ctx->comm) RADNET_FAIL(ctx, RADNET_ERR_ARG, "allreduce_grads: no communicator (radnet_comm_init)");
  Rccl* r = rccl();
  if (!r) RADNET_FAIL(ctx, RADNET_ERR_UNSUPPORTED, "allreduce_grads: librccl not found");
  // in place, fp32 (ncclFloat32 = 7), sum (ncclSum = 0), on the context's stream: ordered after the backward that produced the
  // gradients and before the optimizer step that consumes them, with no host synchronisation
  const int e = r->AllReduce(grads, grads, (size_t)count, 7, 0, ctx->comm, ctx->stream);
  if (e != 0) RADNET_FAIL(ctx, RADNET_ERR_HIP, "ncclAllReduce: %s", r->GetErrorString ? r->GetErrorString(e) : "error");
  ctx->comm_calls += 1;
  ctx->comm_elems += count;
  return RADNET_OK;
}

extern "C" int radnet_comm_stats(radnet_ctx* ctx, int64_t* calls, int64_t* elements) {
  if (!ctx) return RADNET_ERR_ARG;
  if (calls) *calls = ctx->comm_calls;
  if (elements) *elements = ctx->comm_elems;
  return RADNET_OK;
}
