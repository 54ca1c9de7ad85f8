/*
 * radnet_hip.h -- C ABI of libradnet_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * Faster R-CNN train / predict hot path of rock-art-radnet.
 *
 * The reference has NO native layer: everything below replaces work it hands to the TensorFlow
 * runtime (Keras graph in faster_rcnn/base_models/resnet50.py, rpn.py:12-66, RoiPoolingConv.py,
 * FixedBatchNormalization.py, losses.py, keras Adam) or does in single-threaded NumPy/Python
 * (rpn.py:68-455, utils.py:554-822).  Each entry point cites the reference lines it stands in for.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (e.g. torch-ROCm tensor.data_ptr());
 *     the library never allocates or frees caller-visible memory;
 *   - work is enqueued on the hipStream_t given to radnet_create(); calls return immediately;
 *   - activations are NHWC fp32; conv weights are [K = kh*kw*cin][N = cout] row-major (the Keras
 *     HWIO kernel flattened), dense weights [in][out];
 *   - return value 0 = ok, negative = error; radnet_last_error() gives the message;
 *   - one context per thread / stream (no internal locking).
 */
#ifndef RADNET_HIP_H
#define RADNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct radnet_ctx radnet_ctx;

#define RADNET_OK 0
#define RADNET_ERR_ARG (-1)
#define RADNET_ERR_HIP (-2)
#define RADNET_ERR_UNSUPPORTED (-3)

/* ---- context ------------------------------------------------------------------------------ */
int radnet_create(int device, void* hip_stream, radnet_ctx** out);
void radnet_destroy(radnet_ctx* ctx);
const char* radnet_last_error(radnet_ctx* ctx);
int radnet_sync(radnet_ctx* ctx);
/* Re-binds the context to another HIP stream (e.g. a stream in capture mode while a sequence of launches is recorded
 * into a hipGraph; the library performs no allocation, host synchronisation or autotuning for shapes it has already
 * run, so a recorded sequence replays as is). */
int radnet_set_stream(radnet_ctx* ctx, void* hip_stream);
int radnet_version(void);
/* Scratch the library may use for split-K partial sums (caller-owned, >= bytes). */
int radnet_set_workspace(radnet_ctx* ctx, void* ws, uint64_t bytes);
/* Autotuning of the conv GEMM launch shape.  While enabled, the FIRST conv_fwd / conv_dgrad / conv_wgrad call of
 * each problem shape times every (output tile, split-K) candidate on the ctx stream (synchronising it) and caches
 * the fastest; later calls of that shape reuse the choice.  Shapes are static per image size, so a warm-up step
 * tunes the whole layer program.  radnet_tuned_shapes() returns the number of cached shapes.
 * enable = 2: as 1, but a new shape first adopts the cached choice of the shape that differs from it in M only, by at most
 * M/4 (nearest M), and is measured only when there is none -- for training on tiles whose size changes from sample to
 * sample (rotation / shear augmentation, augmentation.py:158-271), where every new size would otherwise re-tune every layer. */
int radnet_set_autotune(radnet_ctx* ctx, int enable);
int radnet_tuned_shapes(radnet_ctx* ctx);
/* Persist / restore the measured choices (text, one shape per line).  A context that loaded a table runs no trial
 * launches for the shapes in it: restarts skip the tuning step, and a profiler sees steady-state launches only. */
int radnet_tune_save(radnet_ctx* ctx, const char* path);
/* `ctx` uses (reads and extends) the table of `owner` from now on: the contexts an engine keeps for its concurrent lanes
 * (one per HIP stream) measure every shape once.  All calls on contexts that share a table come from one host thread.
 * The read-only device tables the weight-gradient kernel keeps per conv geometry are shared the same way, so a context can
 * meet a geometry for the first time inside a stream capture (its sibling built the table in an earlier eager run). */
int radnet_share_tuning(radnet_ctx* ctx, radnet_ctx* owner);
int radnet_tune_load(radnet_ctx* ctx, const char* path);
/* Test hook: force every following conv GEMM launch to use output tile (tile_a x tile_b in {64,128}) and `slices` K
 * slices per tile (negative = same slices with the XCD-aware workgroup order); tile_a = 0 switches it off.  Batched launches
 * (radnet_gemm_batched / radnet_wgrad_batched: the Winograd positions) take slices = 1 or -1 only: -1 renumbers the workgroups
 * so that each XCD runs a contiguous run of (problem, tile)s and the tiles that share a problem's operands share an L2 --
 * same results bit for bit, 5-9 % shorter launches for the 36 GEMMs of a Winograd layer at 1000x600. */
int radnet_force_config(radnet_ctx* ctx, int tile_a, int tile_b, int slices);
/* With radnet_force_config active: waves per workgroup of the forward / dgrad kernel (4, or 8 = every K tile halved
 * between two wave grids and summed through LDS); 0 = default (4). */
int radnet_force_waves(radnet_ctx* ctx, int waves);
/* Reproducible reductions (default ON; the environment variable RADNET_DETERMINISTIC=0 sets the default of new contexts to
 * off).  ON: every floating-point reduction that crosses workgroups -- the pixel splits of conv_wgrad / wgrad_batched /
 * conv_bwd and their bias gradients, radnet_colsum's row blocks, radnet_roi_resize_bwd's overlapping RoIs, the loss sums of
 * radnet_rpn_loss -- is handed to ONE workgroup that adds the partial results in index order, so a launch returns the same
 * bits on every run and on every stream (TF's own GPU gradients, train.py:288/393, give no such guarantee; what this buys is
 * that two schedules of the same training step can be compared bit for bit).  OFF: fp32 / fp64 atomics, whose order -- and
 * therefore last bits -- change from run to run.  Split weight-gradient launches then keep their partial tiles in the
 * workspace (radnet_set_workspace), from its end downwards; launch shapes whose partials do not fit are not used. */
int radnet_set_deterministic(radnet_ctx* ctx, int enable);
/* The context's current setting (1 / 0; negative: error) -- what bench.py's `config.reductions` reports. */
int radnet_get_deterministic(radnet_ctx* ctx);
/* Per-launch timing of the conv/GEMM kernel families with HIP events on the ctx stream (bench.py's roofline leg).
 * enable=1 starts recording; radnet_timing_read returns accumulated milliseconds and launch count since the last reset for
 * kernel class `cls`: 0 fwd, 1 dgrad, 2 wgrad, 4 dgrad + wgrad in one launch -- each launch timed from its own dispatch
 * (hipExtLaunchKernelGGL start / stop events: the kernel's begin and end, what rocprofv3 --kernel-trace reports) --, and
 * 3 Winograd 3x3 LAYERS of a program: transforms + batched GEMMs bracketed by two marker events as one unit (that bracket
 * includes the markers' dispatch latency, ~5 us) and credited the layer's algorithmic 2*M*N*9C flops. */
int radnet_timing_enable(radnet_ctx* ctx, int enable);
int radnet_timing_read(radnet_ctx* ctx, int cls, double* ms, int64_t* launches, double* flops);
int radnet_timing_reset(radnet_ctx* ctx);

/* ---- convolution as implicit GEMM on fp32 MFMA ----------------------------------------------
 * Replaces keras Conv2D / TimeDistributed(Conv2D) + FixedBatchNormalization + Add + Activation
 * (resnet50.py:41-147,183-186; rpn.py:41-64; FixedBatchNormalization.py:59-85) and their TF
 * autodiff gradients.  One descriptor describes the FORWARD convolution; dgrad / wgrad reuse it.
 *
 *   forward :  y[m][n] = act( (sum_k im2col(x)[m][k] * w[k][n]) * scale[n] + shift[n] + addend[m][n] )
 *   dgrad   :  dx[p][c] = ( sum_{kh,kw,n} (dy*gscale)[..][n] * w[(kh,kw,c)][n] + addend[p][c] ) masked
 *              by mask[p][c] > 0   (stride 1 only; ReLU backward of the producer fused as the mask)
 *   wgrad   :  dw[k][n] (+)= sum_m im2col(x)[m][k] * (dy*gscale)[m][n];
 *              db[n] (+)= sum_m (dy*gscale)[m][n] in the same launch when db is set (else radnet_colsum)
 */
typedef struct radnet_conv_desc {
  const float* x;        /* forward input  [nb][h][w][c]                                   */
  const float* w;        /* weights        [kh*kw*c][ldw]                                  */
  float* y;              /* forward output [nb*oh*ow][ldy]                                 */
  const float* scale;    /* per-output-channel scale (frozen BN: gamma/sqrt(var+eps)) or 0 */
  const float* shift;    /* per-output-channel shift (bias and BN shift folded) or 0       */
  const float* addend;   /* residual branch [m][ld_add] or 0                               */
  int32_t nb, h, w_, c;  /* input geometry                                                 */
  int32_t oh, ow;        /* output geometry                                                */
  int32_t kh, kw, stride, pad_t, pad_l;
  int32_t n;             /* output channels                                                */
  int32_t ldw, ldy, ld_add;
  int32_t act;           /* 0 none, 1 relu, 2 sigmoid on columns [0,act_cols) / linear rest */
  int32_t act_cols;
  /* backward-only fields */
  const float* dy;       /* gradient w.r.t. y (post-activation mask already applied) [m][ld_dy] */
  const float* gscale;   /* per-output-channel factor applied to dy on load (BN scale) or 0 */
  float* dx;             /* dgrad output [nb*h*w][ld_dx]                                   */
  const float* dx_add;   /* added to dx before masking (residual-path gradient) or 0       */
  const float* dx_mask;  /* dx zeroed where dx_mask <= 0 (producer's ReLU) or 0            */
  float* dw;             /* wgrad output [kh*kw*c][ldw]                                    */
  int32_t ld_dy, ld_dx, ld_dx_add, ld_dx_mask;
  int32_t dw_accumulate; /* 0: overwrite dw; 1: dw += ; 2: dw was zeroed by the caller (no memset, no atomics unless split) */
  float* db;             /* wgrad: bias gradient [n], same accumulate mode as dw, or 0                   */
} radnet_conv_desc;

int radnet_conv_fwd(radnet_ctx* ctx, const radnet_conv_desc* d);
/* Strided-batched fp32 GEMM on the same matrix-core kernel: y[p][m][n] = a[p][m][k] * b[p][k][n], p < batch, all
 * operands dense row-major and contiguous over p; k a multiple of 32, n of 4. */
int radnet_gemm_batched(radnet_ctx* ctx, const float* a, const float* b, float* y, int32_t batch, int32_t m, int32_t n, int32_t k);
/* Winograd F(2x2,3x3) form of a stride-1 'same' 3x3 convolution (same layers as radnet_conv_fwd; 2.25x fewer matrix-core
 * flops): u = filter transform of w [3][3][c][ldw] -> [16][c][n] (once per weight update); v = input transform of
 * x [nb][h][w][c] -> [16][tiles][c], tiles = nb*ceil(h/2)*ceil(w/2); radnet_gemm_batched(v, u, m, 16, tiles, n, c);
 * output transform of m [16][tiles][n] with the conv epilogue: y = act(conv * scale + shift), act 0 none / 1 relu. */
int radnet_winograd_filter(radnet_ctx* ctx, const float* w, int32_t c, int32_t n, int32_t ldw, float* u);
int radnet_winograd_input(radnet_ctx* ctx, const float* x, int32_t nb, int32_t h, int32_t w, int32_t c, float* v);
int radnet_winograd_output(radnet_ctx* ctx, const float* m, int32_t nb, int32_t oh, int32_t ow, int32_t n, const float* scale,
                           const float* shift, int32_t act, float* y, int32_t ldy);
/* Weight gradient of the same layers in the Winograd domain: dz = A dY A^T per tile of the (gscale-scaled) output gradient
 * [nb][oh][ow][ld_dy] -> [16][tiles][n]; radnet_wgrad_batched(v, dz, du, 16, tiles, c, n, 0): du[p][c][n] = sum over tiles of
 * v[p][tile][c] * dz[p][tile][n] with the v of the forward pass; dw [3][3][c][ldw] (+)= G^T du G.  The bias gradient is the
 * plain column sum of dy (radnet_colsum). */
int radnet_winograd_dy(radnet_ctx* ctx, const float* dy, int32_t nb, int32_t oh, int32_t ow, int32_t n, int32_t ld_dy,
                       const float* gscale, float* dz);
int radnet_wgrad_batched(radnet_ctx* ctx, const float* a, const float* dy, float* dw, int32_t batch, int32_t m, int32_t k, int32_t n,
                         int32_t accumulate);
int radnet_winograd_filter_grad(radnet_ctx* ctx, const float* du, int32_t c, int32_t n, int32_t ldw, float* dw, int32_t accumulate);
/* The same five transforms for Winograd F(4x4,3x3): 36 positions instead of 16 (u [36][c][n], v [36][tiles][c], m / dz
 * [36][tiles][n], batch 36 in radnet_gemm_batched / radnet_wgrad_batched) over tiles = nb*ceil(h/4)*ceil(w/4) -- 4x fewer
 * matrix-core flops than the direct 3x3 form, 1.78x fewer than F(2x2), and 0.56x the transform traffic of F(2x2); the fp32
 * result is off by about 2e-5 of the largest activation from the fp64 sum (F(2x2): 1e-6), inside the stated 2e-4. */
int radnet_winograd4_filter(radnet_ctx* ctx, const float* w, int32_t c, int32_t n, int32_t ldw, float* u);
int radnet_winograd4_input(radnet_ctx* ctx, const float* x, int32_t nb, int32_t h, int32_t w, int32_t c, float* v);
int radnet_winograd4_output(radnet_ctx* ctx, const float* m, int32_t nb, int32_t oh, int32_t ow, int32_t n, const float* scale,
                            const float* shift, int32_t act, float* y, int32_t ldy);
int radnet_winograd4_dy(radnet_ctx* ctx, const float* dy, int32_t nb, int32_t oh, int32_t ow, int32_t n, int32_t ld_dy,
                        const float* gscale, float* dz);
int radnet_winograd4_filter_grad(radnet_ctx* ctx, const float* du, int32_t c, int32_t n, int32_t ldw, float* dw, int32_t accumulate);
int radnet_conv_dgrad(radnet_ctx* ctx, const radnet_conv_desc* d);
int radnet_conv_wgrad(radnet_ctx* ctx, const radnet_conv_desc* d);
/* Weight gradient and data gradient of one layer from one descriptor (train.py's backward pass through a Conv2D: both read
 * dy).  Issued as ONE launch whose workgroups alternate between the two problems when both run as 64x64-tile, 4-wave
 * workgroups -- two short launches in a row each pay their own lockstep prologue / epilogue and launch gap, mixed they hide
 * each other's -- and as radnet_conv_wgrad followed by radnet_conv_dgrad otherwise (other launch shapes, d->dx == 0,
 * RADNET_NO_BWD_PAIR=1).  Same results as the two calls.  Timing class 4. */
int radnet_conv_bwd(radnet_ctx* ctx, const radnet_conv_desc* d);
/* Two INDEPENDENT forward convolutions with the same output grid and reduction depth -- branch2a and the shortcut conv of a conv_block
 * (resnet50.py:100,111: both read the block's input) -- as ONE launch where that measured faster than the two launches with their own
 * launch shapes (decided once per pair of shapes, kept in the tuning table), as radnet_conv_fwd(d1), radnet_conv_fwd(d2) otherwise
 * (different grids, 4-channel input, forced configs, autotuning off, RADNET_NO_FWD_PAIR=1).  Same results as the two calls. */
int radnet_conv_fwd_pair(radnet_ctx* ctx, const radnet_conv_desc* d1, const radnet_conv_desc* d2);
/* The back of a bottleneck block (resnet50.py:53-71, 104-128) as ONE launch: db = its 3x3 convolution (stride 1, 'same', 64 output
 * channels, ReLU), dc = its 1x1 expand on db's output (dc->x == db->y; + shortcut dc->addend, ReLU), da = the NEXT block's 1x1 reduce on
 * dc's output (da->x == dc->y, 64 output channels, ReLU) or 0.  A workgroup that holds [rows x 64] of the 3x3's output holds all of that
 * layer's channels for its rows and goes on with the pointwise convolutions for the same rows through LDS: db->y is NOT written
 * (frozen layers only -- nothing can be differentiated through the block afterwards), dc->y once, da->x is not read back.  Used where it
 * measured faster than the separate launches (decided once per shape, kept in the tuning table); radnet_conv_fwd(db), (dc), (da) -- which
 * do write db->y -- for every other geometry, forced configs, autotuning off, RADNET_NO_BNECK_FUSE=1. */
int radnet_conv_bottleneck(radnet_ctx* ctx, const radnet_conv_desc* db, const radnet_conv_desc* dc, const radnet_conv_desc* da);

/* out[n] (+)= sum_m g[m][n] * gscale[n]   (bias gradients) */
int radnet_colsum(radnet_ctx* ctx, const float* g, int32_t m, int32_t n, int32_t ld, const float* gscale,
                  float* out, int32_t accumulate);

/* ---- pooling / RoI crop-resize ------------------------------------------------------------- */
/* MaxPooling2D k x k / stride s 'valid' (resnet50.py:188: 3,2; VGG16 blocks: 2,2) */
int radnet_maxpool_fwd(radnet_ctx* ctx, const float* x, float* y, int32_t nb, int32_t h, int32_t w, int32_t c,
                       int32_t k, int32_t s);
/* RoiPoolingConv.call (RoiPoolingConv.py:48-88): int-cast RoI, clamped crop, TF1 legacy bilinear
 * resize to ps x ps.  rois: [r][4] fp32 (x,y,w,h) in feature-map units.  y: [r][ps][ps][c]. */
int radnet_roi_resize_fwd(radnet_ctx* ctx, const float* fmap, int32_t h, int32_t w, int32_t c, const float* rois,
                          int32_t r, int32_t ps, float* y);
/* gradient w.r.t. the feature map (scatter-add; dfmap must be zeroed by the caller) */
int radnet_roi_resize_bwd(radnet_ctx* ctx, const float* dy, int32_t h, int32_t w, int32_t c, const float* rois,
                          int32_t r, int32_t ps, float* dfmap);
/* TimeDistributed(AveragePooling2D((7,7))) + Flatten (resnet50.py:260-261): [r][hw][c] -> [r][c] */
int radnet_avgpool_fwd(radnet_ctx* ctx, const float* x, int32_t r, int32_t hw, int32_t c, float* y);
/* dx[r][p][c] = (y_act[r][p][c] > 0) ? dfeat[r][c] / hw : 0  (avg-pool backward fused with the ReLU mask) */
int radnet_avgpool_bwd_relu(radnet_ctx* ctx, const float* dfeat, const float* y_act, int32_t r, int32_t hw, int32_t c,
                            float* dx);

/* ---- classifier dense heads (resnet50.py:263-279) -------------------------------------------
 * w: [k][ldw] with the class columns first then the 4*(nc-1) regression columns; b: [nc+nreg].
 * out_cls = softmax(feat @ w[:, :nc] + b), out_regr = feat @ w[:, nc:] + b. */
int radnet_dense_heads_fwd(radnet_ctx* ctx, const float* feat, int32_t r, int32_t k, const float* w, int32_t ldw,
                           const float* b, int32_t nc, int32_t nreg, float* out_cls, float* out_regr);
/* dz: [r][nc+nreg] gradient w.r.t. the pre-softmax logits / regression outputs.
 * Writes (accumulate=0) or adds to (accumulate=1) dw [k][ldw] and db [ldw]; writes dfeat [r][k]. */
int radnet_dense_heads_bwd(radnet_ctx* ctx, const float* feat, const float* dz, int32_t r, int32_t k, const float* w,
                           int32_t ldw, int32_t nout, float* dw, float* db, float* dfeat, int32_t accumulate);

/* The tail of classifier_layer and its losses in one launch (csrc/head_tail.hip): avg-pool + both dense heads (+ softmax)
 * per RoI and -- with targets (y1 != 0) -- losses.py:69-95 and dz in the same launch.  The r RoIs come in `groups` equal groups, each its own reference step (own normalisers; losses
 * [groups][3] = cls, regr, accuracy); group_live[g] == 0: zero gradient rows, losses untouched.  scratch: device memory of
 * radnet_head_tail_scratch_bytes(r) bytes the caller zeroed ONCE (arrival counters that every launch leaves at zero, the
 * channel slices' partial sums, per-RoI loss terms). */
uint64_t radnet_head_tail_scratch_bytes(int32_t r);
int radnet_head_tail_fwd(radnet_ctx* ctx, const float* y5, int32_t r, int32_t hw, int32_t c, const float* w, int32_t ldw, const float* b,
                         int32_t nc, int32_t nreg, float* feat, float* p_cls, float* p_regr, const float* y1, const float* y2,
                         float* dz, float* losses, int32_t groups, const int32_t* group_live, void* scratch);

/* ---- losses (losses.py) ----------------------------------------------------------------------
 * RPN (losses.py:16-66): pred [m][ld_pred] holds the sigmoid class scores in columns [0,A) and the
 * regression outputs in [A,5A) (the fused head GEMM's layout); y_cls [m][2A], y_regr [m][8A].
 * losses[0..1] = (cls, regr).  dz [m][ld_dz]: gradient w.r.t. the PRE-activation outputs, same
 * column layout (sigmoid' folded in), columns >= 5A zeroed.
 * bce_mode 0: K.binary_crossentropy argument order as executed under Keras 2 (target=y_pred,
 * output=y_true) -- what the reference runs; 1: textbook BCE(y_true, clip(p)). */
int radnet_rpn_loss(radnet_ctx* ctx, const float* pred, int32_t ld_pred, const float* y_cls, const float* y_regr,
                    int32_t m, int32_t a, int32_t bce_mode, float* dz, int32_t ld_dz, float* losses,
                    double* scratch8);
/* Detector (losses.py:69-95): p_cls [r][nc] softmax, p_regr [r][nreg], y1 [r][nc], y2 [r][2*nreg].
 * losses[0..2] = (cls, regr, accuracy).  dz [r][nc+nreg] w.r.t. logits / regression outputs. */
int radnet_det_loss(radnet_ctx* ctx, const float* p_cls, const float* p_regr, const float* y1, const float* y2,
                    int32_t r, int32_t nc, int32_t nreg, float* dz, float* losses);

/* ---- optimizer: keras.optimizers.Adam over one flat arena (train.py:236-252) -----------------
 * zero_grad != 0: the gradient arena is cleared in the same pass (each value is read once and overwritten with 0),
 * so the next step's backward can accumulate into it without a separate memset. */
int radnet_adam_step(radnet_ctx* ctx, float* p, float* g, float* m, float* v, int64_t n, int32_t t, float lr,
                     float beta1, float beta2, float eps, float grad_scale, int32_t zero_grad);
/* The same step with the folded epilogue shifts of the layers whose biases live in p[bias_off, bias_off + bias_len) refreshed in the
 * same pass: shift[j] = scale[j] * p[bias_off + j] + t0[j] (the radnet_affine_vec call that otherwise follows the update of trainable
 * convs with a frozen FixedBatchNormalization behind them, FixedBatchNormalization.py:59-85).  bias_off, bias_len multiples of 4. */
int radnet_adam_step_affine(radnet_ctx* ctx, float* p, float* g, float* m, float* v, int64_t n, int32_t t, float lr, float beta1, float beta2,
                            float eps, float grad_scale, int32_t zero_grad, int64_t bias_off, int64_t bias_len, const float* scale,
                            const float* t0, float* shift);
/* radnet_adam_step_affine (shift may be 0: no bias re-fold) that ALSO rewrites the Winograd F(4x4,3x3) filter transforms of up to twelve
 * 3x3 kernels that live in the arena -- dense [3][3][c][n] at float offset `off` -- into u [36][c][n], from the weights it has just
 * updated: the arithmetic of radnet_winograd4_filter on the new weights, in the optimizer's pass (the classifier's three 3x3 convs run
 * their training forward on U; a transform launch per layer and update cost the step what the Winograd forward gives). */
typedef struct radnet_adam_wino {
  int64_t off;           /* float offset of the kernel in the arena (multiple of 4) */
  int32_t c, n;          /* input / output channels; the kernel is dense: ldw == n  */
  float* u;              /* [36][c][n]                                              */
} radnet_adam_wino;
int radnet_adam_step_fused(radnet_ctx* ctx, float* p, float* g, float* m, float* v, int64_t n, int32_t t, float lr, float beta1, float beta2,
                           float eps, float grad_scale, int32_t zero_grad, int64_t bias_off, int64_t bias_len, const float* scale,
                           const float* t0, float* shift, const radnet_adam_wino* layers, int32_t n_layers);

/* ---- proposal decode + greedy NMS (rpn.py:68-172, 299-344, 380-455), fp64 ---------------------
 * pred: fused head output [rows*cols][ld_pred] (scores in [0,A), regression in [A,5A)).
 * anchor_wh: host array [A][2] of anchor (w,h) in feature-map units.
 * out_boxes [max_boxes][4] int64 (x1,y1,x2,y2), out_probs [max_boxes] fp32, out_count[1] int32.
 * ws: device scratch of >= radnet_proposals_ws_bytes(rows*cols*A) bytes. */
uint64_t radnet_proposals_ws_bytes(int64_t n_anchors_total);
int radnet_rpn_to_roi(radnet_ctx* ctx, const float* pred, int32_t ld_pred, int32_t rows, int32_t cols, int32_t a,
                      const double* anchor_wh_host, double std_scaling, int32_t use_regr, double overlap_thresh,
                      int32_t max_boxes, int64_t* out_boxes, float* out_probs, int32_t* out_count, void* ws);
/* Generic greedy NMS on n boxes [n][4] fp64 xyxy + probs fp32 (rpn.py:380-455).  out_idx [max_boxes]
 * indices into the input.  Malformed boxes (x1>=x2 or y1>=y2) set out_count to -1 (the reference asserts). */
int radnet_nms(radnet_ctx* ctx, const double* boxes, const float* probs, int32_t n, double overlap_thresh,
               int32_t max_boxes, int32_t* out_idx, int32_t* out_count, void* ws);

/* ---- anchor targets: utils.calc_region_props before the random subsampling (utils.py:585-766) --
 * gt [g][4] fp64 (x1,y1,x2,y2) in SOURCE-image pixels, gt_is_bg [g].  anchor_sizes host [ns],
 * anchor_ratios host [nr][2].  Outputs (A = ns*nr): valid, overlap uint8 [A][fh][fw] (the NCHW order
 * the host subsampler indexes), regr fp64 [fh][fw][4A], best_anchor int32 [g][4] (jy,ix,ratio,size)
 * or -1, n_for_gt int32 [g].  scratch: >= 8*g bytes. */
int radnet_anchor_targets(radnet_ctx* ctx, const double* gt, const int32_t* gt_is_bg, int32_t g, int32_t width,
                          int32_t height, int32_t rw, int32_t rh, int32_t fw, int32_t fh, const double* anchor_sizes_host,
                          int32_t ns, const double* anchor_ratios_host, int32_t nr, double rpn_stride,
                          double max_overlap, uint8_t* valid, uint8_t* overlap, double* regr, int32_t* best_anchor,
                          int32_t* n_for_gt, void* scratch);
/* utils.py:815-816 + 475-478: y_cls [fh][fw][2A] = [valid || overlap], y_regr [fh][fw][8A] =
 * [repeat(overlap,4) || regr*std_scaling], fp32 NHWC. */
int radnet_anchor_targets_pack(radnet_ctx* ctx, const uint8_t* valid, const uint8_t* overlap, const double* regr,
                               int32_t fw, int32_t fh, int32_t a, double std_scaling, float* y_cls, float* y_regr);

/* ---- RoI labelling: rpn.calc_iou (rpn.py:176-296) ----------------------------------------------
 * rois int64 [n][4] xyxy (feature-map units), gt as above + gt_cls int32 [g] class index.
 * Per RoI: keep (0/1), cls (class index; bg for hard negatives), box int32 [4] (x,y,w,h),
 * t fp64 [4] = (sx*tx, sy*ty, sw*tw, sh*th) (zeros unless foreground), iou fp64. */
int radnet_roi_targets(radnet_ctx* ctx, const int64_t* rois, int32_t n, const double* gt, const int32_t* gt_cls, int32_t g,
                       int32_t width, int32_t height, int32_t rw, int32_t rh, double rpn_stride, double min_overlap,
                       double max_overlap, const double* regr_std_host4, int32_t bg_class, uint8_t* keep, int32_t* cls,
                       int32_t* box, double* t, double* iou, const int32_t* n_dev);
/* ^ cls is -1 for dropped RoIs.  n_dev (optional device int32): only the first min(*n_dev, n) rows are labelled,
 *   so the NMS count never has to travel to the host between the two kernels. */
/* Build the detector batch for `r` selected RoIs: rois_out fp32 [r][4], y1 [r][nc], y2 [r][8(nc-1)]. */
int radnet_roi_batch_pack(radnet_ctx* ctx, const int32_t* sel, int32_t r, const int32_t* cls, const int32_t* box,
                          const double* t, int32_t nc, int32_t bg_class, float* rois_out, float* y1, float* y2);

/* ---- host-only helper (HOST pointers, no GPU work) ------------------------------------------------
 * One round of NumPy's legacy RandomState.choice(n, size, replace=False, p=p) as used by utils.py:797,812.
 * live_p / live_idx [*n_live_io]: the still non-zero probabilities and their indices (increasing); the round does
 * cumsum + normalise over them (bit-identical to NumPy's cumsum over the full p: zeroed entries add 0.0), maps the
 * k uniforms `x` (drawn by the caller with np.random.random_sample, so the global MT19937 stream is consumed
 * exactly as NumPy would) through searchsorted(side='right'), appends first occurrences to found[n_found...] and
 * removes them from the live lists.  Returns the number appended.  cdf: scratch [n] doubles; sel: scratch [n] bytes. */
int64_t radnet_host_choice_round(double* live_p, int64_t* live_idx, int64_t* n_live_io, int64_t* found, int64_t n_found,
                                 const double* x, int64_t k, double* cdf, uint8_t* sel);

/* ---- small utilities ---------------------------------------------------------------------------- */
/* uint8 BGR HWC image -> fp32 NHWC with `cpad` channels (zeros beyond 3), minus the caffe BGR means
 * (RADNet.py:83-87 / utils.py:468-472 with keras 'caffe' preprocess_input). */
int radnet_preprocess_bgr(radnet_ctx* ctx, const uint8_t* img, int32_t h, int32_t w, int32_t cpad, float* out);
/* dst[0..bytes) = src[0..bytes) by a kernel on the context's stream; either side may be pinned host memory (hipHostMalloc /
 * torch pin_memory: mapped into the device's address space).  The step's small transfers between host and device (image panel
 * in, anchor label maps out and back, RoI class codes out: utils.py:777-813, train.py:300-330 run on the host in the reference)
 * go through this instead of hipMemcpyAsync: an asynchronous copy enqueued on a stream with a few layer programs in arrears
 * was measured to block the enqueuing host thread for 7-11 ms (tools/fill_drain_probe.py); a kernel launch never did. */
int radnet_copy_bytes(radnet_ctx* ctx, void* dst, const void* src, uint64_t bytes);
/* cv2.resize(img, (dw, dh), interpolation=INTER_CUBIC) for uint8 HWC images (RADNet.py:72, utils.py:442-446):
 * a = -0.75 bicubic, half-pixel centres, replicated borders, OpenCV's 11-bit fixed-point arithmetic.
 * Parity unpinned (OpenCV absent offline). */
int radnet_resize_bicubic_u8(radnet_ctx* ctx, const uint8_t* src, int32_t sh, int32_t sw, uint8_t* dst, int32_t dh,
                             int32_t dw, int32_t channels);
/* cv2.warpAffine(src, M, (dw, dh)) with its defaults (INTER_LINEAR, BORDER_CONSTANT 0) for uint8 HWC tiles: the +-3 degree rotation
 * and the shear of the train-time augmentation (augmentation.py:158-271).  The caller inverts M and passes the inverse map's
 * per-column and per-row terms in 10-bit fixed point, rounded in float64 as OpenCV does: col_tab = {adelta[dw], bdelta[dw]},
 * row_tab = {x0[dh], y0[dh]} (device int32 arrays; faster_rcnn/augmentation.py:warp_tables builds them).  Bit-identical to that
 * module's NumPy restatement; against OpenCV itself unpinned (absent here). */
int radnet_warp_affine_u8(radnet_ctx* ctx, const uint8_t* src, int32_t sh, int32_t sw, int32_t channels, uint8_t* dst, int32_t dh,
                          int32_t dw, const int32_t* col_tab, const int32_t* row_tab);
int radnet_fill_zero(radnet_ctx* ctx, void* p, uint64_t bytes);
/* y = x * alpha (n floats); used to average gradients after all-reduce */
int radnet_scale(radnet_ctx* ctx, float* x, int64_t n, float alpha);
/* out[i] = a[i]*b[i] + c[i]: refreshes the folded epilogue shift (BN scale * conv bias + BN shift,
 * FixedBatchNormalization.py:59-85) of the trainable head convs after an optimizer step. */
int radnet_affine_vec(radnet_ctx* ctx, float* out, const float* a, const float* b, const float* c, int64_t n);
/* Input gradient of a stride-s 1x1 convolution (resnet50.py:100,111: the first 1x1 and the shortcut of every
 * conv_block).  The GEMM runs on the compact grid -- radnet_conv_dgrad with a descriptor whose input geometry is
 * the OUTPUT grid (h = oh, w = ow, stride 1) -- and this pass places its rows at (oh*s, ow*s) of the full
 * [nb][h][w][c] gradient, zeros elsewhere, then applies the producer's ReLU mask (mask > 0) if given. */
int radnet_scatter_strided(radnet_ctx* ctx, const float* src, int32_t nb, int32_t oh, int32_t ow, int32_t c, int32_t stride,
                           int32_t h, int32_t w, const float* mask, float* dst);
/* g[i] = act[i] > 0 ? g[i] : 0 -- ReLU backward where it cannot ride a GEMM epilogue (VGG16 head, vgg16.py:98-101) */
int radnet_relu_mask(radnet_ctx* ctx, float* g, const float* act, int64_t n);

/* ==== layer programs and composed entry points (rpn_forward, train_step, predict_tile, allreduce_grads) ====
 * A layer program is a static array of launches the host scheduler builds once per input size from the reference's graph
 * (resnet50.py:150-281 nn_base / classifier_layer, rpn.py:12-66 rpn_layer, and their backward programs); radnet_program_run
 * enqueues it on the context's stream.  No allocation, no synchronisation: running it on a capturing stream records it into
 * a hipGraph.  Argument slots per kind (i = integers, p = device pointers):
 *   CONV_FWD / CONV_DGRAD / CONV_WGRAD   conv
 *   CONV_BWD     conv (radnet_conv_bwd: weight gradient + data gradient of the layer);  NOP: skipped
 *   CONV_FWD_PAIR conv = first convolution, the NEXT op's conv = second (its kind is NOP): radnet_conv_fwd_pair
 *   CONV_BNECK   conv = db, the next op's conv = dc, i[0] = 1: the op after that holds da (both NOP slots): radnet_conv_bottleneck
 *   MAXPOOL      p: x, y                     i: nb, h, w, c, k, s
 *   COLSUM       p: g, gscale|0, out         i: m, n, ld, accumulate
 *   WINO         p: x, v, u, m, scale|0, shift|0, y      i: nb, h, w, c, n, tiles, act, ldy, form   (radnet_winograd_input + 16
 *   WINO_REUSE   same, v already holds this input's transform                             GEMMs + radnet_winograd_output;
 *                                                                         form 4: the radnet_winograd4_* transforms, 36 GEMMs)
 *   WINO_WGRAD   p: dy, v, dz, du, dw, gscale|0   i: nb, h, w, c, n, ld_dy, tiles, ldw, accumulate mode (as radnet_conv_desc), form
 *   SCATTER      p: src, mask|0, dst         i: nb, oh, ow, c, stride, h, w
 *   FILL0        p: dst                      i: bytes (low 32 bits), bytes (high 32 bits)
 *   RELU_MASK    p: g, act                   i: n (low), n (high)
 *   ROI_BWD      p: dy, rois, dfmap          i: h, w, c, r, ps
 *   CHAIN        p: radnet_chain*             (radnet_chain_run: a run of CONV_FWD / WINO ops as one persistent launch) */
enum {
  RADNET_OP_CONV_FWD = 1, RADNET_OP_CONV_DGRAD = 2, RADNET_OP_CONV_WGRAD = 3, RADNET_OP_MAXPOOL = 4, RADNET_OP_COLSUM = 5,
  RADNET_OP_WINO = 6, RADNET_OP_WINO_REUSE = 7, RADNET_OP_WINO_WGRAD = 8, RADNET_OP_SCATTER = 9, RADNET_OP_FILL0 = 10,
  RADNET_OP_RELU_MASK = 11, RADNET_OP_ROI_BWD = 12, RADNET_OP_CONV_BWD = 13, RADNET_OP_CHAIN = 14, RADNET_OP_CONV_FWD_PAIR = 15,
  RADNET_OP_CONV_BNECK = 16,
  RADNET_OP_NOP = 0
};
typedef struct radnet_op {
  int32_t kind;
  int32_t i[11];
  const void* p[8];
  radnet_conv_desc conv;
} radnet_op;
int radnet_program_run(radnet_ctx* ctx, const radnet_op* ops, int32_t n_ops);

/* ---- a run of dependent layers as ONE persistent launch ("chain"; resnet50.py:150-228 nn_base, stages 2-4) --------------
 * radnet_chain_build turns a list of CONV_FWD (channels a multiple of 32) and WINO (form 4) ops -- the same radnet_op[] that
 * radnet_program_run would launch one by one -- into a list of work items (output tiles, transform blocks) in dependency
 * order plus arrival counters; radnet_chain_run launches `workgroups` persistent workgroups (0: two per CU; at most 4 per CU)
 * among which the items are dealt statically (workgroup b: items b, b + grid, ...) and which start each item as soon as the
 * blocks it reads are complete.  The deal is deadlock-free only while EVERY workgroup of the grid is resident -- of every chain
 * running at the same time: callers that run chains side by side divide the 4 x 256 slots between them (the engine does).  Same kernels' code, same arithmetic per output element
 * as the launches it replaces for the convs (64x64 tiles; K-split layers add their slices in slice order); what goes away
 * is the gap between dependent launches, their lockstep prologue / epilogue phases and their tails.  A narrow
 * grid leaves CU slots to launches on other streams.  RADNET_ERR_UNSUPPORTED for an op the chain cannot run: the caller
 * keeps radnet_program_run.  The chain holds device memory of its own (items, counters, K-split slabs): build it outside
 * stream capture; radnet_chain_run allocates nothing and can be captured.  radnet_chain_status synchronises the stream;
 * last_error != 0: a workgroup waited 1 s for an input block that never completed (1 + item index; the launch still
 * drains, its outputs are INVALID).  The first such error is sticky and also lands in a mapped host word: radnet_chain_error
 * reads it without synchronising (0 = none so far; a launch still in flight may yet fail), and radnet_chain_run refuses
 * (RADNET_ERR_HIP) to launch a chain that has failed before.  Every tensor of the list must be written once and never after it
 * has been read (no buffer re-use inside a chain): radnet_chain_build / radnet_chain_check refuse such a list. */
typedef struct radnet_chain radnet_chain;
int radnet_chain_build(radnet_ctx* ctx, const radnet_op* ops, int32_t n_ops, int32_t workgroups, radnet_chain** out);
int radnet_chain_run(radnet_ctx* ctx, radnet_chain* chain);
int radnet_chain_status(radnet_ctx* ctx, radnet_chain* chain, int32_t* last_error, int32_t* runs, int32_t* n_items, int32_t* n_stages,
                        double* flops_executed, double* flops_algorithmic);
uint32_t radnet_chain_error(radnet_chain* chain);
void radnet_chain_destroy(radnet_chain* chain);
/* Diagnosis while a chain launch runs (uses a stream of its own): out[0..7] = {next item, workgroups gone, error, first
 * error, runs, ...}; with item >= 0, out[8..19] = its record {stage, bx, by, bz, dep0 first, dep0 count, dep1 first, dep1 count,
 * signal 0, signal 1, ..} and out[20 + 2k], out[21 + 2k] = current value / expected value of the k-th counter it waits for.
 * Returns the number of such pairs (or a negative error); out_words >= 148. */
int radnet_chain_peek(radnet_chain* chain, int32_t item, uint32_t* out, int32_t out_words);
/* Host-only (no device, no context): plans the work-item list of `ops` as radnet_chain_build would and checks that it can
 * run in list order -- every item finds its input blocks completed by earlier items, every counter reaches exactly the
 * count its waiters expect -- which is what makes the launch deadlock-free while its workgroups are all resident.  first_bad_item:
 * -1, or the offending item (n_items: a counter that never reaches its count).  The pointers in `ops` are used as
 * identities only.  radnet_chain_build runs the same check and refuses a list that fails it. */
int radnet_chain_check(const radnet_op* ops, int32_t n_ops, int32_t* n_items, int32_t* n_stages, int32_t* n_counters, int32_t* first_bad_item,
                       int32_t* bad_counter, char* err, int32_t err_len);

/* model_rpn.predict (RADNet.py:552; train.py:291): nn_base program, then rpn_layer program (outputs where the programs'
 * descriptors point: the fused head matrix [fh*fw][ld] with the sigmoid scores in columns [0,A), regressions in [A,5A)). */
int radnet_rpn_forward(radnet_ctx* ctx, const radnet_op* base_ops, int32_t n_base, const radnet_op* rpn_ops, int32_t n_rpn);

/* classifier_layer on n_rois RoIs of one feature map (resnet50.py:231-281): RoI crop-resize, stage-5 program, avg-pool,
 * dense heads.  rois [n_rois][4] fp32 (x,y,w,h); outputs p_cls [n_rois][nc] (softmax), p_regr [n_rois][nreg]. */
typedef struct radnet_head_desc {
  const float* fmap; int32_t fh, fw, fc;
  const float* rois; int32_t n_rois, pool; float* pooled;
  const radnet_op* fwd_ops; int32_t n_fwd;
  const float* y5; int32_t hw, feat_c; float* feat;
  const float* dense_w; int32_t dense_ld; const float* dense_b; int32_t nc, nreg;
  float* p_cls; float* p_regr;
  void* tail_scratch;      /* radnet_head_tail_scratch_bytes(n_rois) bytes, zeroed once by the caller */
} radnet_head_desc;

/* One tile of RADNet.predict up to the classifier outputs (RADNet.py:520-600): preprocess (img_u8 != 0: uint8 BGR [h][w][3]
 * -> x fp32 [h][w][4]), base + RPN programs, rpn.rpn_to_roi (decode + NMS, <= max_boxes RoIs in R / Rn), then -- when head is
 * set -- the first head->n_rois proposals (padded with copies of the first, RADNet.py:115-122) through classifier_layer. */
typedef struct radnet_tile_desc {
  const uint8_t* img_u8; int32_t h, w; float* x;
  const radnet_op* base_ops; int32_t n_base;
  const radnet_op* rpn_ops; int32_t n_rpn;
  const float* pred; int32_t ld_pred, fh, fw, a;
  const double* anchor_wh_host; double std_scaling, overlap_thresh; int32_t max_boxes;
  int64_t* R; float* Rp; int32_t* Rn; void* prop_ws;
  const radnet_head_desc* head;
} radnet_tile_desc;
int radnet_predict_tile(radnet_ctx* ctx, const radnet_tile_desc* t);

/* One flat optimizer arena (train.py:236-252: one Adam per model). */
typedef struct radnet_adam_desc {
  float* p; float* g; float* m; float* v; int64_t n; int32_t t; float lr;
} radnet_adam_desc;

/* The two places where a reference iteration draws from NumPy's GLOBAL random stream stay on the host, in the caller's RNG:
 *   subsample_anchors  utils.py:785-813: valid / overlap uint8 [a][fh][fw] (host); edits valid in place; returns n_pos, or < 0
 *                      when the labeller fails as the reference's does (the sample is skipped, utils.py:461-465);
 *   select_rois        train.py:93-129: cls[n] class index per proposal (-1 = dropped by calc_iou); writes n_rois indices into
 *                      the proposal list to sel; returns n_rois, or 0 when no proposal was kept (head step skipped). */
typedef struct radnet_host_hooks {
  void* user;
  int32_t (*subsample_anchors)(void* user, uint8_t* valid, const uint8_t* overlap, int32_t a, int32_t fh, int32_t fw);
  int32_t (*select_rois)(void* user, const int32_t* cls, int32_t n, int32_t* sel, int32_t n_rois);
} radnet_host_hooks;

/* One reference training iteration on one image (train.py:288-402), synchronous, on the context's stream:
 * anchor targets -> base + RPN forward -> [hook] -> RPN losses / backward -> [all-reduce] -> Adam #1 -> re-prediction ->
 * proposals -> RoI labelling -> [hook] -> classifier forward / losses / backward -> [all-reduce] -> Adam #2.  The caller owns
 * every buffer (h_* are pinned host mirrors) and increments the optimizers' t before the call.  world > 1: the gradient
 * arenas are summed over the communicator of radnet_comm_init and the update uses 1/world (one image per rank).
 * losses5 = (rpn_cls, rpn_regr, det_cls, det_regr, det_acc); took_head_step = 1, 0 when the classifier step was skipped
 * (no proposal kept), -1 when the labeller hook dropped the image (nothing was trained, no optimizer step). */
typedef struct radnet_train_desc {
  const uint8_t* img_u8; int32_t h, w; float* x;
  const double* gt; const int32_t* gt_is_bg; const int32_t* gt_cls; int32_t g, width, height;
  const double* anchor_sizes_host; int32_t ns; const double* anchor_ratios_host; int32_t nr;
  double rpn_stride, rpn_max_overlap, std_scaling;
  uint8_t* valid; uint8_t* overlap; double* regr; int32_t* best_anchor; int32_t* n_for_gt; void* at_scratch;
  uint8_t* h_valid; uint8_t* h_overlap; float* y_cls; float* y_regr;
  const radnet_op* base_ops; int32_t n_base;
  const radnet_op* rpn_fwd_ops; int32_t n_rpn_fwd;
  const radnet_op* rpn_bwd_ops; int32_t n_rpn_bwd;
  const radnet_op* rpn_refwd_ops; int32_t n_rpn_refwd;
  float* pred; float* dz; int32_t ld_pred, fh, fw, a, bce_mode; double* loss_scratch8; float* rpn_losses;
  radnet_adam_desc rpn_opt, head_opt; int32_t world;
  const float* wino_w; int32_t wino_c, wino_n, wino_ldw, wino_form; float* wino_u;   /* rpn_conv1 filter re-transform after Adam #1; form 4 = F(4x4) */
  const double* anchor_wh_host; double overlap_thresh; int32_t max_boxes;
  int64_t* R; float* Rp; int32_t* Rn; void* prop_ws;
  int32_t rw, rh; double min_overlap, max_overlap; const double* regr_std_host4; int32_t bg_class;
  uint8_t* keep; int32_t* roi_cls; int32_t* roi_box; double* roi_t; double* roi_iou;
  int32_t* h_roi_cls; int32_t* h_n; int32_t* sel; int32_t* h_sel;
  const radnet_head_desc* head; float* y1; float* y2;
  float* head_dz; float* det_losses; float* dense_dw; float* dense_db; float* dfeat; float* g_last;
  const radnet_op* head_bwd_ops; int32_t n_head_bwd;
  float* head_shift; const float* head_scale; const float* head_bias; const float* head_t0; int64_t head_bias_len;
  void* tail_scratch;      /* unused (the head descriptor carries the scratch); kept for layout stability */
  const radnet_adam_wino* head_wino; int32_t n_head_wino;   /* classifier 3x3 kernels whose training forward runs on Winograd filters: Adam #2 rewrites them (radnet_adam_step_fused) */
} radnet_train_desc;
int radnet_train_step(radnet_ctx* ctx, const radnet_train_desc* d, const radnet_host_hooks* hooks, float* losses5,
                      int32_t* took_head_step);

/* ---- data-parallel gradient exchange over RCCL / xGMI ---------------------------------------------------
 * One communicator per context: rank 0 draws the id (radnet_comm_unique_id) and hands it to the other ranks through whatever
 * rendezvous the job has (torch.distributed's store, a file, MPI); every rank then calls radnet_comm_init.  The collective
 * is enqueued on the context's stream, in place, fp32 sum.  RCCL is bound at run time: RADNET_ERR_UNSUPPORTED without it. */
int radnet_comm_unique_id(char out128[128]);
int radnet_comm_init(radnet_ctx* ctx, int32_t world, int32_t rank, const char id128[128]);
int radnet_comm_destroy(radnet_ctx* ctx);
int radnet_allreduce_grads(radnet_ctx* ctx, float* grads, int64_t count);
/* Exchanges issued on this context so far (calls, fp32 elements): lets a caller check that every rank of a data-parallel job
 * takes part in the same sequence of collectives, whatever path its own image took through radnet_train_step. */
int radnet_comm_stats(radnet_ctx* ctx, int64_t* calls, int64_t* elements);

#ifdef __cplusplus
}
#endif
#endif /* RADNET_HIP_H */
