"""ORACLE (test infrastructure, not product): NumPy restatement of the reference's Keras/TF
graph for the Faster R-CNN hot path -- forward, explicit backward, losses and Adam.

PARITY UNPINNED: TensorFlow/Keras are absent from this image and the reference ships no test
or golden vector for this half, so nothing here could be checked against the
reference's own execution.  It follows the cited source lines and the TF1/Keras2 semantics
listed in SURVEY.md A.3, and is cross-checked against torch CPU ops (tests/test_oracle_dense.py)
as an independent second implementation.

Layout: activations NHWC, conv kernels HWIO (kh,kw,Cin,Cout) and dense kernels (in,out) -- the
Keras layouts, so weights keyed by Keras layer names drop in.  dtype follows the inputs
(float32 for parity runs, float64 for gradient checks).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import numpy as np

BN_EPS = 1e-3          # FixedBatchNormalization.py:8
LOSS_EPS = 1e-4        # losses.py:14
KERAS_EPS = 1e-7       # keras.backend.epsilon()


# ----------------------------------------------------------------------------------------
# primitive ops
# ----------------------------------------------------------------------------------------
def conv2d(x, w, b, stride=1, pad=(0, 0, 0, 0)):
    """x (N,H,W,C), w (kh,kw,C,Co), b (Co,) or None.  pad = (top, left, bottom, right)."""
    kh, kw, C, Co = w.shape
    pt, pl, pb, pr = pad
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0))) if any(pad) else x
    N, H, W, _ = xp.shape
    OH = (H - kh) // stride + 1
    OW = (W - kw) // stride + 1
    out = np.zeros((N * OH * OW, Co), dtype=x.dtype)
    for i in range(kh):
        for j in range(kw):
            patch = xp[:, i:i + stride * (OH - 1) + 1:stride, j:j + stride * (OW - 1) + 1:stride, :]
            out += patch.reshape(-1, C) @ w[i, j]
    if b is not None:
        out += b
    return out.reshape(N, OH, OW, Co)


def conv2d_bwd(x, w, dy, stride=1, pad=(0, 0, 0, 0), need_dx=True):
    """Returns (dx or None, dw, db) for y = conv2d(x, w, b)."""
    kh, kw, C, Co = w.shape
    pt, pl, pb, pr = pad
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0))) if any(pad) else x
    N, OH, OW, _ = dy.shape
    dy2 = dy.reshape(-1, Co)
    dw = np.zeros_like(w)
    dxp = np.zeros_like(xp) if need_dx else None
    for i in range(kh):
        for j in range(kw):
            sl = (slice(None), slice(i, i + stride * (OH - 1) + 1, stride), slice(j, j + stride * (OW - 1) + 1, stride), slice(None))
            dw[i, j] = xp[sl].reshape(-1, C).T @ dy2
            if need_dx:
                dxp[sl] += (dy2 @ w[i, j].T).reshape(N, OH, OW, C)
    db = dy2.sum(0)
    dx = None
    if need_dx:
        H, W = xp.shape[1:3]
        dx = dxp[:, pt:H - pb, pl:W - pr, :]
    return dx, dw, db


def bn_affine(bn):
    """FixedBatchNormalization.py:59-85: y = gamma*(x-mean)/sqrt(var+1e-3)+beta, the weight named
    running_std being used as the variance.  Returns per-channel (scale, shift)."""
    s = bn["gamma"] / np.sqrt(bn["var"] + BN_EPS)
    return s, bn["beta"] - bn["mean"] * s


def maxpool_3x3_s2(x):
    """MaxPooling2D((3,3), strides=(2,2)), 'valid' (resnet50.py:188)."""
    N, H, W, C = x.shape
    OH, OW = (H - 3) // 2 + 1, (W - 3) // 2 + 1
    out = np.full((N, OH, OW, C), -np.inf, dtype=x.dtype)
    for i in range(3):
        for j in range(3):
            out = np.maximum(out, x[:, i:i + 2 * (OH - 1) + 1:2, j:j + 2 * (OW - 1) + 1:2, :])
    return out


def maxpool_2x2_s2(x):
    """VGG16 block pools (keras.applications VGG16): 2x2 stride 2 'valid'."""
    N, H, W, C = x.shape
    OH, OW = H // 2, W // 2
    v = x[:, :OH * 2, :OW * 2, :].reshape(N, OH, 2, OW, 2, C)
    return v.max(axis=(2, 4))


def roi_crop_resize(fmap, rois, ps):
    """RoiPoolingConv.py:48-88: per RoI (x,y,w,h) cast to int32 (truncation), crop with slice
    clamping, then TF1 tf.image.resize_images bilinear, align_corners=False, legacy mapping
    src = dst*(in/out) in float32; lerp top/bottom rows then columns (TF resize_bilinear_op).
    fmap (1,H,W,C); rois (R,4).  Returns (R,ps,ps,C)."""
    _, H, W, C = fmap.shape
    R = rois.shape[0]
    out = np.zeros((R, ps, ps, C), dtype=fmap.dtype)
    f32 = np.float32
    for r in range(R):
        x, y, w, h = (int(np.trunc(v)) for v in rois[r])
        y0, y1 = min(max(y, 0), H), min(max(y + h, 0), H)
        x0, x1 = min(max(x, 0), W), min(max(x + w, 0), W)
        crop = fmap[0, y0:y1, x0:x1, :]
        ch, cw = crop.shape[:2]
        hs = f32(ch) / f32(ps)
        ws = f32(cw) / f32(ps)
        for oy in range(ps):
            sy = f32(oy) * hs
            ylo = int(np.floor(sy)); yhi = min(ylo + 1, ch - 1); ly = fmap.dtype.type(sy - f32(ylo))
            for ox in range(ps):
                sx = f32(ox) * ws
                xlo = int(np.floor(sx)); xhi = min(xlo + 1, cw - 1); lx = fmap.dtype.type(sx - f32(xlo))
                tl, tr, bl, br = crop[ylo, xlo], crop[ylo, xhi], crop[yhi, xlo], crop[yhi, xhi]
                top = tl + (tr - tl) * lx
                bot = bl + (br - bl) * lx
                out[r, oy, ox] = top + (bot - top) * ly
    return out


def roi_crop_resize_bwd(fmap_shape, rois, ps, dout):
    """Gradient of roi_crop_resize w.r.t. the feature map (scatter-add of the 4 bilinear taps)."""
    _, H, W, C = fmap_shape
    dF = np.zeros(fmap_shape, dtype=dout.dtype)
    f32 = np.float32
    for r in range(rois.shape[0]):
        x, y, w, h = (int(np.trunc(v)) for v in rois[r])
        y0, y1 = min(max(y, 0), H), min(max(y + h, 0), H)
        x0, x1 = min(max(x, 0), W), min(max(x + w, 0), W)
        ch, cw = y1 - y0, x1 - x0
        if ch <= 0 or cw <= 0:          # empty crop: the forward pass produced zeros that depend on nothing
            continue
        hs = f32(ch) / f32(ps); ws = f32(cw) / f32(ps)
        for oy in range(ps):
            sy = f32(oy) * hs
            ylo = int(np.floor(sy)); yhi = min(ylo + 1, ch - 1); ly = dout.dtype.type(sy - f32(ylo))
            for ox in range(ps):
                sx = f32(ox) * ws
                xlo = int(np.floor(sx)); xhi = min(xlo + 1, cw - 1); lx = dout.dtype.type(sx - f32(xlo))
                g = dout[r, oy, ox]
                dF[0, y0 + ylo, x0 + xlo] += g * (1 - ly) * (1 - lx)
                dF[0, y0 + ylo, x0 + xhi] += g * (1 - ly) * lx
                dF[0, y0 + yhi, x0 + xlo] += g * ly * (1 - lx)
                dF[0, y0 + yhi, x0 + xhi] += g * ly * lx
    return dF


def sigmoid(z):
    return (1.0 / (1.0 + np.exp(-z))).astype(z.dtype)


def softmax(z):
    e = np.exp(z - z.max(-1, keepdims=True))
    return (e / e.sum(-1, keepdims=True)).astype(z.dtype)


# ----------------------------------------------------------------------------------------
# losses (losses.py) -- each returns (value, gradient w.r.t. the network output it consumes)
# ----------------------------------------------------------------------------------------
def _bce_logits_swapped(t):
    """Keras-2 K.binary_crossentropy(target, output) as the reference CALLS it (losses.py:64 passes
    y_pred first): `output` = y_true overlap flag, clipped to [1e-7, 1-1e-7] in fp32 and turned into
    a logit; `target` = the sigmoid prediction.  Returns the logit per element (two possible values)."""
    f = np.float32
    lo = f(KERAS_EPS); hi = f(1.0) - f(KERAS_EPS)
    o = np.clip(t.astype(f), lo, hi)
    return np.log(o / (f(1.0) - o))


def rpn_loss_cls(y_true, p, A, keras2_arg_order=True):
    """losses.py:47-66.  y_true (1,H,W,2A) = [valid || overlap]; p (1,H,W,A) sigmoid output.
    Returns (loss, dL/dp).

    keras2_arg_order=True reproduces what the reference executes under the Keras 2.x API it is
    written against: K.binary_crossentropy(target=y_pred, output=y_true) -> per element
    max(l,0) - l*p + log1p(exp(-|l|)) with l = logit(clip(y_true)).  False gives the textbook
    BCE(y_true, p) with p clipped to [1e-7, 1-1e-7] (the Keras-1 argument order)."""
    dt = p.dtype
    valid = y_true[..., :A].astype(dt)
    t = y_true[..., A:].astype(dt)
    den = (LOSS_EPS + valid).sum(dtype=dt)
    if keras2_arg_order:
        l = _bce_logits_swapped(t).astype(dt)
        ce = np.maximum(l, 0) - l * p + np.log1p(np.exp(-np.abs(l)))
        dce = -l
    else:
        pc = np.clip(p, dt.type(KERAS_EPS), dt.type(1.0) - dt.type(KERAS_EPS))
        z = np.log(pc / (1 - pc))
        ce = np.maximum(z, 0) - z * t + np.log1p(np.exp(-np.abs(z)))
        inr = (p >= dt.type(KERAS_EPS)) & (p <= dt.type(1.0) - dt.type(KERAS_EPS))
        dce = np.where(inr, (pc - t) / (pc * (1 - pc)), 0).astype(dt)
    loss = (valid * ce).sum(dtype=dt) / den
    return loss, (valid * dce / den).astype(dt)


def smooth_l1_masked(y_true, pred, n):
    """losses.py:16-44 and 69-90: y_true = [mask (n) || target (n)] on the last axis."""
    dt = pred.dtype
    mask = y_true[..., :n].astype(dt)
    x = y_true[..., n:].astype(dt) - pred
    ax = np.abs(x)
    small = (ax <= 1.0).astype(dt)
    den = (LOSS_EPS + mask).sum(dtype=dt)
    val = (mask * (small * (0.5 * x * x) + (1 - small) * (ax - 0.5))).sum(dtype=dt) / den
    grad = -(mask * (small * x + (1 - small) * np.sign(x))) / den
    return val, grad.astype(dt)


def class_loss_cls(y_true, p):
    """losses.py:93-95: mean over RoIs of Keras categorical_crossentropy(y_true[0], y_pred[0])
    (renormalise, clip to [1e-7,1-1e-7], -sum t log p).  Returns (loss, dL/dp)."""
    dt = p.dtype
    t = y_true[0].astype(dt)
    q = p[0]
    S = q.sum(-1, keepdims=True)
    o = q / S
    oc = np.clip(o, dt.type(KERAS_EPS), dt.type(1.0) - dt.type(KERAS_EPS))
    R = q.shape[0]
    loss = (-(t * np.log(oc)).sum(-1)).mean(dtype=dt)
    inr = ((o >= dt.type(KERAS_EPS)) & (o <= dt.type(1.0) - dt.type(KERAS_EPS))).astype(dt)
    a = -(t * inr) / oc                     # dL_r/do_k
    dq = (a - (a * o).sum(-1, keepdims=True)) / S
    return loss, (dq / R)[None].astype(dt)


def categorical_accuracy(y_true, p):
    return float((y_true[0].argmax(-1) == p[0].argmax(-1)).mean())


def adam_step(p, g, m, v, t, lr, b1=0.9, b2=0.999, eps=KERAS_EPS):
    """keras.optimizers.Adam (Keras 2, no decay/amsgrad): t counts from 1."""
    lr_t = lr * np.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
    m[...] = b1 * m + (1.0 - b1) * g
    v[...] = b2 * v + (1.0 - b2) * g * g
    p[...] = p - p.dtype.type(lr_t) * m / (np.sqrt(v) + p.dtype.type(eps))


# ----------------------------------------------------------------------------------------
# ResNet50 Faster R-CNN graph (base_models/resnet50.py, rpn.py:12-66)
# ----------------------------------------------------------------------------------------
RES_STAGES = (  # stage, blocks, (f1,f2,f3), stride of the first block
    (2, "abc", (64, 64, 256), 1),
    (3, "abcd", (128, 128, 512), 2),
    (4, "abcdef", (256, 256, 1024), 2),
)
HEAD_STAGE = (5, "abc", (512, 512, 2048), 2)


def resnet50_conv_specs(n_anchors=12, n_classes=7, include_head=True):
    """[(conv_name, bn_name or None, kh, cin, cout)] in execution order, Keras layer names
    (resnet50.py:46-47,72-73,185-186; rpn.py:47,55,63; resnet50.py:269,278)."""
    specs = [("conv1", "bn_conv1", 7, 3, 64)]
    cin = 64
    stages = RES_STAGES + ((HEAD_STAGE,) if include_head else ())
    for st, blocks, (f1, f2, f3), _ in stages:
        if st == 5:
            cin = 1024
        for bl in blocks:
            base = "res%d%s_branch" % (st, bl)
            bnb = "bn%d%s_branch" % (st, bl)
            specs += [(base + "2a", bnb + "2a", 1, cin, f1), (base + "2b", bnb + "2b", 3, f1, f2), (base + "2c", bnb + "2c", 1, f2, f3)]
            if bl == "a":
                specs.append((base + "1", bnb + "1", 1, cin, f3))
            cin = f3
    specs += [("rpn_conv1", None, 3, 1024, 512), ("rpn_out_class", None, 1, 512, n_anchors), ("rpn_out_regress", None, 1, 512, 4 * n_anchors)]
    return specs


def init_params(seed=3, n_anchors=12, n_classes=7, dtype=np.float32, head_scale=1.0):
    """Seeded synthetic weights (no pretrained files exist offline): He-normal convs,
    BN gamma in [0.5,1.5], small beta/mean, var in [0.5,1.5] so activations stay O(1)."""
    rs = np.random.RandomState(seed)
    P = {}
    for name, bn, k, cin, cout in resnet50_conv_specs(n_anchors, n_classes):
        std = np.sqrt(2.0 / (k * k * cin))
        if bn is not None and name.endswith("2c"):
            std *= 0.25                     # keep the residual sum from growing through 16 blocks
        if name == "conv1":
            std /= 70.0                     # mean-subtracted pixels have std ~70: bring conv1 output to O(1)
        P[name] = {"kernel": (rs.standard_normal((k, k, cin, cout)) * std).astype(dtype),
                   "bias": (rs.standard_normal(cout) * 0.05).astype(dtype)}
        if bn is not None:
            P[bn] = {"gamma": rs.uniform(0.5, 1.5, cout).astype(dtype), "beta": (rs.standard_normal(cout) * 0.1).astype(dtype),
                     "mean": (rs.standard_normal(cout) * 0.1).astype(dtype), "var": rs.uniform(0.5, 1.5, cout).astype(dtype)}
    P["rpn_conv1"]["kernel"] *= dtype(0.1)         # F is O(10) after 13 residual adds; bring the RPN hidden layer to O(1)
    P["rpn_out_class"]["kernel"] *= dtype(0.5)
    P["rpn_out_regress"]["kernel"] *= dtype(0.5)
    P["dense_class_%d" % n_classes] = {"kernel": (rs.standard_normal((2048, n_classes)) * 0.004 * head_scale).astype(dtype),
                                       "bias": (rs.standard_normal(n_classes) * 0.01).astype(dtype)}
    P["dense_regress_%d" % n_classes] = {"kernel": (rs.standard_normal((2048, 4 * (n_classes - 1))) * 0.004 * head_scale).astype(dtype),
                                         "bias": (rs.standard_normal(4 * (n_classes - 1)) * 0.01).astype(dtype)}
    return P


def _cbr(P, x, conv, bn, stride=1, pad=(0, 0, 0, 0), relu=True, add=None):
    """conv(+bias) -> frozen BN -> (+add) -> relu.  Returns (y, cache)."""
    z = conv2d(x, P[conv]["kernel"], P[conv]["bias"], stride, pad)
    s, t = bn_affine(P[bn])
    y = z * s + t
    if add is not None:
        y = y + add
    if relu:
        y = np.maximum(y, 0)
    return y.astype(x.dtype), dict(x=x, conv=conv, bn=bn, stride=stride, pad=pad, relu=relu, y=y)


def res_block(P, x, st, bl, stride, first):
    """conv_block (resnet50.py:91-117) when `first`, else identity_block (41-63); the *_td
    variants (65-89,120-147) are the same maths with the RoI axis as batch."""
    base = "res%d%s_branch" % (st, bl)
    bnb = "bn%d%s_branch" % (st, bl)
    a, ca = _cbr(P, x, base + "2a", bnb + "2a", stride=stride if first else 1)
    b, cb = _cbr(P, a, base + "2b", bnb + "2b", pad=(1, 1, 1, 1))
    if first:
        sc, cs = _cbr(P, x, base + "1", bnb + "1", stride=stride, relu=False)
    else:
        sc, cs = x, None
    y, cc = _cbr(P, b, base + "2c", bnb + "2c", relu=True, add=sc)
    return y, dict(a=ca, b=cb, c=cc, s=cs, first=first)


def preprocess_caffe_bgr(img_bgr_u8):
    """RADNet.py:83-87 / utils.py:468-472 with keras 'caffe' preprocess_input: BGR->RGB, then
    RGB->BGR and subtract the ImageNet BGR means; net effect = BGR - [103.939,116.779,123.68]."""
    x = img_bgr_u8.astype(np.float32)
    x = x - np.array([103.939, 116.779, 123.68], dtype=np.float32)
    return x[None]


def base_forward(P, x, want_cache=False):
    """nn_base (resnet50.py:150-228): (1,H,W,3) -> (1,H/16,W/16,1024)."""
    caches = []
    y, c = _cbr(P, x, "conv1", "bn_conv1", stride=2, pad=(3, 3, 3, 3))
    caches.append(c)
    y = maxpool_3x3_s2(y)
    for st, blocks, _, stride in RES_STAGES:
        for bl in blocks:
            y, c = res_block(P, y, st, bl, stride, bl == "a")
            caches.append(c)
    return (y, caches) if want_cache else y


def base_backward(P, caches, dF, stages=(3, 4)):
    """Gradients of the trainable base stages (cont_train.py:120-131: `trainable` from stage 3 on; FixedBatchNormalization
    never trains) given dL/dF.  `caches` as returned by base_forward(want_cache=True): [conv1, then one entry per
    residual block in execution order].  Nothing flows below the first trainable stage."""
    blocks = []
    i = 1
    for st, bls, _, _ in RES_STAGES:
        for _ in bls:
            blocks.append((st, caches[i]))
            i += 1
    grads = {}
    dy = dF
    first_needed = min(stages)
    for st, c in reversed(blocks):
        if st not in stages:
            break
        lowest = st == first_needed and c["first"]              # the block whose input comes from the frozen part
        dy, gr = res_block_backward(P, c, dy, need_dx=not lowest)
        grads.update(gr)
    return grads


def s34_trainable():
    """Conv layers of ResNet50 stages 3 and 4 (the part of the base cont_train.py unfreezes)."""
    return [s[0] for s in resnet50_conv_specs() if s[0].startswith(("res3", "res4"))]


def rpn_forward(P, F):
    """rpn_layer (rpn.py:12-66): returns (cls sigmoid (1,H,W,A), regr (1,H,W,4A), cache)."""
    h = conv2d(F, P["rpn_conv1"]["kernel"], P["rpn_conv1"]["bias"], 1, (1, 1, 1, 1))
    h = np.maximum(h, 0)
    zc = conv2d(h, P["rpn_out_class"]["kernel"], P["rpn_out_class"]["bias"])
    zr = conv2d(h, P["rpn_out_regress"]["kernel"], P["rpn_out_regress"]["bias"])
    return sigmoid(zc), zr, dict(F=F, h=h, zc=zc)


def rpn_backward(P, cache, d_p, d_regr, need_dF=False):
    """Gradients of the three RPN convs given dL/d(sigmoid output) and dL/d(regr output)."""
    p = sigmoid(cache["zc"])
    dzc = d_p * p * (1 - p)
    h, F = cache["h"], cache["F"]
    dh1, dwc, dbc = conv2d_bwd(h, P["rpn_out_class"]["kernel"], dzc)
    dh2, dwr, dbr = conv2d_bwd(h, P["rpn_out_regress"]["kernel"], d_regr)
    dh = (dh1 + dh2) * (h > 0)
    dF, dw1, db1 = conv2d_bwd(F, P["rpn_conv1"]["kernel"], dh, 1, (1, 1, 1, 1), need_dx=need_dF)
    grads = {"rpn_conv1": {"kernel": dw1, "bias": db1}, "rpn_out_class": {"kernel": dwc, "bias": dbc},
             "rpn_out_regress": {"kernel": dwr, "bias": dbr}}
    return grads, dF


def head_forward(P, F, rois, n_classes=7, ps=14):
    """classifier_layer (resnet50.py:231-281): RoI crop-resize 14x14 -> stage 5 -> 7x7 avg pool ->
    dense softmax / dense linear.  rois (R,4) xywh fmap units.  Returns (P_cls (1,R,nc),
    P_regr (1,R,4(nc-1)), cache)."""
    pooled = roi_crop_resize(F, rois, ps)
    y = pooled
    caches = []
    st, blocks, _, stride = HEAD_STAGE
    for bl in blocks:
        y, c = res_block(P, y, st, bl, stride, bl == "a")
        caches.append(c)
    feat = y.mean(axis=(1, 2), dtype=y.dtype)                   # AveragePooling2D((7,7)) + Flatten
    dc, dr = P["dense_class_%d" % n_classes], P["dense_regress_%d" % n_classes]
    logits = feat @ dc["kernel"] + dc["bias"]
    pcls = softmax(logits)
    pregr = feat @ dr["kernel"] + dr["bias"]
    return pcls[None], pregr[None], dict(pooled=pooled, blocks=caches, y5=y, feat=feat, pcls=pcls, n_classes=n_classes)


def _cbr_backward(P, c, dy, need_dx=True):
    """Backward of _cbr: returns (dx, d_add, {conv: grads})."""
    g = dy * (c["y"] > 0) if c["relu"] else dy
    s, _ = bn_affine(P[c["bn"]])
    dz = g * s
    dx, dw, db = conv2d_bwd(c["x"], P[c["conv"]]["kernel"], dz, c["stride"], c["pad"], need_dx=need_dx)
    return dx, g, {c["conv"]: {"kernel": dw, "bias": db}}


def res_block_backward(P, c, dy, need_dx=True):
    grads = {}
    db_, g, gr = _cbr_backward(P, c["c"], dy)
    grads.update(gr)
    da, _, gr = _cbr_backward(P, c["b"], db_)
    grads.update(gr)
    dx, _, gr = _cbr_backward(P, c["a"], da, need_dx=need_dx)
    grads.update(gr)
    if c["first"]:
        dxs, _, gr = _cbr_backward(P, c["s"], g, need_dx=need_dx)
        grads.update(gr)
        dx = (dx + dxs) if need_dx else None
    else:
        dx = dx + g
    return dx, grads


def head_backward(P, cache, d_pcls, d_pregr, need_dpooled=False):
    """Gradients of every trainable head weight given dL/d(softmax output) and dL/d(regr output)."""
    nc = cache["n_classes"]
    q = cache["pcls"]
    dq = d_pcls[0]
    dlogits = q * (dq - (dq * q).sum(-1, keepdims=True))
    dregr = d_pregr[0]
    feat = cache["feat"]
    dc, dr = P["dense_class_%d" % nc], P["dense_regress_%d" % nc]
    grads = {"dense_class_%d" % nc: {"kernel": feat.T @ dlogits, "bias": dlogits.sum(0)},
             "dense_regress_%d" % nc: {"kernel": feat.T @ dregr, "bias": dregr.sum(0)}}
    dfeat = dlogits @ dc["kernel"].T + dregr @ dr["kernel"].T
    y5 = cache["y5"]
    dy = np.broadcast_to(dfeat[:, None, None, :] / (y5.shape[1] * y5.shape[2]), y5.shape).astype(y5.dtype)
    for i, c in enumerate(reversed(cache["blocks"])):
        last = i == len(cache["blocks"]) - 1
        dy, gr = res_block_backward(P, c, dy, need_dx=(not last) or need_dpooled)
        grads.update(gr)
    return grads, dy


# ----------------------------------------------------------------------------------------
# composed steps (train.py:288-402)
# ----------------------------------------------------------------------------------------
def rpn_losses_and_grads(P, F, y_cls, y_regr, A, keras2_arg_order=True):
    p, r, cache = rpn_forward(P, F)
    l_cls, dp = rpn_loss_cls(y_cls, p, A, keras2_arg_order)
    l_regr, dr = smooth_l1_masked(y_regr, r, 4 * A)
    grads, _ = rpn_backward(P, cache, dp, dr)
    return [l_cls + l_regr, l_cls, l_regr], grads


def head_losses_and_grads(P, F, rois, Y1, Y2, n_classes=7):
    pc, pr, cache = head_forward(P, F, rois, n_classes)
    l_cls, dpc = class_loss_cls(Y1, pc)
    l_regr, dpr = smooth_l1_masked(Y2, pr, 4 * (n_classes - 1))
    grads, _ = head_backward(P, cache, dpc, dpr)
    return [l_cls + l_regr, l_cls, l_regr, categorical_accuracy(Y1, pc)], grads


class AdamState:
    """One Keras Adam instance over a named parameter subset (train.py:236-252: the RPN model and
    the classifier model each own one, even for weights they share)."""

    def __init__(self, P, names, lr):
        self.names = list(names)
        self.lr = lr
        self.t = 0
        self.m = {n: {k: np.zeros_like(v) for k, v in P[n].items()} for n in self.names}
        self.v = {n: {k: np.zeros_like(v) for k, v in P[n].items()} for n in self.names}

    def apply(self, P, grads):
        self.t += 1
        for n in self.names:
            for k in ("kernel", "bias"):
                adam_step(P[n][k], grads[n][k].astype(P[n][k].dtype), self.m[n][k], self.v[n][k], self.t, self.lr)


RPN_TRAINABLE = ("rpn_conv1", "rpn_out_class", "rpn_out_regress")


def head_trainable(n_classes=7):
    names = [s[0] for s in resnet50_conv_specs() if s[0].startswith("res5")]
    return names + ["dense_class_%d" % n_classes, "dense_regress_%d" % n_classes]
