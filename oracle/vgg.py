"""ORACLE (test infrastructure, not product): NumPy restatement of the reference's VGG16 variant
(faster_rcnn/base_models/vgg16.py) -- BASELINE config 5.  PARITY UNPINNED (Keras/TF absent; keras.applications
VGG16 is an un-vendored dependency): follows vgg16.py:29-124 and the published VGG16 topology
(13 3x3 'same' convs with ReLU, 2x2/2 max-pools after blocks 1-4, cut at block5_conv3), torch-cross-checked.

Dropout(0.5) (vgg16.py:99,101) is active only in train_on_batch and uses TF's RNG, which cannot be matched; the
functions here take the two keep-masks explicitly (already scaled by 1/(1-rate) = 2) so a test can feed the same
masks to the device path.
"""
import numpy as np

from . import dense

VGG_BLOCKS = ((1, 2, 64), (2, 2, 128), (3, 3, 256), (4, 3, 512), (5, 3, 512))   # block, convs, channels
POOL = 7


def conv_specs():
    specs, cin = [], 3
    for b, n, ch in VGG_BLOCKS:
        for i in range(1, n + 1):
            specs.append(("block%d_conv%d" % (b, i), cin, ch))
            cin = ch
    return specs


def feat_len(n):
    """vgg16.get_img_output_length (vgg16.py:18-23)."""
    return n // 16


def init_params(seed=5, n_anchors=9, n_classes=7, dtype=np.float32):
    rs = np.random.RandomState(seed)
    P = {}
    for name, cin, cout in conv_specs():
        std = np.sqrt(2.0 / (9 * cin)) * (1.0 / 70.0 if name == "block1_conv1" else 1.0)
        P[name] = {"kernel": (rs.standard_normal((3, 3, cin, cout)) * std).astype(dtype), "bias": (rs.standard_normal(cout) * 0.05).astype(dtype)}
    P["rpn_conv1"] = {"kernel": (rs.standard_normal((3, 3, 512, 512)) * np.sqrt(2.0 / (9 * 512)) * 0.3).astype(dtype),
                      "bias": (rs.standard_normal(512) * 0.05).astype(dtype)}
    P["rpn_out_class"] = {"kernel": (rs.standard_normal((1, 1, 512, n_anchors)) * 0.03).astype(dtype), "bias": np.zeros(n_anchors, dtype)}
    P["rpn_out_regress"] = {"kernel": (rs.standard_normal((1, 1, 512, 4 * n_anchors)) * 0.03).astype(dtype), "bias": np.zeros(4 * n_anchors, dtype)}
    P["fc1"] = {"kernel": (rs.standard_normal((POOL * POOL * 512, 4096)) * np.sqrt(2.0 / (POOL * POOL * 512))).astype(dtype),
                "bias": (rs.standard_normal(4096) * 0.05).astype(dtype)}
    P["fc2"] = {"kernel": (rs.standard_normal((4096, 4096)) * np.sqrt(2.0 / 4096)).astype(dtype), "bias": (rs.standard_normal(4096) * 0.05).astype(dtype)}
    nreg = 4 * (n_classes - 1)
    P["dense_class_%d" % n_classes] = {"kernel": (rs.standard_normal((4096, n_classes)) * 0.01).astype(dtype), "bias": (rs.standard_normal(n_classes) * 0.01).astype(dtype)}
    P["dense_regress_%d" % n_classes] = {"kernel": (rs.standard_normal((4096, nreg)) * 0.01).astype(dtype), "bias": (rs.standard_normal(nreg) * 0.01).astype(dtype)}
    return P


def base_forward(P, x):
    """vgg16.nn_base (vgg16.py:29-65): VGG16 up to block5_conv3.  x (1,H,W,3) -> (1,H//16,W//16,512)."""
    y = x
    for b, n, ch in VGG_BLOCKS:
        for i in range(1, n + 1):
            name = "block%d_conv%d" % (b, i)
            y = np.maximum(dense.conv2d(y, P[name]["kernel"], P[name]["bias"], 1, (1, 1, 1, 1)), 0)
        if b < 5:
            y = dense.maxpool_2x2_s2(y)
    return y


def head_forward(P, F, rois, n_classes=7, masks=None):
    """vgg16.classifier_layer (vgg16.py:67-124): RoI crop-resize 7x7 -> Flatten -> fc1 ReLU [Dropout] -> fc2 ReLU
    [Dropout] -> dense softmax / dense linear.  masks = (m1, m2) keep-masks already scaled by 2, or None (inference)."""
    pooled = dense.roi_crop_resize(F, rois, POOL)
    R = pooled.shape[0]
    flat = pooled.reshape(R, -1)
    z1 = flat @ P["fc1"]["kernel"] + P["fc1"]["bias"]
    h1 = np.maximum(z1, 0)
    d1 = h1 * masks[0] if masks is not None else h1
    z2 = d1 @ P["fc2"]["kernel"] + P["fc2"]["bias"]
    h2 = np.maximum(z2, 0)
    d2 = h2 * masks[1] if masks is not None else h2
    dc, dr = P["dense_class_%d" % n_classes], P["dense_regress_%d" % n_classes]
    pcls = dense.softmax(d2 @ dc["kernel"] + dc["bias"])
    pregr = d2 @ dr["kernel"] + dr["bias"]
    return pcls[None], pregr[None], dict(flat=flat, h1=h1, d1=d1, h2=h2, d2=d2, pcls=pcls, masks=masks, n_classes=n_classes)


def head_backward(P, cache, d_pcls, d_pregr):
    nc = cache["n_classes"]
    q = cache["pcls"]
    dq = d_pcls[0]
    dlogits = q * (dq - (dq * q).sum(-1, keepdims=True))
    dregr = d_pregr[0]
    d2, d1, flat, masks = cache["d2"], cache["d1"], cache["flat"], cache["masks"]
    dc, dr = P["dense_class_%d" % nc], P["dense_regress_%d" % nc]
    grads = {"dense_class_%d" % nc: {"kernel": d2.T @ dlogits, "bias": dlogits.sum(0)},
             "dense_regress_%d" % nc: {"kernel": d2.T @ dregr, "bias": dregr.sum(0)}}
    g = dlogits @ dc["kernel"].T + dregr @ dr["kernel"].T
    if masks is not None:
        g = g * masks[1]
    g = g * (cache["h2"] > 0)
    grads["fc2"] = {"kernel": d1.T @ g, "bias": g.sum(0)}
    g = g @ P["fc2"]["kernel"].T
    if masks is not None:
        g = g * masks[0]
    g = g * (cache["h1"] > 0)
    grads["fc1"] = {"kernel": flat.T @ g, "bias": g.sum(0)}
    return grads


def head_losses_and_grads(P, F, rois, Y1, Y2, n_classes=7, masks=None):
    pc, pr, cache = head_forward(P, F, rois, n_classes, masks)
    l_cls, dpc = dense.class_loss_cls(Y1, pc)
    l_regr, dpr = dense.smooth_l1_masked(Y2, pr, 4 * (n_classes - 1))
    return [l_cls + l_regr, l_cls, l_regr, dense.categorical_accuracy(Y1, pc)], head_backward(P, cache, dpc, dpr)


HEAD_TRAINABLE = ("fc1", "fc2")
