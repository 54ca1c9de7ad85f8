"""VGG16 variant of the engine (BASELINE config 5): the reference's base_models/vgg16.py graph on the same kernels.

  base   vgg16.py:29-65    keras.applications VGG16 cut at block5_conv3: 13 3x3 'same' convs + ReLU (bias, no BN),
                           2x2/2 max-pools after blocks 1-4; stride 16, 512 channels
  head   vgg16.py:67-124   RoI crop-resize 7x7 -> Flatten (25088) -> fc1 4096 ReLU -> Dropout .5 -> fc2 4096 ReLU ->
                           Dropout .5 -> dense softmax / dense linear
The fully-connected layers run on the conv GEMM kernels as 1x1 convs over 1x1 "images" (M = #RoIs, K = 25088 / 4096):
weight-bandwidth-bound (fc1 is 411 MB fp32), split-K autotuned.  Dropout masks come from a host RandomState (TF's
RNG cannot be matched: parity unpinned) or are injected by tests.
"""
import ctypes as C

import numpy as np
import torch

from . import lib as L
from .engine import Arena, ConvLayer, FasterRCNNEngine, RPN_LD

VGG_BLOCKS = ((1, 2, 64), (2, 2, 128), (3, 3, 256), (4, 3, 512), (5, 3, 512))
POOL = 7


def vgg_feat_len(n):
    """vgg16.get_img_output_length (vgg16.py:18-23)."""
    return n // 16


class VGG16Engine(FasterRCNNEngine):
    NETWORK = "vgg16"
    N_FEATURES = 512
    HEAD_TRAIN_WINOGRAD = False  # the classifier head is two dense layers here
    CROP_AHEAD = False           # (its head_forward crops 7x7 itself)
    supports_batched = False     # the mini-batch runs image by image (fc head plan is per feature map)
    feat_len = staticmethod(vgg_feat_len)

    # ------------------------------------------------------------------------------------------ layers
    def _build_layers(self):
        dev = self.dev
        self.convs = {}
        cin = 4                                               # image padded to 4 channels
        self.base_names = []
        for b, n, ch in VGG_BLOCKS:
            for i in range(1, n + 1):
                name = "block%d_conv%d" % (b, i)
                c = ConvLayer(name, 3, cin, ch, 1, 1)
                c.weight = torch.zeros(9 * cin, ch, dtype=torch.float32, device=dev)
                c.shift = torch.zeros(ch, dtype=torch.float32, device=dev)       # bias only (no BN in VGG16)
                self.convs[name] = c
                self.base_names.append(name)
                cin = ch
        self.convs["rpn_conv1"] = ConvLayer("rpn_conv1", 3, 512, 512, 1, 1)
        self.convs["rpn_heads"] = ConvLayer("rpn_heads", 1, 512, 5 * self.A, 1, 0, ldw=RPN_LD)
        self.rpn_arena = Arena(dev)
        for name in ("rpn_conv1", "rpn_heads"):
            c = self.convs[name]
            self.rpn_arena.add(name + "/kernel", (c.kh * c.kh * c.cin, c.ldw))
            self.rpn_arena.add(name + "/bias", (c.ldw,))
        self.rpn_arena.finalize()
        for name in ("rpn_conv1", "rpn_heads"):
            c = self.convs[name]
            c.weight, c.bias = self.rpn_arena.param(name + "/kernel"), self.rpn_arena.param(name + "/bias")
            c.dweight, c.dbias = self.rpn_arena.grad(name + "/kernel"), self.rpn_arena.grad(name + "/bias")
            c.shift = c.bias
        # head arena: fc1, fc2 (as 1x1 convs) + fused dense heads
        self.convs["fc1"] = ConvLayer("fc1", 1, POOL * POOL * 512, 4096)
        self.convs["fc2"] = ConvLayer("fc2", 1, 4096, 4096)
        self.head_conv_names = ["fc1", "fc2"]
        self.head_arena = Arena(dev)
        for name in self.head_conv_names:
            c = self.convs[name]
            self.head_arena.add(name + "/kernel", (c.cin, c.cout))
            self.head_arena.add(name + "/bias", (c.cout,))
        self.head_arena.add("dense/kernel", (4096, self.dense_ld))
        self.head_arena.add("dense/bias", (self.dense_ld,))
        self.head_arena.finalize()
        for name in self.head_conv_names:
            c = self.convs[name]
            c.weight, c.bias = self.head_arena.param(name + "/kernel"), self.head_arena.param(name + "/bias")
            c.dweight, c.dbias = self.head_arena.grad(name + "/kernel"), self.head_arena.grad(name + "/bias")
            c.shift = c.bias
        self.dense_w, self.dense_b = self.head_arena.param("dense/kernel"), self.head_arena.param("dense/bias")
        self.dense_dw, self.dense_db = self.head_arena.grad("dense/kernel"), self.head_arena.grad("dense/bias")
        self.dropout_rng = np.random.RandomState(1234)
        self.training = True
        self.forced_masks = None

    def refresh_head_shift(self):
        pass                                                  # no BN in the VGG head: shift IS the bias

    # ------------------------------------------------------------------------------------------ weights
    def set_weights(self, W):
        dev = self.dev

        def t(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)

        for name in self.base_names:
            c = self.convs[name]
            kern = np.asarray(W[name]["kernel"], dtype=np.float32)
            if name == "block1_conv1":
                kern = np.concatenate([kern, np.zeros((3, 3, 1, 64), np.float32)], axis=2)
            c.weight.copy_(t(kern.reshape(-1, c.cout)))
            c.shift.copy_(t(W[name]["bias"]))
        c = self.convs["rpn_conv1"]
        c.weight.copy_(t(np.asarray(W["rpn_conv1"]["kernel"]).reshape(-1, 512))); c.bias.copy_(t(W["rpn_conv1"]["bias"]))
        k = np.zeros((512, RPN_LD), np.float32); b = np.zeros(RPN_LD, np.float32)
        k[:, :self.A] = np.asarray(W["rpn_out_class"]["kernel"]).reshape(512, self.A)
        k[:, self.A:5 * self.A] = np.asarray(W["rpn_out_regress"]["kernel"]).reshape(512, 4 * self.A)
        b[:self.A] = W["rpn_out_class"]["bias"]; b[self.A:5 * self.A] = W["rpn_out_regress"]["bias"]
        self.convs["rpn_heads"].weight.copy_(t(k)); self.convs["rpn_heads"].bias.copy_(t(b))
        for name in self.head_conv_names:
            c = self.convs[name]
            c.weight.copy_(t(W[name]["kernel"])); c.bias.copy_(t(W[name]["bias"]))
        dc, dr = W["dense_class_%d" % self.nc], W["dense_regress_%d" % self.nc]
        k = np.zeros((4096, self.dense_ld), np.float32); b = np.zeros(self.dense_ld, np.float32)
        k[:, :self.nc] = dc["kernel"]; k[:, self.nc:self.nc + self.nreg] = dr["kernel"]
        b[:self.nc] = dc["bias"]; b[self.nc:self.nc + self.nreg] = dr["bias"]
        self.dense_w.copy_(t(k)); self.dense_b.copy_(t(b))
        torch.cuda.synchronize(self.dev)

    def get_weights(self, names=None):
        out = {}
        for name in ["rpn_conv1"] + self.head_conv_names:
            c = self.convs[name]
            kern = c.weight.detach().cpu().numpy()
            out[name] = {"kernel": (kern.reshape(3, 3, 512, 512) if name == "rpn_conv1" else kern).copy(), "bias": c.bias.detach().cpu().numpy().copy()}
        k = self.convs["rpn_heads"].weight.detach().cpu().numpy(); b = self.convs["rpn_heads"].bias.detach().cpu().numpy()
        out["rpn_out_class"] = {"kernel": k[:, :self.A].reshape(1, 1, 512, self.A).copy(), "bias": b[:self.A].copy()}
        out["rpn_out_regress"] = {"kernel": k[:, self.A:5 * self.A].reshape(1, 1, 512, 4 * self.A).copy(), "bias": b[self.A:5 * self.A].copy()}
        k = self.dense_w.detach().cpu().numpy(); b = self.dense_b.detach().cpu().numpy()
        out["dense_class_%d" % self.nc] = {"kernel": k[:, :self.nc].copy(), "bias": b[:self.nc].copy()}
        out["dense_regress_%d" % self.nc] = {"kernel": k[:, self.nc:self.nc + self.nreg].copy(), "bias": b[self.nc:self.nc + self.nreg].copy()}
        return out

    # ------------------------------------------------------------------------------------------ plans
    def _plan_base(self, nb, H, W, slot=0):
        key = ("base", nb, H, W, slot)
        if key in self._plans:
            return self._plans[key]
        dev = self.dev
        ops, keep = [], []

        def buf(*shape):
            b = torch.empty(shape, dtype=torch.float32, device=dev)
            keep.append(b)
            return b

        x = buf(nb, H, W, 4)
        cur, h, w = x, H, W
        for b, n, ch in VGG_BLOCKS:
            for i in range(1, n + 1):
                c = self.convs["block%d_conv%d" % (b, i)]
                y = buf(nb, h, w, ch)
                d, _, _ = self._desc(c, cur, nb, h, w, y, relu=True)
                ops.append(("conv", d))
                cur = y
            if b < 5:
                ph, pw = h // 2, w // 2
                p = buf(nb, ph, pw, ch)
                ops.append(("maxpool", (cur, p, nb, h, w, ch, 2, 2)))
                cur, h, w = p, ph, pw
        plan = dict(ops=ops, x=x, F=cur, fh=h, fw=w, keep=keep)
        self._plans[key] = plan
        return plan

    def _fc_desc(self, c, x, R, y, relu):
        d = L.ConvDesc()
        d.x, d.w, d.y = x.data_ptr(), c.weight.data_ptr(), y.data_ptr()
        d.scale, d.shift, d.addend = None, c.shift.data_ptr(), None
        d.nb, d.h, d.w_, d.c, d.oh, d.ow = R, 1, 1, c.cin, 1, 1
        d.kh = d.kw = 1
        d.stride, d.pad_t, d.pad_l, d.n = 1, 0, 0, c.cout
        d.ldw, d.ldy, d.ld_add, d.act, d.act_cols = c.cout, c.cout, c.cout, 1 if relu else 0, 0
        return d

    def _plan_head(self, R, fh, fw, F, training=True):       # (one plan form: the fc head's buffers are small)
        key = ("head", R, fh, fw, F.data_ptr())
        if key in self._plans:
            return self._plans[key]
        dev = self.dev
        keep = []

        def buf(*shape):
            b = torch.empty(shape, dtype=torch.float32, device=dev)
            keep.append(b)
            return b

        rois = buf(R, 4)
        pooled = buf(R, POOL * POOL * 512)
        h1, d1, h2, d2 = buf(R, 4096), buf(R, 4096), buf(R, 4096), buf(R, 4096)
        m1, m2 = buf(R, 4096), buf(R, 4096)
        zero = torch.zeros(R, 4096, dtype=torch.float32, device=dev)
        keep.append(zero)
        f1, f2 = self.convs["fc1"], self.convs["fc2"]
        fd1 = self._fc_desc(f1, pooled, R, h1, True)
        fd2 = self._fc_desc(f2, d1, R, h2, True)
        pcls, pregr = buf(R, self.nc), buf(R, self.nreg)
        y1, y2 = buf(R, self.nc), buf(R, 2 * self.nreg)
        dz = buf(R, self.nc + self.nreg)
        g2, g1 = buf(R, 4096), buf(R, 4096)
        # backward descriptors: fc2 wgrad + dgrad (masked by d1 > 0 = kept & active), fc1 wgrad
        b2 = L.ConvDesc.from_buffer_copy(fd2)
        b2.dy, b2.ld_dy, b2.gscale = g2.data_ptr(), 4096, None
        b2.dw, b2.dw_accumulate = f2.dweight.data_ptr(), 1
        b2.dx, b2.ld_dx, b2.dx_add, b2.dx_mask, b2.ld_dx_mask = g1.data_ptr(), 4096, None, d1.data_ptr(), 4096
        b1 = L.ConvDesc.from_buffer_copy(fd1)
        b1.dy, b1.ld_dy, b1.gscale = g1.data_ptr(), 4096, None
        b1.dw, b1.dw_accumulate = f1.dweight.data_ptr(), 1
        bwd_a = [("wgrad", b2), ("colsum", [g2.data_ptr(), R, 4096, 4096, None, f2.dbias.data_ptr(), 1]), ("dgrad", b2)]
        bwd_b = [("wgrad", b1), ("colsum", [g1.data_ptr(), R, 4096, 4096, None, f1.dbias.data_ptr(), 1])]
        bwd_a, bwd_b = self._fuse_bias_grads(bwd_a), self._fuse_bias_grads(bwd_b)
        plan = dict(R=R, rois=rois, pooled=pooled, fwd1=[("conv", fd1)], fwd2=[("conv", fd2)], bwd=bwd_a + bwd_b, bwd_a=bwd_a, bwd_b=bwd_b,
                    h1=h1, d1=d1, h2=h2, d2=d2, m1=m1, m2=m2, zero=zero, feat=d2, pcls=pcls, pregr=pregr, y1=y1, y2=y2, dz=dz, g2=g2, g1=g1,
                    F=F, fh=fh, fw=fw, keep=keep)
        self._plans[key] = plan
        return plan

    def _masks(self, hp):
        R = hp["R"]
        if self.forced_masks is not None:
            m1, m2 = self.forced_masks
        else:
            m1 = (self.dropout_rng.uniform(size=(R, 4096)) >= 0.5).astype(np.float32) * 2.0
            m2 = (self.dropout_rng.uniform(size=(R, 4096)) >= 0.5).astype(np.float32) * 2.0
        hp["m1"].copy_(torch.from_numpy(np.ascontiguousarray(m1, dtype=np.float32)))
        hp["m2"].copy_(torch.from_numpy(np.ascontiguousarray(m2, dtype=np.float32)))

    def head_forward(self, hp, training=False, loss_out=None, group_live=None):     # (loss_out: the fc head keeps its separate loss pass)
        """vgg16.classifier_layer forward; Dropout only when `training` (keras learning phase)."""
        n = hp["R"] * 4096
        self.ctx.call("radnet_roi_resize_fwd", hp["F"], hp["fh"], hp["fw"], 512, hp["rois"], hp["R"], POOL, hp["pooled"])
        self._run(hp["fwd1"])
        hp["training"] = training
        if training:
            self._masks(hp)
            self.ctx.call("radnet_affine_vec", hp["d1"], hp["h1"], hp["m1"], hp["zero"], C.c_int64(n))
        else:
            hp["d1"].copy_(hp["h1"])
        self._run(hp["fwd2"])
        if training:
            self.ctx.call("radnet_affine_vec", hp["d2"], hp["h2"], hp["m2"], hp["zero"], C.c_int64(n))
        else:
            hp["d2"].copy_(hp["h2"])
        self.ctx.call("radnet_dense_heads_fwd", hp["d2"], hp["R"], 4096, self.dense_w, self.dense_ld, self.dense_b, self.nc, self.nreg,
                      hp["pcls"], hp["pregr"])

    def head_losses_only(self, hp, loss_out=None):
        """model_classifier.test_on_batch (train.py:513): Dropout off (Keras test phase), losses + accuracy, no gradients."""
        self.head_forward(hp, training=False)
        self.ctx.call("radnet_det_loss", hp["pcls"], hp["pregr"], hp["y1"], hp["y2"], hp["R"], self.nc, self.nreg, hp["dz"],
                      self.det_losses if loss_out is None else loss_out)

    def head_backward(self, hp, accumulate=False, loss_out=None):
        n = hp["R"] * 4096
        self.ctx.call("radnet_det_loss", hp["pcls"], hp["pregr"], hp["y1"], hp["y2"], hp["R"], self.nc, self.nreg, hp["dz"],
                      self.det_losses if loss_out is None else loss_out)
        # dense heads: dW, db, and g2 = gradient w.r.t. d2
        self.ctx.call("radnet_dense_heads_bwd", hp["d2"], hp["dz"], hp["R"], 4096, self.dense_w, self.dense_ld, self.nc + self.nreg,
                      self.dense_dw, self.dense_db, hp["g2"], 1 if accumulate else 0)
        # through Dropout (x mask) and ReLU: g2 *= m2 ; zero where h2 <= 0 (d2 > 0 <=> kept and active)
        if hp.get("training"):
            self.ctx.call("radnet_affine_vec", hp["g2"], hp["g2"], hp["m2"], hp["zero"], C.c_int64(n))
        self.ctx.call("radnet_relu_mask", hp["g2"], hp["h2"], C.c_int64(n))
        self._run(hp["bwd_a"])                       # fc2: wgrad, bias grad, dgrad -> g1 masked by d1 > 0
        if hp.get("training"):
            self.ctx.call("radnet_affine_vec", hp["g1"], hp["g1"], hp["m1"], hp["zero"], C.c_int64(n))
        self._run(hp["bwd_b"])                       # fc1: wgrad, bias grad


def make_engine(C_cfg, **kw):
    """Engine for Config.network (train.py:145-151 / RADNet.py:727-733 select the backbone module the same way)."""
    if C_cfg.network == "resnet50":
        return FasterRCNNEngine(C_cfg, **kw)
    if C_cfg.network == "vgg16":
        return VGG16Engine(C_cfg, **kw)
    raise L.RadnetError("Not a valid base model! (%r)" % (C_cfg.network,))
