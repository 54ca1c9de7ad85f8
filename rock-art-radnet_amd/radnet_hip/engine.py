"""Device-side engine of the MI355X-native Faster R-CNN path.

Python here is a *scheduler*: it owns the weight / activation buffers (torch-ROCm tensors used purely
as containers), lays the reference's Keras graph out as a static list of kernel launches per image
size, and calls libradnet_hip.so through the C ABI (radnet_hip.lib).  No torch arithmetic runs on the
hot path.

Graph restated from the reference (never imported):
  base   faster_rcnn/base_models/resnet50.py:150-228   conv1 + stages 2-4, frozen BN folded into epilogues
  RPN    faster_rcnn/rpn.py:12-66                      3x3 conv + ONE fused 1x1 GEMM for both heads
  head   faster_rcnn/base_models/resnet50.py:231-281   RoI crop-resize 14x14 + stage 5 + avgpool + dense
  step   train.py:288-402                              RPN train -> re-predict -> propose/label/sample -> head train
"""
import collections
import ctypes as C
import contextlib
import os
import math

import numpy as np
import torch

from . import lib as L

BN_EPS = 1e-3        # FixedBatchNormalization.py:8
RES_STAGES = ((2, "abc", (64, 64, 256), 1), (3, "abcd", (128, 128, 512), 2), (4, "abcdef", (256, 256, 1024), 2))
HEAD_STAGE = (5, "abc", (512, 512, 2048), 2)
RPN_LD = 64          # fused RPN head GEMM width (A + 4A = 60 for 12 anchors, padded)


def feat_len(n):
    """resnet50.get_img_output_length (resnet50.py:19-35)."""
    n += 6
    for k in (7, 3, 1, 1):
        n = (n - k + 2) // 2
    return n


def _pad4(n):
    return (n + 3) // 4 * 4


class Arena:
    """One flat fp32 parameter arena (+ grads, Adam moments) with named views: a single Adam launch and a
    single all-reduce cover every tensor of an optimizer (train.py:236-252 has one Adam per model)."""

    def __init__(self, device):
        self.device = device
        self.spec = []       # (name, shape)
        self.views = {}
        self.offsets = {}
        self.n = 0

    def add(self, name, shape):
        size = int(np.prod(shape))
        self.spec.append((name, tuple(shape)))
        self.offsets[name] = (self.n, size)
        self.n += _pad4(size)

    def finalize(self):
        self.n = (self.n + 255) // 256 * 256
        self.p = torch.zeros(self.n, dtype=torch.float32, device=self.device)
        self.g = torch.zeros_like(self.p)
        self.m = torch.zeros_like(self.p)
        self.v = torch.zeros_like(self.p)
        self.t = 0
        for name, shape in self.spec:
            o, s = self.offsets[name]
            self.views[name] = self.p[o:o + s].view(shape)
            self.views["d:" + name] = self.g[o:o + s].view(shape)

    def param(self, name):
        return self.views[name]

    def grad(self, name):
        return self.views["d:" + name]


class PlanCache(collections.OrderedDict):
    """Layer plans keyed by (kind, shape..., buffer set / feature-map pointer), least-recently-used first.  Every distinct
    image / tile shape and RoI count allocates a plan (activation buffers, descriptors, recorded hipGraphs): a long
    predict run over variable-size inputs would otherwise grow device memory without bound.  At most `limit` plans per
    kind stay; the oldest is dropped through `on_evict` (the engine waits for the device, then forgets its graphs)."""

    def __init__(self, limit, on_evict):
        super().__init__()
        self.limit, self.on_evict = limit, on_evict

    def __getitem__(self, key):
        v = super().__getitem__(key)
        self.move_to_end(key)
        return v

    def __setitem__(self, key, value):
        super().__setitem__(key, value)
        self.move_to_end(key)
        same = [k for k in self if k[0] == key[0]]
        while len(same) > self.limit:
            old = same.pop(0)
            value = collections.OrderedDict.__getitem__(self, old)
            collections.OrderedDict.__delitem__(self, old)
            self.on_evict(old, value)


class ConvLayer:
    """One convolution of the static layer program (descriptor prebuilt; pointers are stable)."""

    def __init__(self, name, kh, cin, cout, stride=1, pad=0, ldw=None):
        self.name, self.kh, self.cin, self.cout, self.stride, self.pad = name, kh, cin, cout, stride, pad
        self.ldw = ldw or cout
        self.weight = None      # [kh*kh*cin][ldw]
        self.bias = None        # [cout] raw conv bias
        self.scale = None       # folded BN scale or None
        self.shift = None       # folded epilogue shift
        self.t0 = None          # BN shift without the conv bias (trainable convs refresh shift = scale*bias + t0)
        self.dweight = None
        self.dbias = None
        self.wino_u = None      # Winograd-transformed filter [(m+2)^2][cin][cout] when the layer runs as F(m x m, 3x3)
        self.wino_m = 2         # output tile of the Winograd form: 2 = F(2x2,3x3), 4 = F(4x4,3x3)


class FasterRCNNEngine:
    """ResNet50 Faster R-CNN on one MI355X.  `mode`: 'train' = train.py trainability (whole base frozen);
    inference uses the same object."""

    NETWORK = "resnet50"
    N_FEATURES = 1024
    # Shipped launch-shape tables (radnet_hip/tuned/<workload>_<network>_*.txt) belong to the workload AND network they were tuned in:
    # the pipelined train.py step of this network by default; an engine built for another workload names it (`workload="predict"`:
    # RADNet.predict's engine loads no train-step table -- its shapes are measured alone on first use, as they run)
    WORKLOAD = "train"
    supports_batched = True      # per-GPU mini-batch as one layer program (upload_images / _plan_rpn(nb) / _plan_head(groups))
    feat_len = staticmethod(feat_len)

    def __init__(self, C_cfg, device_index=0, n_classes=None, bce_mode=0, lr=5e-5, autotune=True, workload=None):
        self.workload = workload or self.WORKLOAD
        self.TUNED_PREFIX = "%s_%s_" % (self.workload, self.NETWORK)
        if C_cfg.network != self.NETWORK:
            raise L.RadnetError("engine: %s asked to run network %r (use radnet_hip.make_engine)" % (type(self).__name__, C_cfg.network))
        self.C = C_cfg
        self.dev = torch.device("cuda", device_index)
        torch.cuda.set_device(self.dev)
        self.ctx = L.Context(device_index)
        self.lib = self.ctx.lib
        self.A = len(C_cfg.anchor_box_scales) * len(C_cfg.anchor_box_ratios)
        self.nc = n_classes or len(C_cfg.class_mapping)
        self.nreg = 4 * (self.nc - 1)
        self.bg = C_cfg.class_mapping.get("bg", self.nc - 1)
        self.bce_mode = bce_mode
        self.lr = lr
        if 5 * self.A > RPN_LD:
            raise L.RadnetError("engine: %d anchors exceed the fused RPN head width" % self.A)
        if self.nc + self.nreg > 64:
            raise L.RadnetError("engine: %d classes need %d dense-head columns, the fused dense-head kernel holds 64 (at most 13 classes)"
                                % (self.nc, self.nc + self.nreg))
        self.dense_ld = 32 if self.nc + self.nreg <= 32 else 64
        self._build_layers()
        # 32 per kind: the training step keeps 6 buffer sets x images per GPU alive at once; beyond that, least recently used
        self._plans = PlanCache(int(os.environ.get("RADNET_PLAN_CACHE", "32")), self._evict_plan)
        self._graphs = {}
        self._compiled = {}
        self.use_graphs = os.environ.get("RADNET_NO_GRAPHS", "0") != "1"
        # classifier tail (avg-pool, dense heads, detector losses) as one launch instead of three
        self.fuse_tail = os.environ.get("RADNET_NO_TAIL_FUSION", "0") != "1"
        self.use_winograd = os.environ.get("RADNET_NO_WINOGRAD", "0") != "1"
        # frozen base forward (stages 2-4) as one persistent launch of `chain_wgs` workgroups (0: two per CU); DESIGN.md 4
        self.use_chain = os.environ.get("RADNET_CHAIN", "0") == "1"
        # branch2a + shortcut conv of a conv_block as one call (radnet_conv_fwd_pair decides per shape pair whether one launch is faster)
        self.fwd_pair = os.environ.get("RADNET_NO_FWD_PAIR", "0") != "1"      # (a base plan built for the chain kernel keeps the single convs)
        # frozen stage-2 blocks: 3x3 + 1x1 expand (+ the next block's 1x1 reduce) as one call (radnet_conv_bottleneck decides per shape)
        self.bneck_fuse = os.environ.get("RADNET_NO_BNECK_FUSE", "0") != "1"
        # the classifier's three 3x3 convs run their TRAINING forward on Winograd F(4x4,3x3) filters too; Adam #2 rewrites the transformed
        # filters in its own pass (radnet_adam_step_fused).  The backward stays the direct form.
        self.head_train_wino = (self.HEAD_TRAIN_WINOGRAD and self.use_winograd and os.environ.get("RADNET_NO_HEAD_TRAIN_WINOGRAD", "0") != "1"
                                and os.environ.get("RADNET_NO_INFERENCE_WINOGRAD", "0") != "1")
        self._adam_wino = None
        # 256 by default: the chain's static deal needs every workgroup of every concurrently running chain resident, and the
        # pipelined step runs two of them (prefetch lanes) beside the RPN and classifier lanes' launches (1 024 slots on the chip)
        self.chain_wgs = int(os.environ.get("RADNET_CHAIN_WGS", "256"))
        self._chain_plans = []     # plans whose base forward is a chain launch: check_chains() reads their sticky error words
        self.wino_wgrad = os.environ.get("RADNET_NO_WINOGRAD_WGRAD", "0") != "1"
        # ... and their weight gradients in the Winograd domain, on the transformed input the forward pass left (the rpn_conv1 path)
        self.head_wino_wgrad = self.head_train_wino and self.wino_wgrad and os.environ.get("RADNET_NO_HEAD_WINO_WGRAD", "0") != "1"
        self.ws = torch.empty(256 << 20, dtype=torch.uint8, device=self.dev)         # split-K partials
        self.ctx.check(self.lib.radnet_set_workspace(self.ctx.h, self.ws.data_ptr(), self.ws.numel()), "set_workspace")
        # 0 off / 1 measure every new GEMM shape / 2 adopt the nearest measured M first (variable tile sizes, lib header)
        self.autotune_mode = int(os.environ.get("RADNET_AUTOTUNE", int(autotune))) if autotune else 0
        self.ctx.check(self.lib.radnet_set_autotune(self.ctx.h, self.autotune_mode), "set_autotune")
        # Lanes: further contexts on their own streams (TrainStep's pipelined step, see lane()).  (Forking only the wgrad
        # GEMMs of a backward program to a second stream was measured too: 3.48 ms/step against 3.41 on one stream.)
        # A separate upload stream is OFF by default: with the lanes in place the upload already runs on a prefetch lane,
        # off every critical chain, and one more busy HIP stream made the mapping of streams to hardware queues erratic
        # (same code 181 ... 480 images/s depending on how many streams existed).  RADNET_COPY_STREAM=1 enables it.
        self.use_copy_stream = os.environ.get("RADNET_COPY_STREAM", "0") == "1"
        self.copy_stream = torch.cuda.Stream(device=self.dev) if self.use_copy_stream else None
        self.side_stream = torch.cuda.Stream(device=self.dev)
        self.ctx2 = L.Context(device_index, stream_handle=self.side_stream.cuda_stream)
        self.ctx2.check(self.lib.radnet_set_autotune(self.ctx2.h, self.autotune_mode), "set_autotune")
        self.ws2 = torch.empty(256 << 20, dtype=torch.uint8, device=self.dev)
        self.ctx2.check(self.lib.radnet_set_workspace(self.ctx2.h, self.ws2.data_ptr(), self.ws2.numel()), "set_workspace")
        self.head_stream = torch.cuda.Stream(device=self.dev)
        self.ctx3 = L.Context(device_index, stream_handle=self.head_stream.cuda_stream)
        self.ctx3.check(self.lib.radnet_set_autotune(self.ctx3.h, self.autotune_mode), "set_autotune")
        self.ws3 = torch.empty(256 << 20, dtype=torch.uint8, device=self.dev)
        self.ctx3.check(self.lib.radnet_set_workspace(self.ctx3.h, self.ws3.data_ptr(), self.ws3.numel()), "set_workspace")
        self._lanes = {"side": (self.ctx2, self.side_stream), "head": (self.ctx3, self.head_stream)}
        # further prefetch lanes: base forwards of different batches are independent GEMM chains, and the chip packs several
        # of them better than one (tools/concurrency_probe.py); TrainStep deals announced batches over them round-robin
        self.n_side_lanes = max(1, int(os.environ.get("RADNET_SIDE_LANES", "2")))
        self._extra_lanes = []
        for k in range(1, self.n_side_lanes):
            st = torch.cuda.Stream(device=self.dev)
            cx = L.Context(device_index, stream_handle=st.cuda_stream)
            cx.check(self.lib.radnet_set_autotune(cx.h, self.autotune_mode), "set_autotune")
            ws = torch.empty(256 << 20, dtype=torch.uint8, device=self.dev)
            cx.check(self.lib.radnet_set_workspace(cx.h, ws.data_ptr(), ws.numel()), "set_workspace")
            self._extra_lanes.append((st, cx, ws))
            self._lanes["side%d" % k] = (cx, st)
        if os.environ.get("RADNET_FORCE_CONFIG"):      # experiment: "tile_a,tile_b,slices" for every GEMM launch of every lane
            fa, fb, fs = (int(v) for v in os.environ["RADNET_FORCE_CONFIG"].split(","))
            for c in [self.ctx, self.ctx2, self.ctx3] + [e[1] for e in self._extra_lanes]:
                c.check(self.lib.radnet_force_config(c.h, fa, fb, fs), "force_config")
        for c in [self.ctx2, self.ctx3] + [e[1] for e in self._extra_lanes]:           # one table of measured launch choices for all lanes
            self.ctx.check(self.lib.radnet_share_tuning(c.h, self.ctx.h), "share_tuning")
        # Launch shapes tuned IN SITU for known workloads (tools/insitu_tune.py: against the throughput of the pipelined step, where a
        # launch shares the chip with the other lanes', instead of each launch alone): radnet_hip/tuned/*.txt, loaded before anything
        # is measured.  Keys are exact problem shapes; every other shape is measured on first use as before.  A table belongs to the
        # workload it was tuned in (file name prefix = TUNED_PREFIX of the engine class).  RADNET_SHIPPED_TUNING=0: off.
        self.shipped_tuning = []
        tuned_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned")
        if self.autotune_mode and os.environ.get("RADNET_SHIPPED_TUNING", "1") != "0" and os.path.isdir(tuned_dir):
            for name in sorted(os.listdir(tuned_dir)):
                if name.endswith(".txt") and name.startswith(self.TUNED_PREFIX):
                    self.load_tuning(os.path.join(tuned_dir, name))
                    self.shipped_tuning.append(name)
        self.anchor_wh = np.array([[(s * r[0]) / C_cfg.rpn_stride, (s * r[1]) / C_cfg.rpn_stride]
                                   for s in C_cfg.anchor_box_scales for r in C_cfg.anchor_box_ratios], dtype=np.float64)
        self.anchor_sizes = np.array(C_cfg.anchor_box_scales, dtype=np.float64)
        self.anchor_ratios = np.array(C_cfg.anchor_box_ratios, dtype=np.float64).reshape(-1, 2)
        self.regr_std = np.array(C_cfg.classifier_regr_std, dtype=np.float64)
        self.loss_scratch = torch.zeros(8, dtype=torch.float64, device=self.dev)
        self.rpn_losses = torch.zeros(2, dtype=torch.float32, device=self.dev)
        self.det_losses = torch.zeros(3, dtype=torch.float32, device=self.dev)

    # ------------------------------------------------------------------------------------------ layers
    def _build_layers(self):
        dev = self.dev
        self.base_layers = []           # execution order, tuples describing the program
        self.convs = {}
        frozen_elems = 0

        def conv(name, kh, cin, cout, stride=1, pad=0, ldw=None):
            c = ConvLayer(name, kh, cin, cout, stride, pad, ldw)
            self.convs[name] = c
            return c

        conv("conv1", 7, 4, 64, 2, 3)                       # image padded to 4 channels
        cin = 64
        for st, blocks, (f1, f2, f3), stride in RES_STAGES + (HEAD_STAGE,):
            if st == 5:
                cin = 1024
            for bl in blocks:
                b = "res%d%s_branch" % (st, bl)
                first = bl == "a"
                conv(b + "2a", 1, cin, f1, stride if first else 1)
                conv(b + "2b", 3, f1, f2, 1, 1)
                conv(b + "2c", 1, f2, f3)
                if first:
                    conv(b + "1", 1, cin, f3, stride)
                cin = f3
        conv("rpn_conv1", 3, 1024, 512, 1, 1)
        conv("rpn_heads", 1, 512, 5 * self.A, 1, 0, ldw=RPN_LD)

        # frozen base: plain tensors
        for name, c in self.convs.items():
            if name.startswith(("res5", "rpn")):
                continue
            c.weight = torch.zeros(c.kh * c.kh * c.cin, c.ldw, dtype=torch.float32, device=dev)
            c.scale = torch.ones(c.cout, dtype=torch.float32, device=dev)
            c.shift = torch.zeros(c.cout, dtype=torch.float32, device=dev)

        # RPN optimizer arena (rpn.py:41-64: no BN, bias straight into the epilogue shift)
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

        # classifier-head optimizer arena: all stage-5 kernels, then all stage-5 biases contiguous (so one
        # affine_vec launch refreshes every folded shift), then the dense heads
        self.head_arena = Arena(dev)
        head_names = [n for n in self.convs if n.startswith("res5")]
        self.head_conv_names = head_names
        for name in head_names:
            c = self.convs[name]
            self.head_arena.add(name + "/kernel", (c.kh * c.kh * c.cin, c.cout))
        self.head_bias_off = self.head_arena.n
        for name in head_names:
            self.head_arena.add(name + "/bias", (self.convs[name].cout,))
        self.head_bias_len = self.head_arena.n - self.head_bias_off
        self.head_arena.add("dense/kernel", (2048, self.dense_ld))
        self.head_arena.add("dense/bias", (self.dense_ld,))
        self.head_arena.finalize()
        self.head_scale = torch.ones(self.head_bias_len, dtype=torch.float32, device=dev)
        self.head_t0 = torch.zeros(self.head_bias_len, dtype=torch.float32, device=dev)
        self.head_shift = torch.zeros(self.head_bias_len, dtype=torch.float32, device=dev)
        for name in head_names:
            c = self.convs[name]
            c.weight, c.bias = self.head_arena.param(name + "/kernel"), self.head_arena.param(name + "/bias")
            c.dweight, c.dbias = self.head_arena.grad(name + "/kernel"), self.head_arena.grad(name + "/bias")
            o = self.head_arena.offsets[name + "/bias"][0] - self.head_bias_off
            c.scale = self.head_scale[o:o + c.cout]
            c.t0 = self.head_t0[o:o + c.cout]
            c.shift = self.head_shift[o:o + c.cout]
        self.dense_w, self.dense_b = self.head_arena.param("dense/kernel"), self.head_arena.param("dense/bias")
        self.dense_dw, self.dense_db = self.head_arena.grad("dense/kernel"), self.head_arena.grad("dense/bias")

    def sync_inference_filters(self):
        """Winograd filter transforms of the classifier's 3x3 convs (inference plans only) after the head weights changed."""
        if getattr(self, "_inference_filters_stale", False):
            self._inference_filters_stale = False
            self._refresh_winograd([n for n in self.INFERENCE_WINOGRAD_LAYERS if getattr(self.convs.get(n), "wino_u", None) is not None])

    def refresh_head_shift(self):
        """shift = scale * bias + t0 for every stage-5 conv (FixedBatchNormalization.py:59-85 folded).  A no-op right after adam()
        of the head arena, which refreshes the shifts in the same launch (radnet_adam_step_affine)."""
        if getattr(self, "_head_shift_fresh", False):
            self._head_shift_fresh = False
            return
        bias = self.head_arena.p[self.head_bias_off:self.head_bias_off + self.head_bias_len]
        self.ctx.call("radnet_affine_vec", self.head_shift, self.head_scale, bias, self.head_t0, C.c_int64(self.head_bias_len))

    # ------------------------------------------------------------------------------------------ weights
    def set_weights(self, W):
        """Load weights keyed by the reference's Keras layer names: conv/dense {'kernel','bias'} in HWIO /
        (in,out) layout, FixedBatchNormalization {'gamma','beta','mean','var'} (weight order of
        FixedBatchNormalization.py:26-51)."""
        dev = self.dev

        def t(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)

        def bn_name(conv_name):
            if conv_name == "conv1":
                return "bn_conv1"
            return conv_name.replace("res", "bn", 1)

        for name, c in self.convs.items():
            if name == "rpn_heads":
                kc, kr = W["rpn_out_class"], W["rpn_out_regress"]
                k = np.zeros((512, RPN_LD), np.float32)
                k[:, :self.A] = kc["kernel"].reshape(512, self.A)
                k[:, self.A:5 * self.A] = kr["kernel"].reshape(512, 4 * self.A)
                b = np.zeros(RPN_LD, np.float32)
                b[:self.A] = kc["bias"]
                b[self.A:5 * self.A] = kr["bias"]
                c.weight.copy_(t(k)); c.bias.copy_(t(b))
                continue
            kern = np.asarray(W[name]["kernel"], dtype=np.float32)
            bias = np.asarray(W[name]["bias"], dtype=np.float32)
            if name == "conv1":
                kern = np.concatenate([kern, np.zeros((7, 7, 1, 64), np.float32)], axis=2)
            c.weight.copy_(t(kern.reshape(-1, c.cout)))
            if name == "rpn_conv1":
                c.bias.copy_(t(bias))
                continue
            bn = W[bn_name(name)]
            s = (np.asarray(bn["gamma"], np.float64) / np.sqrt(np.asarray(bn["var"], np.float64) + BN_EPS))
            t0 = np.asarray(bn["beta"], np.float64) - np.asarray(bn["mean"], np.float64) * s
            c.scale.copy_(t(s))
            if c.bias is None:                      # frozen: fold the bias once
                c.shift.copy_(t(s * bias.astype(np.float64) + t0))
            else:
                c.bias.copy_(t(bias)); c.t0.copy_(t(t0))
        dc, dr = W["dense_class_%d" % self.nc], W["dense_regress_%d" % self.nc]
        k = np.zeros((2048, self.dense_ld), np.float32)
        k[:, :self.nc] = dc["kernel"]; k[:, self.nc:self.nc + self.nreg] = dr["kernel"]
        b = np.zeros(self.dense_ld, np.float32)
        b[:self.nc] = dc["bias"]; b[self.nc:self.nc + self.nreg] = dr["bias"]
        self.dense_w.copy_(t(k)); self.dense_b.copy_(t(b))
        self.refresh_head_shift()
        self._refresh_winograd()
        torch.cuda.synchronize(self.dev)

    def get_weights(self, names=None):
        """Trainable weights back in Keras layout (host numpy)."""
        torch.cuda.synchronize()                   # updates may be in flight on the head lane
        out = {}
        for name in ["rpn_conv1"] + self.head_conv_names:
            c = self.convs[name]
            out[name] = {"kernel": c.weight.detach().cpu().numpy().reshape(c.kh, c.kh, c.cin, c.cout).copy(),
                         "bias": c.bias.detach().cpu().numpy()[:c.cout].copy()}
        k = self.convs["rpn_heads"].weight.detach().cpu().numpy()
        b = self.convs["rpn_heads"].bias.detach().cpu().numpy()
        out["rpn_out_class"] = {"kernel": k[:, :self.A].reshape(1, 1, 512, self.A).copy(), "bias": b[:self.A].copy()}
        out["rpn_out_regress"] = {"kernel": k[:, self.A:5 * self.A].reshape(1, 1, 512, 4 * self.A).copy(), "bias": b[self.A:5 * self.A].copy()}
        k = self.dense_w.detach().cpu().numpy(); b = self.dense_b.detach().cpu().numpy()
        out["dense_class_%d" % self.nc] = {"kernel": k[:, :self.nc].copy(), "bias": b[:self.nc].copy()}
        out["dense_regress_%d" % self.nc] = {"kernel": k[:, self.nc:self.nc + self.nreg].copy(), "bias": b[self.nc:self.nc + self.nreg].copy()}
        return out

    # ------------------------------------------------------------------------------------------ descriptors
    def _desc(self, c, x, nb, h, w, y, relu=True, addend=None, act=None, act_cols=0):
        oh = (h + 2 * c.pad - c.kh) // c.stride + 1
        ow = (w + 2 * c.pad - c.kh) // c.stride + 1
        d = L.ConvDesc()
        d.x, d.w, d.y = x.data_ptr(), c.weight.data_ptr(), y.data_ptr()
        d.scale = c.scale.data_ptr() if c.scale is not None else None
        d.shift = c.shift.data_ptr() if c.shift is not None else None
        d.addend = addend.data_ptr() if addend is not None else None
        d.nb, d.h, d.w_, d.c, d.oh, d.ow = nb, h, w, c.cin, oh, ow
        d.kh = d.kw = c.kh
        d.stride, d.pad_t, d.pad_l, d.n = c.stride, c.pad, c.pad, c.ldw if c.name == "rpn_heads" else c.cout
        d.ldw, d.ldy, d.ld_add = c.ldw, d.n, d.n
        d.act = (1 if relu else 0) if act is None else act
        d.act_cols = act_cols
        return d, oh, ow

    # Winograd F(2x2,3x3) for the 3x3 layers where it measured faster than the direct implicit GEMM at 1000x600
    # (tools/winograd_timing.py: rpn_conv1 201 -> 119 us, res4x_2b 39 -> 34, res3x_2b 40 -> 37; stage 2 loses, the
    # trainable stage-5 convs would pay a filter transform per step).  Trainable layers re-transform after Adam.
    WINOGRAD_LAYERS = ("rpn_conv1",) + tuple("res%d%s_branch2b" % (st, bl) for st, bls in ((3, "abcd"), (4, "abcdef")) for bl in bls)
    if os.environ.get("RADNET_WINOGRAD_EXTRA"):        # experiment knob: comma-separated layer names
        WINOGRAD_LAYERS = WINOGRAD_LAYERS + tuple(os.environ["RADNET_WINOGRAD_EXTRA"].split(","))

    # Inference-only Winograd layers: the classifier's 3x3 convs on ALL RoIs of a tile (RADNet's predict path: 300 RoIs,
    # M = 14 700).  Their weights train, so in the train step a filter transform per update would eat the gain at M = 980;
    # at inference the transform is made once per weight change (sync_inference_filters).
    INFERENCE_WINOGRAD_LAYERS = tuple("res5%s_branch2b" % b for b in "abc")

    def _uses_winograd(self, c, inference=False):
        listed = c.name in self.WINOGRAD_LAYERS or ((inference or getattr(self, "head_train_wino", False)) and c.name in self.INFERENCE_WINOGRAD_LAYERS
                                                    and os.environ.get("RADNET_NO_INFERENCE_WINOGRAD", "0") != "1")
        return self.use_winograd and listed and c.kh == 3 and c.stride == 1 and c.pad == 1 and c.cin % 32 == 0

    # F(4x4,3x3) (36 GEMMs on a quarter of the tiles) where it measured faster than F(2x2,3x3) at 1000x600
    # (tools/winograd_timing.py; bench.py roofline.layers_3x3): RADNET_WINOGRAD_TILE=2 / 4 forces one form everywhere.
    WINOGRAD_F4_LAYERS = WINOGRAD_LAYERS

    def _wino_form(self, c):
        forced = os.environ.get("RADNET_WINOGRAD_TILE")
        if forced:
            return 4 if forced == "4" else 2
        return 4 if c.name in self.WINOGRAD_F4_LAYERS else 2

    def _refresh_winograd(self, names=None):
        """Filter transform U = G g G^T of the Winograd layers (all of them, or the named ones after a weight update)."""
        for name in (names if names is not None else self.WINOGRAD_LAYERS + tuple(n for n in self.INFERENCE_WINOGRAD_LAYERS
                                                                                   if getattr(self.convs.get(n), "wino_u", None) is not None
                                                                                   or getattr(self, "head_train_wino", False))):
            c = self.convs.get(name)
            if c is None or not self._uses_winograd(c, inference=True):
                continue
            if c.wino_u is None:
                c.wino_m = 4 if name in self.INFERENCE_WINOGRAD_LAYERS else self._wino_form(c)
                c.wino_u = torch.empty((c.wino_m + 2) ** 2, c.cin, c.cout, dtype=torch.float32, device=self.dev)
            self.ctx.call("radnet_winograd4_filter" if c.wino_m == 4 else "radnet_winograd_filter", c.weight, c.cin, c.cout, c.ldw, c.wino_u)

    def _fwd_op(self, c, x, nb, h, w, y, keep, relu=True, inference=False):
        """Forward op of conv `c` on x -> y: the direct implicit GEMM, or the Winograd form for the layers listed above."""
        d, oh, ow = self._desc(c, x, nb, h, w, y, relu=relu)
        if not self._uses_winograd(c, inference):
            return ("conv", d), d
        if c.wino_u is None:
            self._refresh_winograd([c.name])
        m = c.wino_m
        T = nb * ((h + m - 1) // m) * ((w + m - 1) // m)
        V = torch.empty((m + 2) ** 2, T, c.cin, dtype=torch.float32, device=self.dev)
        M = torch.empty((m + 2) ** 2, T, c.cout, dtype=torch.float32, device=self.dev)
        keep += [V, M]
        op = ("wino", (x.data_ptr(), nb, h, w, c.cin, c.cout, V.data_ptr(), c.wino_u.data_ptr(), M.data_ptr(), T,
                       c.scale.data_ptr() if c.scale is not None else None, c.shift.data_ptr() if c.shift is not None else None,
                       1 if relu else 0, y.data_ptr(), c.cout, m))
        return op, d

    def _plan_base(self, nb, H, W, slot=0):
        """Static launch list of nn_base for (nb,H,W) input: [(kind, payload)], output tensor F.
        `slot` selects an independent buffer set (one per image of a per-GPU mini-batch)."""
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
        c = self.convs["conv1"]
        oh, ow = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        y = buf(nb, oh, ow, 64)
        d, _, _ = self._desc(c, x, nb, H, W, y)
        ops.append(("conv", d))
        ph, pw = (oh - 3) // 2 + 1, (ow - 3) // 2 + 1
        p = buf(nb, ph, pw, 64)
        ops.append(("maxpool", (y, p, nb, oh, ow, 64, 3, 2)))
        cur, h, w = p, ph, pw
        for st, blocks, (f1, f2, f3), stride in RES_STAGES:
            for bl in blocks:
                b = "res%d%s_branch" % (st, bl)
                first = bl == "a"
                ca, cb, cc = self.convs[b + "2a"], self.convs[b + "2b"], self.convs[b + "2c"]
                oh, ow = ((h - 1) // ca.stride + 1, (w - 1) // ca.stride + 1)
                a = buf(nb, oh, ow, f1)
                da, _, _ = self._desc(ca, cur, nb, h, w, a)
                if first:
                    sc = buf(nb, oh, ow, f3)
                    ds, _, _ = self._desc(self.convs[b + "1"], cur, nb, h, w, sc, relu=False)
                else:
                    sc = cur
                pair = first and self.fwd_pair and not self.use_chain
                if pair:                           # branch2a and the shortcut read the same input: one launch where it measures faster
                    ops += [("conv_pair_first", da), ("conv_pair_second", ds)]
                else:
                    ops.append(("conv", da))
                bb = buf(nb, oh, ow, f2)
                op, _ = self._fwd_op(cb, a, nb, oh, ow, bb, keep); ops.append(op)
                if first and not pair:
                    ops.append(("conv", ds))
                out = buf(nb, oh, ow, f3)
                d, _, _ = self._desc(cc, bb, nb, oh, ow, out, relu=True, addend=sc); ops.append(("conv", d))
                cur, h, w = out, oh, ow
        if self.bneck_fuse and not self.use_chain and self.FROZEN_BASE_FUSION:
            ops = self._fuse_bottlenecks(ops)
        plan = dict(ops=ops, x=x, F=cur, fh=h, fw=w, keep=keep, nb=nb)
        if self.use_chain:
            self._chain_ops(plan, first=2)               # conv1 (4-channel stem) and the max-pool stay launches of their own
        self._plans[key] = plan
        return plan

    HEAD_WINO_WGRAD_MIN_ROIS = int(os.environ.get("RADNET_HEAD_WINO_WGRAD_MIN_ROIS", "40"))
    CROP_AHEAD = True                  # TrainStep may issue head_crop() ahead of head_forward(cropped=True)
    HEAD_TRAIN_WINOGRAD = True         # classifier 3x3 convs: Winograd forward in training too (engine_cont keeps the direct form)
    FROZEN_BASE_FUSION = True          # nn_base's stage 2 is frozen in every mode this engine runs (train.py, cont_train.py: stages 3-4 only)

    @staticmethod
    def _fuse_bottlenecks(ops):
        """3x3 conv (64 -> 64 channels) + the 1x1 expand on its output (+ the next block's 1x1 reduce on THAT output) -> one
        radnet_conv_bottleneck call: the 3x3's output and the expand's re-read never touch memory.  Only where the 3x3 output has no other
        reader in the list (it is not written any more) -- stage 2 of nn_base (resnet50.py:197-199)."""
        def conv(i):
            return ops[i][1] if i < len(ops) and ops[i][0] == "conv" else None

        def reads(d, ptr):
            return ptr in (getattr(d, "x", None), getattr(d, "addend", None))

        out, k = [], 0
        while k < len(ops):
            db, dc = conv(k), conv(k + 1)
            ok = (db is not None and dc is not None and db.kh == 3 and db.stride == 1 and db.n == 64 and db.c % 32 == 0 and not db.addend and db.act == 1
                  and dc.kh == 1 and dc.stride == 1 and dc.x == db.y and dc.c == 64 and dc.n % 64 == 0 and dc.act == 1)
            if ok:                                  # nobody else may read the tensor that is no longer written
                ok = not any(reads(p, db.y) for j, (kind, p) in enumerate(ops) if j != k + 1 and kind in ("conv", "conv_pair_first", "conv_pair_second"))
                ok = ok and not any(kind in ("wino", "wino_reuse") and p[0] == db.y for kind, p in ops)
            if not ok:
                out.append(ops[k])
                k += 1
                continue
            da = conv(k + 2)
            if da is not None and not (da.kh == 1 and da.stride == 1 and da.x == dc.y and da.c == dc.n and da.n == 64 and not da.addend and da.act == 1):
                da = None
            out += [("bneck_first", db), ("bneck_second", dc)] + ([("bneck_third", da)] if da is not None else [])
            k += 3 if da is not None else 2
        return out

    def _chain_ops(self, plan, first=0):
        """Replace plan['ops'][first:] by ONE persistent launch (radnet_chain_build, include/radnet_hip.h): the same
        convs / Winograd layers as work items with arrival counters instead of ~50 dependent launches.  Falls back to the
        launch list (and says so once) when the library refuses an op."""
        sub = plan["ops"][first:]
        arr = self._compile(sub)
        h = C.c_void_p()
        # the static deal needs every workgroup of every chain that runs at the same time resident (4 per CU = 1 024 slots): one
        # chain per prefetch lane may be in flight, and the RPN / classifier lanes' launches need room beside them
        lanes = max(1, getattr(self, "n_side_lanes", 1))
        cap = max(64, (4 * 256) // (lanes + 1))
        wgs = self.chain_wgs if self.chain_wgs > 0 else 512
        if wgs > cap:
            if not getattr(self, "_chain_clamped", False):
                self._chain_clamped = True
                import sys
                sys.stderr.write("radnet: RADNET_CHAIN_WGS=%d exceeds what %d concurrent chains can keep resident; using %d\n" % (wgs, lanes, cap))
            wgs = cap
        rc = self.lib.radnet_chain_build(self.ctx.h, C.cast(arr, C.c_void_p), len(sub), wgs, C.byref(h))
        if rc != 0:
            if not getattr(self, "_chain_warned", False):
                self._chain_warned = True
                import sys
                sys.stderr.write("radnet: chain refused (%s); the layer program runs launch by launch\n" % self.lib.radnet_last_error(self.ctx.h).decode())
            return
        plan["chain"] = h
        plan["chain_sub"] = sub                           # keeps the descriptors (and the compiled array) alive
        plan["ops"] = plan["ops"][:first] + [("chain", h)]
        self._chain_plans.append(plan)

    def release_slot(self, slot):
        """Drop every plan of buffer set `slot` (base / label plans keyed by it, the RPN / classifier plans built on its feature
        maps) with their buffers and hipGraphs.  TrainStep.validate runs on a buffer set of its own and gives it back here, so a
        validation pass neither keeps ~0.6 GB per panel size alive between epochs nor counts against the plan cache's limit."""
        OD = collections.OrderedDict
        fptrs, drop = set(), []
        for k in list(self._plans.keys()):
            if k[0] in ("base", "atgt", "rtgt") and k[-1] == slot:
                drop.append(k)
                if k[0] == "base":
                    fptrs.add(OD.__getitem__(self._plans, k)["F"].data_ptr())
        for k in list(self._plans.keys()):
            if k[0] in ("rpn", "head", "head_inf") and len(k) > 4 and (k[3] in fptrs or k[4] in fptrs):
                drop.append(k)
        for k in drop:
            plan = OD.__getitem__(self._plans, k)
            OD.__delitem__(self._plans, k)
            self._evict_plan(k, plan)

    def contexts(self):
        """Every native context of this engine (main + lanes)."""
        return [self.ctx, self.ctx2, self.ctx3] + [e[1] for e in self._extra_lanes]

    def check_chains(self):
        """Raise if any chain launch so far gave up waiting (its feature map was invalid).  Reads the chains' mapped host error
        words: no synchronisation -- called where the host has just waited for something downstream of the base forward
        (roi_targets_finish, TrainStep.flush / validate, base_forward of a plan that runs again)."""
        for plan in self._chain_plans:
            h = plan.get("chain")
            if h is not None:
                e = int(self.lib.radnet_chain_error(h))
                if e:
                    raise L.RadnetError("chain launch gave up waiting at work item %d: not every workgroup of its grid was resident "
                                        "(RADNET_CHAIN_WGS too large for the launches running beside it?); the feature maps it "
                                        "produced are invalid" % (e - 1))

    def chain_status(self, plan):
        """(last_error, runs, items, stages, executed flops, algorithmic flops) of a plan's chain; synchronises the lane."""
        e, r, n, st = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        fe, fa = C.c_double(), C.c_double()
        self.ctx.check(self.lib.radnet_chain_status(self.ctx.h, plan["chain"], C.byref(e), C.byref(r), C.byref(n), C.byref(st), C.byref(fe), C.byref(fa)), "chain_status")
        return e.value, r.value, n.value, st.value, fe.value, fa.value

    def _evict_plan(self, key, plan):
        """A plan leaves the cache: nothing may still be reading its buffers (lanes run ahead of the host), and the
        hipGraphs recorded from its launch lists go with it."""
        torch.cuda.synchronize(self.dev)
        if plan.get("chain") is not None:
            self.lib.radnet_chain_destroy(plan["chain"])
            plan["chain"] = None
            self._chain_plans = [p for p in self._chain_plans if p is not plan]
        lists = set()
        for v in plan.values():
            if isinstance(v, list):
                lists.add(id(v))
                lists.update(id(p[0]) for p in v if isinstance(p, tuple) and p and isinstance(p[0], list))     # bwd_parts
        for gk in [k for k in self._graphs if k[0] in lists]:
            del self._graphs[gk]
        for ck in [k for k in self._compiled if k[0] in lists]:
            del self._compiled[ck]

    def save_tuning(self, path):
        """Write the measured GEMM launch choices of this engine (one table for all lanes) to a text file."""
        self.ctx.check(self.lib.radnet_tune_save(self.ctx.h, path.encode()), "radnet_tune_save")

    def load_tuning(self, path):
        """Restore choices written by save_tuning: no trial launches for the shapes in the file."""
        self.ctx.check(self.lib.radnet_tune_load(self.ctx.h, path.encode()), "radnet_tune_load")

    @contextlib.contextmanager
    def lane(self, name):
        """Everything enqueued inside goes to lane `name`: its own HIP stream and its own context (split-K slabs,
        arrival counters, tuning tables -- two launches of one shape may be in flight at once).  'side' carries the next
        batch's frozen base forward, 'head' the classifier phase; the caller orders lanes with mark() / after()."""
        ctx, stream = self._lanes[name]
        prev = self.ctx
        self.ctx = ctx
        try:
            with torch.cuda.stream(stream):
                yield
        finally:
            self.ctx = prev

    @staticmethod
    def mark():
        """Event after everything enqueued so far on the current lane."""
        ev = torch.cuda.Event()
        ev.record()
        return ev

    @staticmethod
    def after(ev):
        """The current lane continues only once `ev` (mark() of any lane) has happened."""
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def _run(self, ops):
        """Run a layer program on the current lane.  Programs are static (fixed buffers, fixed descriptors), so after one eager run -- which
        autotunes every new GEMM shape and builds its work-unit tables -- the launch sequence is recorded into a
        hipGraph and replayed: ~20 us of host time per launch (ctypes + hipLaunchKernel) become one graph launch, and
        the host thread stays ahead of the GPU (tools/host_timeline.py).  Keyed by the program and the gradient
        write modes of its wgrad descriptors (set_accumulate edits them in place)."""
        key = (id(ops), id(self.ctx), tuple(p.dw_accumulate if kind == "wgrad" else p[-1] for kind, p in ops if kind in ("wgrad", "wino_wgrad")))
        ent = self._graphs.get(key)
        if ent is None:
            # first run of this program: every new GEMM shape is measured here, so launches run one at a time
            self._graphs[key] = [ops, None]          # holds `ops` so its id stays unique
            return self._run_eager(ops)
        if not self.use_graphs or self.ctx.timing_on or torch.cuda.is_current_stream_capturing():
            return self._run_eager(ops)
        if ent[1] is None:
            g = torch.cuda.CUDAGraph()
            prev = self.ctx.stream_handle
            try:
                # thread_local: other threads of the process (the collective backend's watchdog polling its events)
                # must not invalidate the capture
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
                    try:
                        self._run_eager(ops)
                    finally:
                        self.ctx.set_stream(prev)
                ent[1] = g
            except Exception as e:      # nothing recorded has executed: run this program eagerly, now and from here on
                import sys
                sys.stderr.write("radnet: hipGraph capture failed (%s); layer programs run eagerly\n" % (e,))
                self.ctx.set_stream(prev)
                self.use_graphs = False
                return self._run_eager(ops)
        ent[1].replay()

    def _compile(self, ops):
        """The launch list as a radnet_op array (include/radnet_hip.h): what radnet_program_run executes and what the composed
        entry points (radnet_rpn_forward / radnet_predict_tile / radnet_train_step) take.  Cached per list and gradient write
        mode (set_accumulate edits the Python descriptors in place; the array holds copies)."""
        key = (id(ops), tuple(p.dw_accumulate if kind == "wgrad" else (p[-1] if kind == "wino_wgrad" else p[6]) for kind, p in ops
                              if kind in ("wgrad", "wino_wgrad", "colsum")))
        ent = self._compiled.get(key)
        if ent is not None:
            return ent[0]
        arr = (L.Op * max(len(ops), 1))()
        ptr = lambda v: v.data_ptr() if hasattr(v, "data_ptr") else v
        paired = False
        for k, (kind, p) in enumerate(ops):
            o = arr[k]
            if paired:                              # the dgrad half of a pair: issued by the entry before (stays a no-op slot)
                paired = False
                o.kind = L.OP_NOP
                continue
            if kind == "conv_pair_first":           # branch2a + shortcut conv of a conv_block: one call (radnet_conv_fwd_pair), the second
                o.kind = L.OP_CONV_FWD_PAIR         # descriptor rides in the following NOP slot
                o.conv = p
            elif kind == "conv_pair_second":
                o.kind = L.OP_NOP
                o.conv = p
            elif kind == "bneck_first":             # 3x3 + 1x1 expand (+ next 1x1 reduce): one call, the other descriptors ride in the NOP slots behind
                o.kind = L.OP_CONV_BNECK
                o.conv = p
                o.i[0] = 1 if k + 2 < len(ops) and ops[k + 2][0] == "bneck_third" else 0
            elif kind in ("bneck_second", "bneck_third"):
                o.kind = L.OP_NOP
                o.conv = p
            elif kind in ("conv", "dgrad", "wgrad"):
                o.kind = {"conv": L.OP_CONV_FWD, "dgrad": L.OP_CONV_DGRAD, "wgrad": L.OP_CONV_WGRAD}[kind]
                o.conv = p
                # weight gradient and data gradient of one layer (same descriptor) -> one launch (radnet_conv_bwd)
                if kind == "wgrad" and k + 1 < len(ops) and ops[k + 1][0] == "dgrad" and ops[k + 1][1] is p:
                    o.kind = L.OP_CONV_BWD
                    paired = True
            elif kind == "maxpool":
                x, y, nb, hh, ww, c, kk, st = p
                o.kind = L.OP_MAXPOOL
                o.p[0], o.p[1] = ptr(x), ptr(y)
                o.i[0], o.i[1], o.i[2], o.i[3], o.i[4], o.i[5] = nb, hh, ww, c, kk, st
            elif kind == "colsum":
                g, m, n, ld, gs, out, acc = p
                o.kind = L.OP_COLSUM
                o.p[0], o.p[1], o.p[2] = ptr(g), ptr(gs), ptr(out)
                o.i[0], o.i[1], o.i[2], o.i[3] = m, n, ld, acc
            elif kind in ("wino", "wino_reuse"):
                x, nb, hh, ww, c, n, V, U, M, T, scale, shift, act, y, ldy, form = p
                o.kind = L.OP_WINO if kind == "wino" else L.OP_WINO_REUSE
                for j, v in enumerate((x, V, U, M, scale, shift, y)):
                    o.p[j] = ptr(v)
                for j, v in enumerate((nb, hh, ww, c, n, T, act, ldy, form)):
                    o.i[j] = v
            elif kind == "wino_wgrad":
                dy, nb, hh, ww, c, n, ld_dy, V, dZ, dU, T, dw, ldw, form, gscale, mode = p
                o.kind = L.OP_WINO_WGRAD
                for j, v in enumerate((dy, V, dZ, dU, dw, gscale)):
                    o.p[j] = ptr(v)
                for j, v in enumerate((nb, hh, ww, c, n, ld_dy, T, ldw, mode, form)):
                    o.i[j] = v
            elif kind == "scatter":
                src, nb, oh, ow, c, st, hh, ww, mask, dst = p
                o.kind = L.OP_SCATTER
                o.p[0], o.p[1], o.p[2] = ptr(src), ptr(mask), ptr(dst)
                for j, v in enumerate((nb, oh, ow, c, st, hh, ww)):
                    o.i[j] = v
            elif kind == "fill0":
                dst, nbytes = p
                o.kind = L.OP_FILL0
                o.p[0] = ptr(dst)
                o.i[0], o.i[1] = C.c_int32(nbytes & 0xFFFFFFFF).value, int(nbytes) >> 32
            elif kind == "relu_mask":
                g, act, n = p
                o.kind = L.OP_RELU_MASK
                o.p[0], o.p[1] = ptr(g), ptr(act)
                o.i[0], o.i[1] = C.c_int32(n & 0xFFFFFFFF).value, int(n) >> 32
            elif kind == "chain":
                o.kind = L.OP_CHAIN
                o.p[0] = p.value
            elif kind == "roi_bwd":
                dy, hh, ww, c, rois, r, ps, dF = p
                o.kind = L.OP_ROI_BWD
                o.p[0], o.p[1], o.p[2] = ptr(dy), ptr(rois), ptr(dF)
                for j, v in enumerate((hh, ww, c, r, ps)):
                    o.i[j] = v
            else:
                raise L.RadnetError("unknown op " + kind)
        self._compiled[key] = (arr, ops)              # holds `ops` so its id stays unique
        return arr

    def _run_eager(self, ops):
        """One native call per program (csrc/program.hip); Winograd layers are timed per layer inside it while ctx.timing is on
        (timing class 3, read back by bench.py's roofline leg)."""
        rc = self.lib.radnet_program_run(self.ctx.h, self._compile(ops), len(ops))
        if rc != 0:
            self.ctx.check(rc, "radnet_program_run")

    @staticmethod
    def _fuse_bias_grads(ops):
        """A bias-gradient column sum right after the wgrad of the same layer (same dy, pitch and scale) moves into
        that wgrad launch (radnet_conv_desc.db): 12 launches of ~7 us less per train step."""
        out = []
        for kind, p in ops:
            if kind == "colsum" and out and out[-1][0] == "wgrad":
                d = out[-1][1]
                g, m, n, ld, gs, db, _ = p
                if d.dy == g and d.ld_dy == ld and d.n == n and (d.gscale or None) == (gs or None) and d.nb * d.oh * d.ow == m:
                    d.db = db
                    continue
            out.append((kind, p))
        return out

    @staticmethod
    def set_accumulate(ops, flag, prezeroed=False):
        """Gradient write mode of a backward program: flag=False -> overwrite (self-contained; each split wgrad /
        colsum zeroes its own slice), flag=True -> add.  prezeroed=True with flag=False: the caller zeroed the whole
        arena with ONE memset, so the ~25 per-layer memsets disappear (dw_accumulate = 2)."""
        v = 1 if flag else (2 if prezeroed else 0)
        for kind, p in ops:
            if kind == "wgrad":
                p.dw_accumulate = v
            elif kind == "wino_wgrad":
                p[-1] = v
            elif kind == "colsum":
                p[6] = 1 if (flag or prezeroed) else 0

    # ------------------------------------------------------------------------------------------ forward pieces
    KERNEL_COPIES = os.environ.get("RADNET_KERNEL_COPIES", "1") == "1"

    def _copy(self, dst, src):
        """dst <- src (same byte count, contiguous; one side pinned host memory) on the current lane.  A copy kernel reading /
        writing the mapped host buffer instead of hipMemcpyAsync: the latter was measured to block the enqueuing host thread
        for 7-11 ms when the lane has a few layer programs in arrears (radnet_copy_bytes, include/radnet_hip.h)."""
        if not self.KERNEL_COPIES:
            dst.copy_(src, non_blocking=True)
            return
        n = dst.numel() * dst.element_size()
        if n != src.numel() * src.element_size() or not dst.is_contiguous() or not src.is_contiguous():
            raise L.RadnetError("engine._copy: %d != %d bytes, or a strided side" % (n, src.numel() * src.element_size()))
        self.ctx.call("radnet_copy_bytes", dst, src, C.c_uint64(n))

    def upload_image(self, img_bgr_u8, slot=0):
        """uint8 BGR HWC host image -> preprocessed fp32 NHWC(4) on device (RADNet.py:83-87)."""
        H, W = img_bgr_u8.shape[:2]
        plan = self._plan_base(1, H, W, slot)
        # through a pinned staging buffer: a pageable H2D copy blocks the host thread for ~1 ms per 1.8 MB panel,
        # the pinned one is an asynchronous DMA behind a 0.1 ms memcpy (a pinned torch tensor skips the memcpy too)
        if "raw" not in plan:
            plan["raw"] = torch.empty(H, W, 3, dtype=torch.uint8, device=self.dev)
            plan["h_raw"] = torch.empty(H, W, 3, dtype=torch.uint8).pin_memory()
            plan["raw_free"] = torch.cuda.Event()       # staging buffer drained
            plan["raw_read"] = torch.cuda.Event()       # device panel consumed by the preprocess kernel
            plan["raw_read"].record()
        if isinstance(img_bgr_u8, torch.Tensor) and img_bgr_u8.is_pinned():
            src = img_bgr_u8
        else:
            plan["raw_free"].synchronize()         # the previous DMA out of the staging buffer has finished
            src = plan["h_raw"]
            np.copyto(src.numpy(), img_bgr_u8)     # NumPy's single-threaded memcpy (a torch CPU copy_ fans out to OpenMP)
        if self.use_copy_stream:
            # the DMA runs on a copy stream: enqueued ahead (TrainStep prefetches the next batch), it overlaps the
            # current step's kernels instead of taking ~40 us of the compute queue (+1 % images/s, long A/B runs)
            cs = self.copy_stream
            cs.wait_event(plan["raw_read"])
            with torch.cuda.stream(cs):
                plan["raw"].copy_(src, non_blocking=True)
                plan["raw_free"].record()
            torch.cuda.current_stream().wait_event(plan["raw_free"])
        else:
            self._copy(plan["raw"], src)
            plan["raw_free"].record()
        self.ctx.call("radnet_preprocess_bgr", plan["raw"], H, W, 4, plan["x"])
        plan["raw_read"].record()
        return plan

    def upload_images(self, imgs, slot=0):
        """Per-GPU mini-batch as ONE layer program (BASELINE cfg 4: per-GPU batch 2): the nb images (same size) land in one
        [nb][H][W][4] input tensor, so every base / RPN GEMM runs once with M = nb * H' * W' rows instead of nb times with
        H' * W' -- at batch 1 the launches are too small to fill 256 CUs, doubling M halves the fixed cost
        per image.  Same pinned-staging path per image as upload_image."""
        nb = len(imgs)
        H, W = imgs[0].shape[:2]
        plan = self._plan_base(nb, H, W, slot)
        if "raws" not in plan:
            plan["raws"] = [torch.empty(H, W, 3, dtype=torch.uint8, device=self.dev) for _ in range(nb)]
            plan["h_raws"] = [torch.empty(H, W, 3, dtype=torch.uint8).pin_memory() for _ in range(nb)]
            plan["raw_free"] = torch.cuda.Event()
            plan["raw_free"].record()
        plan["raw_free"].synchronize()             # the previous DMAs out of the staging buffers have finished
        for i, img in enumerate(imgs):
            if img.shape[:2] != (H, W):
                raise L.RadnetError("upload_images: images of one mini-batch must share their size")
            np.copyto(plan["h_raws"][i].numpy(), img)
            self._copy(plan["raws"][i], plan["h_raws"][i])
        plan["raw_free"].record()
        for i in range(nb):
            self.ctx.call("radnet_preprocess_bgr", plan["raws"][i], H, W, 4, plan["x"][i])
        return plan

    def upload_preprocessed(self, X):
        """X: (1,H,W,3) fp32 already preprocessed by the caller (Keras-style model.predict input)."""
        _, H, W, _ = X.shape
        plan = self._plan_base(1, H, W)
        x4 = np.zeros((1, H, W, 4), np.float32)
        x4[..., :3] = X
        plan["x"].copy_(torch.from_numpy(x4).to(self.dev))
        return plan

    def base_forward(self, plan):
        self._run(plan["ops"])
        return plan["F"]

    def _plan_rpn(self, fh, fw, F, nb=1):
        """nb > 1: the RPN of a per-GPU mini-batch as one program (rows of image i are [i*M, (i+1)*M)); losses and
        proposals stay per image (rpn_image), the backward GEMMs sum the gradient over the nb images by construction."""
        key = ("rpn", fh, fw, F.data_ptr(), nb)
        if key in self._plans:
            return self._plans[key]
        dev = self.dev
        M1 = fh * fw
        M = nb * M1
        hbuf = torch.empty(nb, fh, fw, 512, dtype=torch.float32, device=dev)
        pred = torch.empty(M, RPN_LD, dtype=torch.float32, device=dev)
        c1, ch = self.convs["rpn_conv1"], self.convs["rpn_heads"]
        wino_keep = []
        op1, d1 = self._fwd_op(c1, F, nb, fh, fw, hbuf, wino_keep, relu=True)
        d2, _, _ = self._desc(ch, hbuf, nb, fh, fw, pred, act=2, act_cols=self.A)
        # backward
        dz = torch.zeros(M, RPN_LD, dtype=torch.float32, device=dev)
        dh = torch.empty(M, 512, dtype=torch.float32, device=dev)
        # heads: wgrad + dgrad (masked by relu of rpn_conv1)
        b2 = L.ConvDesc.from_buffer_copy(d2)
        b2.dy, b2.ld_dy, b2.gscale = dz.data_ptr(), RPN_LD, None
        b2.dw, b2.dw_accumulate = ch.dweight.data_ptr(), 1
        b2.dx, b2.ld_dx, b2.dx_add, b2.dx_mask, b2.ld_dx_mask = dh.data_ptr(), 512, None, hbuf.data_ptr(), 512
        b1 = L.ConvDesc.from_buffer_copy(d1)
        b1.dy, b1.ld_dy, b1.gscale = dh.data_ptr(), 512, None
        b1.dw, b1.dw_accumulate = c1.dweight.data_ptr(), 1
        if op1[0] == "wino" and self.wino_wgrad:
            # dW in the Winograd domain on the V of the forward pass (F does not change between forward and backward)
            V, T = wino_keep[0], wino_keep[0].shape[1]
            P = (c1.wino_m + 2) ** 2
            dZ = torch.empty(P, T, c1.cout, dtype=torch.float32, device=dev)
            dU = torch.empty(P, c1.cin, c1.cout, dtype=torch.float32, device=dev)
            wino_keep += [dZ, dU]
            wg1 = ("wino_wgrad", [dh.data_ptr(), nb, fh, fw, c1.cin, c1.cout, 512, V.data_ptr(), dZ.data_ptr(), dU.data_ptr(), T,
                                  c1.dweight.data_ptr(), c1.ldw, c1.wino_m, None, 1])
        else:
            wg1 = ("wgrad", b1)
        bwd = [("wgrad", b2), ("colsum", [dz.data_ptr(), M, RPN_LD, RPN_LD, None, ch.dbias.data_ptr(), 1]),
               ("dgrad", b2), wg1, ("colsum", [dh.data_ptr(), M, 512, 512, None, c1.dbias.data_ptr(), 1])]      # M = all nb images
        bwd = self._fuse_bias_grads(bwd)
        ws_bytes = int(self.lib.radnet_proposals_ws_bytes(M1 * self.A))
        # the re-prediction after Adam #1 (train.py:291) sees the same feature map: its input transform is already in V
        refwd = [("wino_reuse", op1[1]) if op1[0] == "wino" else op1, ("conv", d2)]
        plan = dict(fwd=[op1, ("conv", d2)], refwd=refwd, bwd=bwd, b1=b1, h=hbuf, pred=pred, dz=dz, dh=dh, M=M1, nb=nb, fh=fh, fw=fw,
                    wino_keep=wino_keep,
                    prop_ws=torch.empty(ws_bytes, dtype=torch.uint8, device=dev),
                    R=torch.zeros(1024, 4, dtype=torch.int64, device=dev), Rp=torch.zeros(1024, dtype=torch.float32, device=dev),
                    Rn=torch.zeros(1, dtype=torch.int32, device=dev))
        if nb > 1:          # per-image views: scores / gradients of image i, its own proposal buffers
            plan["images"] = [dict(pred=pred[i * M1:(i + 1) * M1], dz=dz[i * M1:(i + 1) * M1], M=M1, fh=fh, fw=fw,
                                   prop_ws=torch.empty(ws_bytes, dtype=torch.uint8, device=dev),
                                   R=torch.zeros(1024, 4, dtype=torch.int64, device=dev), Rp=torch.zeros(1024, dtype=torch.float32, device=dev),
                                   Rn=torch.zeros(1, dtype=torch.int32, device=dev)) for i in range(nb)]
        self._plans[key] = plan
        return plan

    def rpn_forward(self, bplan):
        rp = self._plan_rpn(bplan["fh"], bplan["fw"], bplan["F"], bplan.get("nb", 1))
        self._run(rp["fwd"])
        return rp

    @staticmethod
    def rpn_image(rp, i):
        """View of image i of a (possibly batched) RPN plan: what proposals() and the capture hooks read."""
        return rp["images"][i] if rp.get("nb", 1) > 1 else rp

    def rpn_loss_image(self, rp, i, y_cls, y_regr, loss_out=None):
        """Losses of ONE image of a batched RPN plan (every image is an independent reference step with its own
        normalisers, losses.py:16-66) + its rows of the pre-activation gradient; rpn_backward_batched() then runs the
        backward GEMMs once over all images.  y_cls None: the image was dropped by the labeller -- zero gradient rows."""
        v = self.rpn_image(rp, i)
        if y_cls is None:
            self.ctx.call("radnet_fill_zero", v["dz"], C.c_uint64(v["M"] * RPN_LD * 4))
            return
        self.ctx.call("radnet_rpn_loss", v["pred"], RPN_LD, y_cls, y_regr, v["M"], self.A, self.bce_mode, v["dz"], RPN_LD,
                      self.rpn_losses if loss_out is None else loss_out, self.loss_scratch)

    def rpn_backward_batched(self, rp):
        self._run(rp["bwd"])

    def rpn_losses_only(self, rp, y_cls, y_regr, loss_out=None):
        """model_rpn.test_on_batch (train.py:494): the two RPN losses of the forward pass in `rp`, nothing differentiated,
        no gradient arena touched (the logit gradient the kernel also writes goes to the plan's scratch)."""
        self.ctx.call("radnet_rpn_loss", rp["pred"], RPN_LD, y_cls, y_regr, rp["M"], self.A, self.bce_mode, rp["dz"], RPN_LD,
                      self.rpn_losses if loss_out is None else loss_out, self.loss_scratch)

    def head_losses_only(self, hp, loss_out=None):
        """model_classifier.test_on_batch (train.py:513) on a head plan whose RoIs and targets are packed: forward in inference
        mode, the two detector losses + accuracy, nothing differentiated."""
        lo = self.det_losses if loss_out is None else loss_out
        self.head_forward(hp, training=True, loss_out=lo)          # `training` only selects the fused tail here (no Dropout in this head)
        if not hp.get("_tail_fused"):
            self.ctx.call("radnet_det_loss", hp["pcls"], hp["pregr"], hp["y1"], hp["y2"], hp["R"], self.nc, self.nreg, hp["dz"], lo)

    def rpn_backward(self, rp, y_cls, y_regr, loss_out=None):
        """losses (losses.py:16-66) + gradients of rpn_conv1 / fused heads accumulated into the RPN grad arena."""
        self.ctx.call("radnet_rpn_loss", rp["pred"], RPN_LD, y_cls, y_regr, rp["M"], self.A, self.bce_mode, rp["dz"], RPN_LD,
                      self.rpn_losses if loss_out is None else loss_out, self.loss_scratch)
        self._run(rp["bwd"])

    def adam(self, arena, grad_scale=1.0, zero_grad=True):
        """One Keras-2 Adam step over the arena.  zero_grad: the gradient arena is cleared in the same pass, so the
        next step's backward accumulates into zeros without a memset (arenas start zeroed, Arena.finalize)."""
        arena.t += 1
        is_head = arena is getattr(self, "head_arena", None)
        fused = (is_head and getattr(self, "head_bias_len", 0) > 0 and self.head_bias_off % 4 == 0
                 and self.head_bias_len % 4 == 0 and os.environ.get("RADNET_NO_ADAM_AFFINE", "0") != "1")
        wino = self._head_adam_wino() if is_head and getattr(self, "head_train_wino", False) else None
        if wino is not None:      # Adam #2 + folded shifts + the Winograd filters of the classifier's 3x3 convs, one launch
            arr, n_l = wino
            self.ctx.check(self.lib.radnet_adam_step_fused(
                self.ctx.h, arena.p.data_ptr(), arena.g.data_ptr(), arena.m.data_ptr(), arena.v.data_ptr(), C.c_int64(arena.n), arena.t, C.c_float(self.lr),
                C.c_float(0.9), C.c_float(0.999), C.c_float(1e-7), C.c_float(grad_scale), 1 if zero_grad else 0,
                C.c_int64(self.head_bias_off if fused else 0), C.c_int64(self.head_bias_len if fused else 0),
                self.head_scale.data_ptr() if fused else None, self.head_t0.data_ptr() if fused else None, self.head_shift.data_ptr() if fused else None,
                arr, n_l), "radnet_adam_step_fused")
            if fused:
                self._head_shift_fresh = True
            return
        if fused:       # Adam #2 and the refresh of the classifier convs' folded shifts as one launch (round 4: one launch fewer on the head lane)
            self.ctx.call("radnet_adam_step_affine", arena.p, arena.g, arena.m, arena.v, C.c_int64(arena.n), arena.t, C.c_float(self.lr),
                          C.c_float(0.9), C.c_float(0.999), C.c_float(1e-7), C.c_float(grad_scale), 1 if zero_grad else 0,
                          C.c_int64(self.head_bias_off), C.c_int64(self.head_bias_len), self.head_scale, self.head_t0, self.head_shift)
            self._head_shift_fresh = True
        else:
            self.ctx.call("radnet_adam_step", arena.p, arena.g, arena.m, arena.v, C.c_int64(arena.n), arena.t, C.c_float(self.lr),
                          C.c_float(0.9), C.c_float(0.999), C.c_float(1e-7), C.c_float(grad_scale), 1 if zero_grad else 0)
        if arena is self.rpn_arena:
            self._refresh_winograd(["rpn_conv1"])          # its forward runs on the transformed filter
        elif arena is getattr(self, "head_arena", None):
            self._inference_filters_stale = True           # transformed lazily, by the next inference pass (if any)
            if getattr(self, "head_train_wino", False):    # (the fused pass above could not be used: dense-kernel check failed)
                self._refresh_winograd(list(self.INFERENCE_WINOGRAD_LAYERS))
                self._inference_filters_stale = False

    def _wino_wgrad_op(self, c, V, dy, ld_dy, nb, h, w, shared):
        """Weight gradient of a Winograd layer in the transformed domain, on the V its forward pass left: dy transform, one batched
        reduction over the tiles, filter-gradient transform (the rpn_conv1 path; bias gradient = the colsum that follows)."""
        T = V.shape[1]
        P = (c.wino_m + 2) ** 2
        key = (P, T, c.cin, c.cout)
        if key not in shared:
            shared[key] = (torch.empty(P, T, c.cout, dtype=torch.float32, device=self.dev), torch.empty(P, c.cin, c.cout, dtype=torch.float32, device=self.dev))
        dZ, dU = shared[key]
        return ("wino_wgrad", [dy.data_ptr(), nb, h, w, c.cin, c.cout, ld_dy, V.data_ptr(), dZ.data_ptr(), dU.data_ptr(), T, c.dweight.data_ptr(), c.ldw,
                               c.wino_m, c.scale.data_ptr() if c.scale is not None else None, 1])

    def _head_adam_wino(self):
        """(radnet_adam_wino[], n) of the classifier's 3x3 kernels for radnet_adam_step_fused, or None when one of them is not a dense
        [3][3][c][n] slice of the head arena (then Adam #2 is followed by the filter transforms as launches of their own)."""
        if self._adam_wino is None:
            ent = []
            for name in self.INFERENCE_WINOGRAD_LAYERS:
                c = self.convs.get(name)
                if c is None or c.kh != 3 or c.ldw != c.cout or c.cout % 4:
                    ent = None
                    break
                if c.wino_u is None:
                    self._refresh_winograd([name])
                off, size = self.head_arena.offsets[name + "/kernel"]
                if c.wino_m != 4 or off % 4 or size != 9 * c.cin * c.cout or c.weight.data_ptr() != self.head_arena.p[off:].data_ptr():
                    ent = None
                    break
                ent.append((off, c.cin, c.cout, c.wino_u.data_ptr()))
            if ent:
                arr = (L.AdamWino * len(ent))()
                for k, (off, ci, co, u) in enumerate(ent):
                    arr[k].off, arr[k].c, arr[k].n, arr[k].u = off, ci, co, u
                self._adam_wino = (arr, len(ent))
            else:
                self._adam_wino = False
        return self._adam_wino or None

    def zero_grads(self, arena):
        self.ctx.call("radnet_fill_zero", arena.g, C.c_uint64(arena.n * 4))

    def proposals(self, rp, overlap_thresh=0.7, max_boxes=300, use_regr=True):
        """rpn.rpn_to_roi on device; returns (R int64 [max][4] device, count device int32)."""
        awh = self.anchor_wh.ctypes.data_as(C.POINTER(C.c_double))
        rc = self.lib.radnet_rpn_to_roi(self.ctx.h, rp["pred"].data_ptr(), RPN_LD, rp["fh"], rp["fw"], self.A, awh,
                                        float(self.C.std_scaling), 1 if use_regr else 0, float(overlap_thresh), int(max_boxes),
                                        rp["R"].data_ptr(), rp["Rp"].data_ptr(), rp["Rn"].data_ptr(), rp["prop_ws"].data_ptr())
        self.ctx.check(rc, "radnet_rpn_to_roi")
        return rp["R"], rp["Rn"]

    # ------------------------------------------------------------------------------------------ classifier head
    def _plan_head(self, R, fh, fw, F, training=True, groups=1):
        """training=False (DetectorModel.predict / RADNet's tile path): forward buffers only -- the backward program's
        gradient buffers (~1.4 GB at R = 300) are not allocated for a plan that never differentiates.
        groups > 1 (per-GPU mini-batch): F is [groups][fh][fw][C] and the R RoIs are groups x R/groups, group g cropped from
        feature map g; stage 5 then runs ONCE over all R RoIs (GEMM M = R * 49), losses stay per image."""
        key = ("head" if training else "head_inf", R, fh, fw, F.data_ptr(), groups)
        if key in self._plans:
            return self._plans[key]
        if R % groups:
            raise L.RadnetError("head plan: %d RoIs do not split into %d groups" % (R, groups))
        dev = self.dev
        keep = []

        def buf(*shape):
            b = torch.empty(shape, dtype=torch.float32, device=dev)
            keep.append(b)
            return b

        rois = buf(R, 4)
        pooled = buf(R, 14, 14, 1024)
        fwd, blocks = [], []
        cur, h, w = pooled, 14, 14
        st, bls, (f1, f2, f3), stride = HEAD_STAGE
        for bl in bls:
            b = "res5%s_branch" % bl
            first = bl == "a"
            ca, cb, cc = self.convs[b + "2a"], self.convs[b + "2b"], self.convs[b + "2c"]
            oh, ow = ((h - 1) // ca.stride + 1, (w - 1) // ca.stride + 1)
            a = buf(R, oh, ow, f1); da, _, _ = self._desc(ca, cur, R, h, w, a)
            ds = None
            if first:
                sc = buf(R, oh, ow, f3); ds, _, _ = self._desc(self.convs[b + "1"], cur, R, h, w, sc, relu=False)
            else:
                sc = cur
            if first and self.fwd_pair:            # branch2a and the shortcut read the same input: one launch where it measures faster
                fwd += [("conv_pair_first", da), ("conv_pair_second", ds)]
            else:
                fwd.append(("conv", da))
            bb = buf(R, oh, ow, f2)
            n_keep = len(keep)
            op_b, db = self._fwd_op(cb, a, R, oh, ow, bb, keep, relu=True, inference=not training)
            fwd.append(op_b)
            # V of the forward pass, for the weight gradient in the Winograd domain -- where the pass is large enough for it to pay: measured
            # 554.6 against 579.9 images/s at 20 RoIs (four launches instead of a share of the paired one, on the saturated lane), 667.7 against
            # 654.5 at 2 x 20 (per-GPU batch 2)
            wino_v = keep[n_keep] if (training and op_b[0] == "wino" and self.head_wino_wgrad and R >= self.HEAD_WINO_WGRAD_MIN_ROIS) else None
            if first and not self.fwd_pair:
                fwd.append(("conv", ds))
            out = buf(R, oh, ow, f3); dc, _, _ = self._desc(cc, bb, R, oh, ow, out, relu=True, addend=sc); fwd.append(("conv", dc))
            blocks.append(dict(first=first, x=cur, a=a, b=bb, out=out, da=da, db=db, dc=dc, ds=ds, names=(b + "2a", b + "2b", b + "2c", b + "1"), wino_v=wino_v))
            cur, h, w = out, oh, ow
        M = R * h * w
        feat = buf(R, 2048)
        pcls, pregr = buf(R, self.nc), buf(R, self.nreg)
        y1, y2 = buf(R, self.nc), buf(R, 2 * self.nreg)
        tail_scratch = torch.zeros(int(self.lib.radnet_head_tail_scratch_bytes(R)), dtype=torch.uint8, device=dev)
        keep.append(tail_scratch)
        if not training:
            plan = dict(R=R, rois=rois, pooled=pooled, fwd=fwd, blocks=blocks, y5=cur, hw=h * w, M=M, feat=feat, pcls=pcls, pregr=pregr,
                        F=F, fh=fh, fw=fw, keep=keep, groups=groups, tail_scratch=tail_scratch)
            self._plans[key] = plan
            return plan
        dz = buf(R, self.nc + self.nreg)
        dfeat = buf(R, 2048)
        # backward program (train.py mode: nothing flows below the RoI crop, the base is frozen)
        bwd = []
        shared_wg = {}                         # dZ / dU scratch of the Winograd-domain weight gradients: one set for layers of one shape
        bwd_parts = []                         # the same program cut per block, last block first (bucketed gradient exchange)
        g_out = buf(M, f3)                     # gradient w.r.t. the last block's output, ReLU mask applied
        g_first = g_out
        for bi in range(len(blocks) - 1, -1, -1):
            B = blocks[bi]
            part_from = len(bwd)
            ca, cb, cc = (self.convs[n] for n in B["names"][:3])
            g_b, g_a = buf(M, f2), buf(M, f1)

            def bdesc(fdesc, conv, dy, ld_dy, dx=None, ld_dx=0, dx_add=None, dx_mask=None):
                d = L.ConvDesc.from_buffer_copy(fdesc)
                d.dy, d.ld_dy, d.gscale = dy.data_ptr(), ld_dy, conv.scale.data_ptr()
                d.dw, d.dw_accumulate = conv.dweight.data_ptr(), 1
                if dx is not None:
                    d.dx, d.ld_dx = dx.data_ptr(), ld_dx
                    d.dx_add = dx_add.data_ptr() if dx_add is not None else None
                    d.ld_dx_add = ld_dx
                    d.dx_mask = dx_mask.data_ptr() if dx_mask is not None else None
                    d.ld_dx_mask = ld_dx
                return d

            B["g_out"], B["g_a"], B["g_b"] = g_out, g_a, g_b      # gradients w.r.t. this block's output / 2a / 2b outputs
            dC = bdesc(B["dc"], cc, g_out, f3, g_b, f2, None, B["b"])
            bwd += [("wgrad", dC), ("colsum", [g_out.data_ptr(), M, f3, f3, cc.scale.data_ptr(), cc.dbias.data_ptr(), 1]), ("dgrad", dC)]
            dB = bdesc(B["db"], cb, g_b, f2, g_a, f1, None, B["a"])
            wg_b = self._wino_wgrad_op(cb, B.get("wino_v"), g_b, f2, R, B["db"].oh, B["db"].ow, shared_wg) if B.get("wino_v") is not None else None
            bwd += [wg_b or ("wgrad", dB), ("colsum", [g_b.data_ptr(), M, f2, f2, cb.scale.data_ptr(), cb.dbias.data_ptr(), 1]), ("dgrad", dB)]
            if B["first"]:
                dA = bdesc(B["da"], ca, g_a, f1)
                bwd += [("wgrad", dA), ("colsum", [g_a.data_ptr(), M, f1, f1, ca.scale.data_ptr(), ca.dbias.data_ptr(), 1])]
                cs = self.convs[B["names"][3]]
                dS = bdesc(B["ds"], cs, g_out, f3)
                bwd += [("wgrad", dS), ("colsum", [g_out.data_ptr(), M, f3, f3, cs.scale.data_ptr(), cs.dbias.data_ptr(), 1])]
            else:
                g_prev = buf(M, f3)            # grad w.r.t. this block's input = previous block's output (post-ReLU)
                dA = bdesc(B["da"], ca, g_a, f1, g_prev, f3, g_out, B["x"])
                bwd += [("wgrad", dA), ("colsum", [g_a.data_ptr(), M, f1, f1, ca.scale.data_ptr(), ca.dbias.data_ptr(), 1]), ("dgrad", dA)]
                g_out = g_prev
            # gradient slice of this block's kernels: contiguous in the arena (kernels are laid out in forward order)
            k0 = self.head_arena.offsets[B["names"][0] + "/kernel"][0]
            kl, sl = self.head_arena.offsets[B["names"][3 if B["first"] else 2] + "/kernel"]
            bwd_parts.append((self._fuse_bias_grads(bwd[part_from:]), (k0, kl + _pad4(sl))))
        bwd = [op for part, _ in bwd_parts for op in part]
        cover = sorted(sl for _, sl in bwd_parts)
        if cover[0][0] != 0 or cover[-1][1] != self.head_bias_off or any(a[1] != b[0] for a, b in zip(cover, cover[1:])):
            raise RuntimeError("head gradient buckets do not tile the kernel part of the arena: %r" % (cover,))
        if [sl for _, sl in bwd_parts] != self.head_exchange_slices():
            raise RuntimeError("head backward parts and head_exchange_slices() disagree")
        live = torch.ones(groups, dtype=torch.int32, device=dev)
        keep.append(live)
        plan = dict(R=R, rois=rois, pooled=pooled, fwd=fwd, bwd=bwd, bwd_parts=bwd_parts, blocks=blocks, y5=cur, hw=h * w, M=M, feat=feat, pcls=pcls,
                    tail_scratch=tail_scratch, live=live, live_host=[1] * groups, wg_scratch=shared_wg,
                    pregr=pregr, y1=y1, y2=y2, dz=dz, dfeat=dfeat, g_last=g_first, F=F, fh=fh, fw=fw, keep=keep, groups=groups)
        self._plans[key] = plan
        return plan

    def head_exchange_slices(self):
        """Slices [lo, hi) of the flat head gradient arena in the order head_backward(on_part=...) completes them (last
        block first), or None when this engine's head backward is not cut per block.  Every rank of a data-parallel job
        must issue the same sequence of collectives, also a rank whose images all skipped their classifier step."""
        if not hasattr(self, "_head_slices"):
            self._head_slices = None
            names = getattr(self, "head_conv_names", [])
            blocks = []
            for n in names:                     # res5a_branch2a ... -> block prefix 'res5a'
                b = n.split("_")[0]
                if not blocks or blocks[-1][0] != b:
                    blocks.append([b, n, n])
                blocks[-1][2] = n
            if blocks and all(n.startswith("res5") for n in names):
                sl = []
                for _, first, last in blocks:
                    o0 = self.head_arena.offsets[first + "/kernel"][0]
                    ol, szl = self.head_arena.offsets[last + "/kernel"]
                    sl.append((o0, ol + _pad4(szl)))
                self._head_slices = sl[::-1]
        return self._head_slices

    def head_crop(self, hp):
        """RoiPoolingConv (RoiPoolingConv.py:48-88) of the plan's RoIs: the first launch of the classifier pass.  TrainStep issues it on the
        RPN lane right after the host's sample selection (head_forward(cropped=True) then starts at stage 5)."""
        G = hp.get("groups", 1)
        if G == 1:
            self.ctx.call("radnet_roi_resize_fwd", hp["F"], hp["fh"], hp["fw"], 1024, hp["rois"], hp["R"], 14, hp["pooled"])
        else:
            rg = hp["R"] // G
            for g in range(G):                  # RoIs of image g crop feature map g
                self.ctx.call("radnet_roi_resize_fwd", hp["F"][g], hp["fh"], hp["fw"], 1024, hp["rois"][g * rg:], rg, 14, hp["pooled"][g * rg:])

    def head_forward(self, hp, training=False, loss_out=None, group_live=None, cropped=False):
        """classifier_layer forward (`training` only matters for the VGG16 head's Dropout).  loss_out (training plans whose
        targets y1 / y2 are already packed): the detector losses (row g of loss_out for group g: cls, regr, accuracy) and the
        gradient w.r.t. the logits are computed in the same launch as the dense heads (csrc/head_tail.hip); head_backward
        then skips its loss pass.  group_live: per group 0/1 -- 0 = that image takes no classifier step (zero gradient rows)."""
        G = hp.get("groups", 1)
        if "live" not in hp:
            self.sync_inference_filters()
        if not cropped:
            self.head_crop(hp)
        self._run(hp["fwd"])
        hp["_tail_fused"] = False
        if self.fuse_tail:
            fused_loss = training and loss_out is not None and "live" in hp
            if fused_loss:
                flags = [1] * G if group_live is None else [1 if f else 0 for f in group_live]
                if flags != hp["live_host"]:
                    hp["live"].copy_(torch.tensor(flags, dtype=torch.int32), non_blocking=False)
                    hp["live_host"] = flags
            self.ctx.call("radnet_head_tail_fwd", hp["y5"], hp["R"], hp["hw"], 2048, self.dense_w, self.dense_ld, self.dense_b, self.nc, self.nreg,
                          hp["feat"], hp["pcls"], hp["pregr"], hp["y1"] if fused_loss else None, hp["y2"] if fused_loss else None,
                          hp["dz"] if fused_loss else None, loss_out if fused_loss else None, G, hp["live"] if fused_loss else None,
                          hp["tail_scratch"])
            hp["_tail_fused"] = fused_loss
            return
        self.ctx.call("radnet_avgpool_fwd", hp["y5"], hp["R"], hp["hw"], 2048, hp["feat"])
        self.ctx.call("radnet_dense_heads_fwd", hp["feat"], hp["R"], 2048, self.dense_w, self.dense_ld, self.dense_b, self.nc, self.nreg,
                      hp["pcls"], hp["pregr"])

    def head_backward(self, hp, accumulate=False, loss_out=None, on_part=None):
        """losses (losses.py:69-95) + gradients of every stage-5 conv and both dense heads into the head grad arena.
        on_part(lo, hi): called after each block of the backward program (last block first) with the slice [lo, hi) of
        the flat gradient arena that block has just completed -- the data-parallel trainer starts that slice's all-reduce
        while the earlier blocks are still being differentiated; the biases and the dense heads (the arena's tail from
        head_bias_off) are complete when the call returns."""
        G = hp.get("groups", 1)
        if hp.get("_tail_fused"):
            pass                              # losses + dz came out of head_forward's launch
        elif G == 1:
            self.ctx.call("radnet_det_loss", hp["pcls"], hp["pregr"], hp["y1"], hp["y2"], hp["R"], self.nc, self.nreg, hp["dz"],
                          self.det_losses if loss_out is None else loss_out)
        else:
            # every image is its own reference step (losses.py:69-95 normalise per call): loss + gradient rows per group;
            # loss_out is then a list with one slot per group, None = that image takes no classifier step (zero rows)
            rg = hp["R"] // G
            for g in range(G):
                lo = loss_out[g] if loss_out is not None else self.det_losses
                if lo is None:
                    self.ctx.call("radnet_fill_zero", hp["dz"][g * rg:], C.c_uint64(rg * (self.nc + self.nreg) * 4))
                    continue
                self.ctx.call("radnet_det_loss", hp["pcls"][g * rg:], hp["pregr"][g * rg:], hp["y1"][g * rg:], hp["y2"][g * rg:], rg, self.nc,
                              self.nreg, hp["dz"][g * rg:], lo)
        self.ctx.call("radnet_dense_heads_bwd", hp["feat"], hp["dz"], hp["R"], 2048, self.dense_w, self.dense_ld, self.nc + self.nreg,
                      self.dense_dw, self.dense_db, hp["dfeat"], 1 if accumulate else 0)
        self.ctx.call("radnet_avgpool_bwd_relu", hp["dfeat"], hp["y5"], hp["R"], hp["hw"], 2048, hp["g_last"])
        if on_part is None or "bwd_parts" not in hp:
            self._run(hp["bwd"])
            return
        for part, (lo, hi) in hp["bwd_parts"]:
            self._run(part)
            on_part(lo, hi)

    # ------------------------------------------------------------------------------------------ targets
    def upload_gt(self, gt_boxes, gt_is_bg, gt_cls):
        """GT boxes [g][4] fp64 (x1,y1,x2,y2 source px), is-bg flags and class indices -> device (once per sample)."""
        g = len(gt_boxes)
        dev = self.dev
        return dict(g=g,
                    boxes=torch.from_numpy(np.ascontiguousarray(gt_boxes, dtype=np.float64).reshape(-1, 4)).to(dev) if g else None,
                    isbg=torch.from_numpy(np.ascontiguousarray(gt_is_bg, dtype=np.int32)).to(dev) if g else None,
                    cls=torch.from_numpy(np.ascontiguousarray(gt_cls, dtype=np.int32)).to(dev) if g else None)

    def anchor_targets_launch(self, gt, width, height, rw, rh, slot=0):
        """Device half of utils.calc_region_props (utils.py:585-766) + ASYNC copy of the valid/overlap maps to pinned
        host memory.  The caller may enqueue unrelated GPU work (base forward) before anchor_targets_finish()."""
        fw, fh = self.feat_len(rw), self.feat_len(rh)
        A = self.A
        key = ("atgt", fh, fw, slot)
        if key not in self._plans:
            dev = self.dev
            self._plans[key] = dict(valid=torch.zeros(A, fh, fw, dtype=torch.uint8, device=dev), overlap=torch.zeros(A, fh, fw, dtype=torch.uint8, device=dev),
                                    regr=torch.zeros(fh, fw, 4 * A, dtype=torch.float64, device=dev),
                                    ycls=torch.zeros(fh, fw, 2 * A, dtype=torch.float32, device=dev),
                                    yregr=torch.zeros(fh, fw, 8 * A, dtype=torch.float32, device=dev),
                                    h_valid=torch.zeros(A, fh, fw, dtype=torch.uint8).pin_memory(), h_overlap=torch.zeros(A, fh, fw, dtype=torch.uint8).pin_memory(),
                                    best=torch.zeros(1024, 4, dtype=torch.int32, device=dev), nfor=torch.zeros(1024, dtype=torch.int32, device=dev),
                                    scratch=torch.zeros(1024, dtype=torch.int64, device=dev), event=torch.cuda.Event(), fw=fw, fh=fh)
        P = self._plans[key]
        g = gt["g"]
        rc = self.lib.radnet_anchor_targets(self.ctx.h, gt["boxes"].data_ptr() if g else None, gt["isbg"].data_ptr() if g else None, g, int(width), int(height),
                                            int(rw), int(rh), fw, fh, self.anchor_sizes.ctypes.data_as(C.POINTER(C.c_double)), len(self.anchor_sizes),
                                            self.anchor_ratios.ctypes.data_as(C.POINTER(C.c_double)), len(self.anchor_ratios), float(self.C.rpn_stride),
                                            float(self.C.rpn_max_overlap), P["valid"].data_ptr(), P["overlap"].data_ptr(), P["regr"].data_ptr(),
                                            P["best"].data_ptr(), P["nfor"].data_ptr(), P["scratch"].data_ptr())
        self.ctx.check(rc, "radnet_anchor_targets")
        self._copy(P["h_valid"], P["valid"])
        self._copy(P["h_overlap"], P["overlap"])
        P["event"].record()
        P["g"] = g
        return P

    def anchor_targets_finish(self, P):
        """Host half (utils.py:777-813: the RNG-driven subsampling stays on NumPy's global stream by design), then the
        device packs the fp32 NHWC training tensors.  Raises KeyError exactly where the reference does."""
        P["event"].synchronize()
        valid = P["h_valid"].numpy()
        n_pos = subsample_valid(valid, P["h_overlap"].numpy())
        self._copy(P["valid"], P["h_valid"])
        self.ctx.call("radnet_anchor_targets_pack", P["valid"], P["overlap"], P["regr"], P["fw"], P["fh"], self.A, C.c_double(float(self.C.std_scaling)),
                      P["ycls"], P["yregr"])
        return P["ycls"], P["yregr"], n_pos

    def anchor_targets(self, gt_boxes, gt_is_bg, width, height, rw, rh, slot=0):
        """utils.calc_region_props end to end: (y_cls, y_regr) fp32 NHWC device tensors, best_anchor (host), n_pos."""
        gt = self.upload_gt(gt_boxes, gt_is_bg, np.zeros(len(gt_boxes), np.int32))
        P = self.anchor_targets_launch(gt, width, height, rw, rh, slot)
        ycls, yregr, n_pos = self.anchor_targets_finish(P)
        g = gt["g"]
        best_h = P["best"][:g].cpu().numpy().astype(np.int64) if g else np.zeros((0, 4), np.int64)
        return ycls, yregr, best_h, n_pos

    def roi_targets(self, R_dev, n_dev, gt, width, height, rw, rh, n_max=300):
        """rpn.calc_iou on device for the first min(*n_dev, n_max) proposals; one pinned D2H brings back the per-RoI
        class code (-1 = dropped).  Returns (plan, cls host int32 [n], n)."""
        return self.roi_targets_finish(self.roi_targets_launch(R_dev, n_dev, gt, width, height, rw, rh, n_max))

    def roi_targets_launch(self, R_dev, n_dev, gt, width, height, rw, rh, n_max=300, slot=0):
        """Device half of roi_targets + asynchronous copy of the class codes to pinned memory (`slot`: buffer set)."""
        dev = self.dev
        key = ("rtgt", slot)
        if key not in self._plans:
            self._plans[key] = dict(keep=torch.zeros(1024, dtype=torch.uint8, device=dev), cls=torch.zeros(1024, dtype=torch.int32, device=dev),
                                    box=torch.zeros(1024, 4, dtype=torch.int32, device=dev), t=torch.zeros(1024, 4, dtype=torch.float64, device=dev),
                                    iou=torch.zeros(1024, dtype=torch.float64, device=dev), h_cls=torch.zeros(1024, dtype=torch.int32).pin_memory(),
                                    h_n=torch.zeros(1, dtype=torch.int32).pin_memory(), sel=torch.zeros(1024, dtype=torch.int32, device=dev),
                                    h_sel=torch.zeros(1024, dtype=torch.int32).pin_memory())
        P = self._plans[key]
        rc = self.lib.radnet_roi_targets(self.ctx.h, R_dev.data_ptr(), int(n_max), gt["boxes"].data_ptr() if gt["g"] else None,
                                         gt["cls"].data_ptr() if gt["g"] else None, gt["g"], int(width), int(height), int(rw), int(rh),
                                         float(self.C.rpn_stride), float(self.C.classifier_min_overlap), float(self.C.classifier_max_overlap),
                                         self.regr_std.ctypes.data_as(C.POINTER(C.c_double)), int(self.bg), P["keep"].data_ptr(), P["cls"].data_ptr(),
                                         P["box"].data_ptr(), P["t"].data_ptr(), P["iou"].data_ptr(), n_dev.data_ptr())
        self.ctx.check(rc, "radnet_roi_targets")
        self._copy(P["h_cls"][:n_max], P["cls"][:n_max])
        self._copy(P["h_n"], n_dev)
        if "event" not in P:
            P["event"] = torch.cuda.Event()
        P["event"].record()
        return P

    def roi_targets_finish(self, P):
        """Waits for the class codes only (an event after their copy), not for whatever was enqueued behind them."""
        P["event"].synchronize()
        if self._chain_plans:
            self.check_chains()            # the codes depend on the base forward: a chain that gave up has written its error word by now
        n = int(P["h_n"][0])
        return P, P["h_cls"].numpy()[:max(n, 0)], n

    def pack_roi_batch(self, P, sel, hp, group=0):
        """Selected RoIs + targets of one image into the head plan (rows of `group` in a per-GPU mini-batch plan)."""
        k = len(sel)
        o = group * (hp["R"] // hp.get("groups", 1))
        P["h_sel"][:k] = torch.from_numpy(np.ascontiguousarray(sel, dtype=np.int32))
        # the pack kernel reads the selection straight from the pinned (device-mapped) host buffer: no copy launch in front of it on the
        # classifier lane (the buffer belongs to this buffer set: the host writes it again NBUF steps later at the earliest)
        if self.KERNEL_COPIES and os.environ.get("RADNET_NO_PINNED_SEL", "0") != "1":
            self.ctx.call("radnet_roi_batch_pack", P["h_sel"], k, P["cls"], P["box"], P["t"], self.nc, int(self.bg), hp["rois"][o:], hp["y1"][o:], hp["y2"][o:])
            return
        self._copy(P["sel"][:k], P["h_sel"][:k])
        self.ctx.call("radnet_roi_batch_pack", P["sel"], k, P["cls"], P["box"], P["t"], self.nc, int(self.bg), hp["rois"][o:], hp["y1"][o:], hp["y2"][o:])

    def idle_roi_group(self, hp, group):
        """An image of the mini-batch that takes no classifier step (no RoI kept / dropped by the labeller): its rows still
        pass through the stage-5 GEMMs, on a harmless 1x1 RoI; head_backward zeroes their gradient."""
        rg = hp["R"] // hp["groups"]
        hp["rois"][group * rg:(group + 1) * rg].copy_(torch.tensor([0.0, 0.0, 1.0, 1.0], device=self.dev).expand(rg, 4))


def _uniform_table_probs(chan_of_item, neg_chans, n_total):
    """The reference builds p[i] = (count[ch_i] / n) / count[ch_i] from the NEGATIVES' channel histogram
    (utils.py:789-795, 804-810) and raises KeyError for a channel without negatives.  Vectorised: the same IEEE
    operations element-wise, hence the same p and the same draws."""
    counts = np.bincount(neg_chans, minlength=int(chan_of_item.max()) + 1 if len(chan_of_item) else 1)
    if (counts == 0).any():
        bad = counts[chan_of_item] == 0
        if bad.any():
            raise KeyError(int(chan_of_item[np.argmax(bad)]))
    per_chan = np.zeros(counts.shape[0], dtype=np.float64)       # p depends on the item's channel only: same two divisions
    nz = counts > 0
    per_chan[nz] = (counts[nz] / n_total) / counts[nz]
    return per_chan[chan_of_item]


def choice_without_replacement(n, size, p):
    """== np.random.choice(n, size, replace=False, p=p) on NumPy's GLOBAL legacy RandomState: same result, same
    consumption of the MT19937 stream (the uniforms are drawn here with np.random.random_sample, round by round,
    exactly as RandomState.choice does), with each round's cumsum / searchsorted / de-duplication done by the C
    helper radnet_host_choice_round instead of ~10 NumPy passes (4 ms -> 0.3 ms for 20 000 anchors)."""
    lib = L.load_library()
    p = np.array(p, dtype=np.float64, copy=True)
    if p.ndim != 1 or p.shape[0] != n:
        raise ValueError("'a' and 'p' must have same size")
    if size > n:
        raise ValueError("Cannot take a larger sample than population when 'replace=False'")
    if np.count_nonzero(p > 0) < size:
        raise ValueError("Fewer non-zero entries in p than size")
    found = np.zeros(size + 1, dtype=np.int64)               # + 1: the helper's branch-free append writes one slot ahead
    live_idx = np.flatnonzero(p > 0).astype(np.int64)        # zero entries only ever add 0.0 to NumPy's cumsum
    live_p = np.ascontiguousarray(p[live_idx])
    n_live = np.array([live_idx.shape[0]], dtype=np.int64)
    cdf = np.empty(n, dtype=np.float64)
    sel = np.empty(n, dtype=np.uint8)
    n_uniq = 0
    # raw addresses once: ndarray.ctypes builds a helper object per access (~3 us each, x7 per round)
    a_p, a_idx, a_n, a_found, a_cdf, a_sel = (v.__array_interface__["data"][0] for v in (live_p, live_idx, n_live, found, cdf, sel))
    fn = lib.radnet_host_choice_round
    while n_uniq < size:
        x = np.random.random_sample(size - n_uniq)
        n_uniq += int(fn(a_p, a_idx, a_n, a_found, n_uniq, x.__array_interface__["data"][0], x.shape[0], a_cdf, a_sel))
    return found[:size]


def subsample_valid(valid, overlap, max_regions=256):
    """Host half of utils.calc_region_props (utils.py:777-813): random disabling of surplus positives /
    negatives on the *global NumPy RNG stream*, which is part of the reference's contract (train.py:41,134).
    valid / overlap: uint8 [A][fh][fw] (the NCHW order np.where enumerates in the reference); `valid` is
    edited in place.  Returns n_pos."""
    if not (valid.flags.c_contiguous and overlap.flags.c_contiguous):
        raise ValueError("subsample_valid: label maps must be C-contiguous")
    vf, of = valid.reshape(-1), overlap.reshape(-1)      # flat C order == the order np.where enumerates (a, y, x) in
    hw = valid.shape[1] * valid.shape[2]
    live = vf == 1
    pos = np.flatnonzero(live & (of == 1))
    neg = np.flatnonzero(live & (of == 0))
    n_pos, n_neg = len(pos), len(neg)
    neg_ch = neg // hw
    half = int(max_regions / 2)
    if n_pos > max_regions / 2:
        p = _uniform_table_probs(pos // hw, neg_ch, n_pos)
        off = choice_without_replacement(n_pos, n_pos - half, p)
        vf[pos[off]] = 0
        n_pos = half
    if n_neg + n_pos > max_regions:
        p = _uniform_table_probs(neg_ch, neg_ch, n_neg)
        off = choice_without_replacement(n_neg, n_neg - n_pos, p)
        vf[neg[off]] = 0
    return n_pos


def select_samples(cls_kept, bg, n_rois):
    """train.get_selected_samples (train.py:93-129) on the kept RoIs' class indices; host NumPy RNG."""
    cls_kept = np.asarray(cls_kept)
    neg = np.where(cls_kept == bg)[0]
    pos = np.where(cls_kept != bg)[0]
    if len(pos) < n_rois // 2:
        sel_pos = pos.tolist()
    else:
        sel_pos = np.random.choice(pos, n_rois // 2, replace=False).tolist()
    if len(neg) > 0:
        need = n_rois - len(sel_pos)
        try:
            sel_neg = np.random.choice(neg, need, replace=False).tolist()
        except Exception:
            sel_neg = np.random.choice(neg, need, replace=True).tolist()
        return sel_pos + sel_neg, len(pos)
    sel_pos = np.random.choice(pos, len(pos), replace=False).tolist()
    sel_pos += np.random.choice(pos, n_rois - len(sel_pos), replace=True).tolist()
    return sel_pos, len(pos)
