"""cont_train.py trainability on the MI355X engine (SURVEY.md 8d cfg 2, secondary mode): ResNet50 stages 3 and 4 train
in BOTH models (cont_train.py:120-131), FixedBatchNormalization stays frozen, each model's Adam keeps its own moments
for those shared weights (cont_train.py:169-185).

What this adds to FasterRCNNEngine (train.py mode: whole base frozen):
  * stage-3/4 conv weights + biases live in a third flat arena (grad twin, TWO sets of Adam moments);
  * the base program is split into the frozen stem (conv1 .. stage 2) and the trainable part (stages 3-4), the latter
    with a backward program: dgrad / wgrad per conv exactly as the stage-5 head blocks, plus the input gradient of the
    stride-2 1x1 convs of res4a (GEMM on the compact grid + radnet_scatter_strided);
  * the RPN backward continues through rpn_conv1 into dL/dF; the head backward continues through res5a into the pooled
    RoIs and RoiPoolingConv's crop-resize (radnet_roi_resize_bwd) into dL/dF.
Nothing flows below stage 3 (conv1 / stage 2 are frozen in every mode, resnet50.py:178-195).
"""
import ctypes as C
import os

import numpy as np
import torch

from . import lib as L
from .engine import RES_STAGES, Arena, FasterRCNNEngine


class ContEngine(FasterRCNNEngine):
    HEAD_TRAIN_WINOGRAD = False        # cont_train.py mode: gradients flow through these layers into two optimizers; direct form kept
    WORKLOAD = "cont"            # one lane, other launch mix: the pipelined step's in-situ tables do not apply (measured: -1 %)
    supports_batched = False     # cont_train.py's step runs one image at a time (both optimizers move the shared stages)
    # rpn_conv1 keeps F(2x2,3x3) here: its weight gradient is formed in the Winograd domain and flows on into stages 3 / 4, whose first Adam
    # steps divide by |g| + 1e-7 -- F(4x4)'s 15x larger fp32 rounding would show in the updates of small-gradient weights.  The ten
    # stage-3/4 3x3 layers run their FORWARD as F(4x4) (round 4; their gradients stay direct): both optimizers' passes over the shared arena
    # rewrite the transformed filters (radnet_adam_step_fused), RADNET_CONT_DIRECT_3X3=1 brings the direct forward back.
    WINOGRAD_F4_LAYERS = tuple("res%d%s_branch2b" % (st, bl) for st, bls in ((3, "abcd"), (4, "abcdef")) for bl in bls)
    S34_WINOGRAD = os.environ.get("RADNET_CONT_DIRECT_3X3", "0") != "1"
    # their weight gradients in the Winograd domain too (rpn_conv1's path): measured SLOWER here -- 134.2 against 138.9 images/s: three
    # launches + a column sum + the unpaired data gradient against one paired launch per layer -- off unless RADNET_CONT_WINO_WGRAD=1
    S34_WINO_WGRAD = os.environ.get("RADNET_CONT_WINO_WGRAD", "0") == "1"

    def __init__(self, C_cfg, device_index=0, n_classes=None, bce_mode=0, lr=2e-5, autotune=True):
        super().__init__(C_cfg, device_index, n_classes, bce_mode, lr, autotune)

    # ------------------------------------------------------------------------------------------ layers
    def _build_layers(self):
        super()._build_layers()
        dev = self.dev
        names = [n for n in self.convs if n.startswith(("res3", "res4"))]
        self.s34_names = names
        ar = Arena(dev)
        for n in names:
            c = self.convs[n]
            ar.add(n + "/kernel", (c.kh * c.kh * c.cin, c.cout))
        self.s34_bias_off = ar.n
        for n in names:
            ar.add(n + "/bias", (self.convs[n].cout,))
        self.s34_bias_len = ar.n - self.s34_bias_off
        ar.finalize()
        ar.m2, ar.v2, ar.t2 = torch.zeros_like(ar.p), torch.zeros_like(ar.p), 0      # the classifier model's Adam state
        self.s34_arena = ar
        self.s34_scale = torch.ones(self.s34_bias_len, dtype=torch.float32, device=dev)
        self.s34_t0 = torch.zeros(self.s34_bias_len, dtype=torch.float32, device=dev)
        self.s34_shift = torch.zeros(self.s34_bias_len, dtype=torch.float32, device=dev)
        for n in names:
            c = self.convs[n]
            c.weight, c.bias = ar.param(n + "/kernel"), ar.param(n + "/bias")
            c.dweight, c.dbias = ar.grad(n + "/kernel"), ar.grad(n + "/bias")
            o = ar.offsets[n + "/bias"][0] - self.s34_bias_off
            c.scale = self.s34_scale[o:o + c.cout]
            c.t0 = self.s34_t0[o:o + c.cout]
            c.shift = self.s34_shift[o:o + c.cout]

    def refresh_s34_shift(self):
        """shift = scale * bias + t0 for every stage-3/4 conv (frozen BN folded around the trainable bias)."""
        bias = self.s34_arena.p[self.s34_bias_off:self.s34_bias_off + self.s34_bias_len]
        self.ctx.call("radnet_affine_vec", self.s34_shift, self.s34_scale, bias, self.s34_t0, C.c_int64(self.s34_bias_len))

    def set_weights(self, W):
        super().set_weights(W)            # trainable convs (bias is not None) get bias / t0 / scale separately
        self.refresh_s34_shift()
        torch.cuda.synchronize(self.dev)

    def get_weights(self, names=None):
        out = super().get_weights(names)
        for n in self.s34_names:
            c = self.convs[n]
            out[n] = {"kernel": c.weight.detach().cpu().numpy().reshape(c.kh, c.kh, c.cin, c.cout).copy(),
                      "bias": c.bias.detach().cpu().numpy()[:c.cout].copy()}
        return out

    def adam_s34(self, which, grad_scale=1.0):
        """Adam over the shared stage-3/4 arena with the moments of optimizer `which` (0: RPN model, 1: classifier
        model); the gradient arena is cleared in the same pass."""
        ar = self.s34_arena
        if which == 0:
            ar.t += 1
            m, v, t = ar.m, ar.v, ar.t
        else:
            ar.t2 += 1
            m, v, t = ar.m2, ar.v2, ar.t2
        wino = self._s34_adam_wino()
        if wino is not None:          # Adam + folded shifts + the Winograd filters of the ten 3x3 kernels, one launch
            self.ctx.check(self.lib.radnet_adam_step_fused(
                self.ctx.h, ar.p.data_ptr(), ar.g.data_ptr(), m.data_ptr(), v.data_ptr(), C.c_int64(ar.n), t, C.c_float(self.lr), C.c_float(0.9),
                C.c_float(0.999), C.c_float(1e-7), C.c_float(grad_scale), 1, C.c_int64(self.s34_bias_off), C.c_int64(self.s34_bias_len),
                self.s34_scale.data_ptr(), self.s34_t0.data_ptr(), self.s34_shift.data_ptr(), wino[0], wino[1]), "radnet_adam_step_fused")
            return
        self.ctx.call("radnet_adam_step", ar.p, ar.g, m, v, C.c_int64(ar.n), t, C.c_float(self.lr), C.c_float(0.9), C.c_float(0.999),
                      C.c_float(1e-7), C.c_float(grad_scale), 1)
        self.refresh_s34_shift()
        if self.S34_WINOGRAD and self.use_winograd:
            self._refresh_winograd([n for n in self.WINOGRAD_F4_LAYERS if n in self.convs])

    def _s34_adam_wino(self):
        """(radnet_adam_wino[], n) of the stage-3/4 3x3 kernels whose forward runs on Winograd F(4x4) filters, or None."""
        if getattr(self, "_s34_wino", None) is None:
            ent = []
            if self.S34_WINOGRAD and self.use_winograd and self.s34_bias_off % 4 == 0 and self.s34_bias_len % 4 == 0:
                for name in self.WINOGRAD_F4_LAYERS:
                    c = self.convs.get(name)
                    if c is None or not self._uses_winograd(c):
                        continue
                    if c.wino_u is None:
                        self._refresh_winograd([name])
                    off, size = self.s34_arena.offsets[name + "/kernel"]
                    if c.wino_m != 4 or c.ldw != c.cout or off % 4 or size != 9 * c.cin * c.cout or (c.cin * c.cout // 4) % 64:
                        ent = []
                        break
                    ent.append((off, c.cin, c.cout, c.wino_u.data_ptr()))
            if ent and len(ent) <= 12:
                arr = (L.AdamWino * len(ent))()
                for k, (off, ci, co, u) in enumerate(ent):
                    arr[k].off, arr[k].c, arr[k].n, arr[k].u = off, ci, co, u
                self._s34_wino = (arr, len(ent))
            else:
                self._s34_wino = False
        return self._s34_wino or None

    # ------------------------------------------------------------------------------------------ base program
    def _plan_base(self, nb, H, W, slot=0):
        key = ("base", nb, H, W, slot)
        if key in self._plans:
            return self._plans[key]
        dev = self.dev
        keep = []

        def buf(*shape):
            b = torch.empty(shape, dtype=torch.float32, device=dev)
            keep.append(b)
            return b

        stem, s34, blocks = [], [], []
        x = buf(nb, H, W, 4)
        oh, ow = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        y = buf(nb, oh, ow, 64)
        d, _, _ = self._desc(self.convs["conv1"], x, nb, H, W, y)
        stem.append(("conv", d))
        ph, pw = (oh - 3) // 2 + 1, (ow - 3) // 2 + 1
        p = buf(nb, ph, pw, 64)
        stem.append(("maxpool", (y, p, nb, oh, ow, 64, 3, 2)))
        cur, h, w = p, ph, pw
        for st, bls, (f1, f2, f3), stride in RES_STAGES:
            ops = s34 if st >= 3 else stem
            for bl in bls:
                b = "res%d%s_branch" % (st, bl)
                first = bl == "a"
                ca, cb, cc = self.convs[b + "2a"], self.convs[b + "2b"], self.convs[b + "2c"]
                oh, ow = ((h - 1) // ca.stride + 1, (w - 1) // ca.stride + 1)
                a = buf(nb, oh, ow, f1); da, _, _ = self._desc(ca, cur, nb, h, w, a); ops.append(("conv", da))
                ds = None
                if first:
                    sc = buf(nb, oh, ow, f3); ds, _, _ = self._desc(self.convs[b + "1"], cur, nb, h, w, sc, relu=False)
                    if st < 3:                     # frozen stem: the shortcut first, so that 3x3 + expand (+ next reduce) are neighbours (_fuse_bottlenecks)
                        ops.append(("conv", ds))
                else:
                    sc = cur
                bb = buf(nb, oh, ow, f2)
                wino_v = None
                if st >= 3 and self.S34_WINOGRAD:      # trainable 3x3: Winograd forward on filters the optimizers keep transformed
                    n_keep = len(keep)
                    op_b, db = self._fwd_op(cb, a, nb, oh, ow, bb, keep)
                    ops.append(op_b)
                    if op_b[0] == "wino" and self.wino_wgrad and self.S34_WINO_WGRAD:
                        wino_v = keep[n_keep]          # the transformed input: the weight gradient is formed on it (rpn_conv1's path)
                else:
                    db, _, _ = self._desc(cb, a, nb, oh, ow, bb); ops.append(("conv", db))
                if first and st >= 3:
                    ops.append(("conv", ds))
                out = buf(nb, oh, ow, f3); dc, _, _ = self._desc(cc, bb, nb, oh, ow, out, relu=True, addend=sc); ops.append(("conv", dc))
                if st >= 3:
                    blocks.append(dict(st=st, first=first, x=cur, a=a, b=bb, out=out, da=da, db=db, dc=dc, ds=ds, h=h, w=w, oh=oh, ow=ow, wino_v=wino_v,
                                       cin=ca.cin, f=(f1, f2, f3), stride=ca.stride, names=(b + "2a", b + "2b", b + "2c", b + "1")))
                cur, h, w = out, oh, ow
        F = cur
        if self.bneck_fuse and not self.use_chain:
            stem[:] = self._fuse_bottlenecks(stem)
        dF = buf(nb * h * w, 1024)                # dL/dF, the producer's ReLU mask (F > 0) already applied
        bwd = self._blocks_backward(blocks, dF, nb, buf, lowest_stage=3)
        plan = dict(ops=stem + s34, ops_stem=stem, ops_s34=s34, bwd34=bwd, x=x, F=F, dF=dF, fh=h, fw=w, keep=keep, blocks=blocks)
        self._plans[key] = plan
        self._base_of_F = getattr(self, "_base_of_F", {})
        self._base_of_F[F.data_ptr()] = plan
        return plan

    def _bdesc(self, fdesc, conv, dy, ld_dy, dx=None, ld_dx=0, dx_add=None, dx_mask=None):
        d = L.ConvDesc.from_buffer_copy(fdesc)
        d.dy, d.ld_dy, d.gscale = dy.data_ptr(), ld_dy, conv.scale.data_ptr()
        d.dw, d.dw_accumulate, d.db = conv.dweight.data_ptr(), 1, conv.dbias.data_ptr()
        if dx is not None:
            d.dx, d.ld_dx = dx.data_ptr(), ld_dx
            d.dx_add = dx_add.data_ptr() if dx_add is not None else None
            d.ld_dx_add = ld_dx
            d.dx_mask = dx_mask.data_ptr() if dx_mask is not None else None
            d.ld_dx_mask = ld_dx
        return d

    @staticmethod
    def _compact(desc):
        """The same 1x1 convolution seen on its OUTPUT grid (stride 1): radnet_conv_dgrad then yields the compact input
        gradient, one row per sampled pixel (radnet_scatter_strided spreads it over the full grid)."""
        d = L.ConvDesc.from_buffer_copy(desc)
        d.h, d.w_, d.stride = desc.oh, desc.ow, 1
        return d

    def _blocks_backward(self, blocks, g_top, nb, buf, lowest_stage):
        """Backward program of a chain of conv_block / identity_block (resnet50.py:41-117) given the gradient w.r.t. the
        last block's output (ReLU mask applied).  Gradients of the blocks' weights accumulate into their arenas; the
        input gradient stops at the first block of `lowest_stage` (its input comes from frozen layers)."""
        bwd = []
        g_out = g_top
        for bi in range(len(blocks) - 1, -1, -1):
            B = blocks[bi]
            f1, f2, f3 = B["f"]
            M = nb * B["oh"] * B["ow"]
            ca, cb, cc = (self.convs[n] for n in B["names"][:3])
            g_b, g_a = buf(M, f2), buf(M, f1)
            dC = self._bdesc(B["dc"], cc, g_out, f3, g_b, f2, None, B["b"])
            bwd += [("wgrad", dC), ("dgrad", dC)]
            dB = self._bdesc(B["db"], cb, g_b, f2, g_a, f1, None, B["a"])
            if B.get("wino_v") is not None:        # weight gradient in the Winograd domain (4x fewer flops), bias gradient as a column sum
                self._wg_scratch = getattr(self, "_wg_scratch", {})
                bwd += [self._wino_wgrad_op(cb, B["wino_v"], g_b, f2, nb, B["oh"], B["ow"], self._wg_scratch),
                        ("colsum", [g_b.data_ptr(), M, f2, f2, cb.scale.data_ptr(), cb.dbias.data_ptr(), 1]), ("dgrad", dB)]
            else:
                bwd += [("wgrad", dB), ("dgrad", dB)]
            if B["first"]:
                cs = self.convs[B["names"][3]]
                dA = self._bdesc(B["da"], ca, g_a, f1)
                dS = self._bdesc(B["ds"], cs, g_out, f3)
                bwd += [("wgrad", dA), ("wgrad", dS)]
                if B["st"] > lowest_stage:
                    # input gradient through the two stride-s 1x1 convs: both GEMMs on the compact grid, summed by the
                    # second one's epilogue, then spread over the full grid under the previous block's ReLU mask
                    cin = B["cin"]
                    ga_c, gs_c = buf(M, cin), buf(M, cin)
                    dAc = self._bdesc(self._compact(B["da"]), ca, g_a, f1, ga_c, cin)
                    dSc = self._bdesc(self._compact(B["ds"]), cs, g_out, f3, gs_c, cin, ga_c)
                    g_prev = buf(nb * B["h"] * B["w"], cin)
                    bwd += [("dgrad", dAc), ("dgrad", dSc),
                            ("scatter", (gs_c.data_ptr(), nb, B["oh"], B["ow"], cin, B["stride"], B["h"], B["w"], B["x"].data_ptr(), g_prev.data_ptr()))]
                    g_out = g_prev
            else:
                g_prev = buf(M, f3)
                dA = self._bdesc(B["da"], ca, g_a, f1, g_prev, f3, g_out, B["x"])
                bwd += [("wgrad", dA), ("dgrad", dA)]
                g_out = g_prev
        return bwd

    def stem_forward(self, plan):
        self._run(plan["ops_stem"])

    def s34_forward(self, plan):
        self._run(plan["ops_s34"])
        return plan["F"]

    def s34_backward(self, plan):
        """Stages 3-4 backward from plan['dF']; weight gradients accumulate into the stage-3/4 arena."""
        self._run(plan["bwd34"])

    # ------------------------------------------------------------------------------------------ RPN: continue into dF
    def _plan_rpn(self, fh, fw, F, nb=1):
        if nb != 1:
            raise L.RadnetError("ContEngine: the cont_train.py step runs one image at a time")
        rp = super()._plan_rpn(fh, fw, F)
        if "cont" not in rp:
            base = self._base_of_F[F.data_ptr()]
            b1 = rp["b1"]                                                   # rpn_conv1's backward descriptor
            b1.dx, b1.ld_dx, b1.dx_add, b1.dx_mask, b1.ld_dx_mask = base["dF"].data_ptr(), 1024, None, F.data_ptr(), 1024
            rp["bwd"].append(("dgrad", b1))
            rp["cont"] = True
        return rp

    # ------------------------------------------------------------------------------------------ head: continue into dF
    def _plan_head(self, R, fh, fw, F, training=True):
        hp = super()._plan_head(R, fh, fw, F, training)
        if not training:
            return hp
        if "pool_bwd" not in hp:
            base = self._base_of_F[F.data_ptr()]
            B = hp["blocks"][0]                                          # res5a: stride-2 2a and shortcut read the pooled RoIs
            ca, cs = self.convs[B["names"][0]], self.convs[B["names"][3]]
            M = hp["M"]

            def buf(*shape):
                b = torch.empty(shape, dtype=torch.float32, device=self.dev)
                hp["keep"].append(b)
                return b

            ga_c, gs_c = buf(M, 1024), buf(M, 1024)
            dAc = self._bdesc(self._compact(B["da"]), ca, B["g_a"], ca.cout, ga_c, 1024)
            dSc = self._bdesc(self._compact(B["ds"]), cs, B["g_out"], cs.cout, gs_c, 1024, ga_c)
            dpooled = buf(R, 14, 14, 1024)
            dF = base["dF"]
            hp["dpooled"] = dpooled
            hp["pool_bwd"] = [("dgrad", dAc), ("dgrad", dSc),
                              ("scatter", (gs_c.data_ptr(), R, 7, 7, 1024, 2, 14, 14, None, dpooled.data_ptr())),
                              ("fill0", (dF.data_ptr(), dF.numel() * 4)),
                              ("roi_bwd", (dpooled.data_ptr(), fh, fw, 1024, hp["rois"].data_ptr(), R, 14, dF.data_ptr())),
                              ("relu_mask", (dF.data_ptr(), F.data_ptr(), dF.numel()))]
        return hp

    def head_backward(self, hp, accumulate=False, loss_out=None):
        """Head losses and gradients, then on through RoiPoolingConv into dL/dF (the base plan's dF buffer)."""
        super().head_backward(hp, accumulate, loss_out)
        self._run(hp["pool_bwd"])
