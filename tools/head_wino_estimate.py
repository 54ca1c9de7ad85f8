#!/usr/bin/env python3
"""What would the classifier's 3x3 convs (res5x_branch2b on 20 RoIs of 7x7: M = 980, 512 -> 512) cost as Winograd in the TRAIN step?
Times the pieces alone on the chip (autotuned launch shapes): batched forward GEMMs and batched weight-gradient GEMMs for F(2x2)
(16 positions, 320 tiles) and F(4x4) (36 positions, 80 tiles), the transforms, against the direct forward / dgrad / wgrad."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rock-art-radnet_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from radnet_hip import lib as L  # noqa: E402


def timeit(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    ctx = L.Context(0)
    ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    ctx.check(ctx.lib.radnet_set_workspace(ctx.h, ws.data_ptr(), ws.numel()), "ws")
    ctx.check(ctx.lib.radnet_set_autotune(ctx.h, 1), "autotune")
    R, c, n = 20, 512, 512
    x = torch.randn(R, 7, 7, c, device="cuda")
    for form, P, T in ((2, 16, R * 16), (4, 36, R * 4)):
        V = torch.randn(P, T, c, device="cuda")
        U = torch.randn(P, c, n, device="cuda")
        M = torch.empty(P, T, n, device="cuda")
        dZ = torch.randn(P, T, n, device="cuda")
        dU = torch.empty(P, c, n, device="cuda")
        y = torch.empty(R, 7, 7, n, device="cuda")
        sfx = "4" if form == 4 else ""
        t_in = timeit(lambda: ctx.call("radnet_winograd%s_input" % sfx, x, R, 7, 7, c, V))
        t_g = timeit(lambda: ctx.call("radnet_gemm_batched", V, U, M, P, T, n, c))
        t_out = timeit(lambda: ctx.call("radnet_winograd%s_output" % sfx, M, R, 7, 7, n, None, None, 1, y, n))
        t_dy = timeit(lambda: ctx.call("radnet_winograd%s_dy" % sfx, y, R, 7, 7, n, n, None, dZ))
        t_wg = timeit(lambda: ctx.call("radnet_wgrad_batched", V, dZ, dU, P, T, c, n, 0))
        w = torch.randn(9 * c, n, device="cuda")
        dw = torch.empty(9 * c, n, device="cuda")
        t_f = timeit(lambda: ctx.call("radnet_winograd%s_filter" % sfx, w, c, n, n, U))
        t_fg = timeit(lambda: ctx.call("radnet_winograd%s_filter_grad" % sfx, dU, c, n, n, dw, 0))
        print("F(%dx%d): tiles %d  input %.1f  GEMMs %.1f  output %.1f  | dY transform %.1f  batched wgrad %.1f  filter-grad %.1f | filter transform %.1f us"
              % (form, form, T, t_in, t_g, t_out, t_dy, t_wg, t_fg, t_f))
        print("   forward %.1f   dgrad (same three kernels, other filter) %.1f   wgrad %.1f   + 2 filter transforms per step %.1f   -> per layer %.1f us"
              % (t_in + t_g + t_out, t_in + t_g + t_out, t_dy + t_wg + t_fg, 2 * t_f, 2 * (t_in + t_g + t_out) + t_dy + t_wg + t_fg + 2 * t_f))
    # direct forms
    from test_gpu_kernels import conv_desc
    wt = torch.randn(9 * c, n, device="cuda")
    yd = torch.empty(R, 7, 7, n, device="cuda")
    d = conv_desc(L, x, wt, yd, R, 7, 7, c, 7, 7, 3, 1, 1, n, n)
    t_fwd = timeit(lambda: ctx.check(ctx.lib.radnet_conv_fwd(ctx.h, C.byref(d)), "fwd"))
    dy = torch.randn(R, 7, 7, n, device="cuda")
    dx = torch.empty(R, 7, 7, c, device="cuda")
    dwd = torch.zeros(9 * c, n, device="cuda")
    d.dy, d.ld_dy, d.dx, d.ld_dx, d.dw, d.dw_accumulate = dy.data_ptr(), n, dx.data_ptr(), c, dwd.data_ptr(), 2
    t_dg = timeit(lambda: ctx.check(ctx.lib.radnet_conv_dgrad(ctx.h, C.byref(d)), "dgrad"))
    t_wgd = timeit(lambda: ctx.check(ctx.lib.radnet_conv_wgrad(ctx.h, C.byref(d)), "wgrad"))
    print("direct: forward %.1f  dgrad %.1f  wgrad %.1f  -> per layer %.1f us" % (t_fwd, t_dg, t_wgd, t_fwd + t_dg + t_wgd))


if __name__ == "__main__":
    main()
