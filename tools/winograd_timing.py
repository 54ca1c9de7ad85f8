"""Winograd F(2x2,3x3) vs the direct implicit GEMM for the 3x3 layers of the 1000x600 step (per-kernel HIP-event times)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rock-art-radnet_amd")]

from radnet_hip import lib as L  # noqa: E402

CASES = [("rpn_conv1", 1, 38, 63, 1024, 512), ("res5x_2b (20 RoIs)", 20, 7, 7, 512, 512), ("res4x_2b", 1, 38, 63, 256, 256),
         ("res3x_2b", 1, 75, 125, 128, 128), ("res2x_2b", 1, 150, 250, 64, 64)]


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    ctx = L.Context(0)
    lib = ctx.lib
    ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    ctx.check(lib.radnet_set_workspace(ctx.h, ws.data_ptr(), ws.numel()), "ws")
    ctx.check(lib.radnet_set_autotune(ctx.h, 1), "tune")
    for name, nb, h, w, cin, cout in CASES:
        x = torch.randn(nb, h, w, cin, device="cuda").relu_()
        wt = torch.randn(9 * cin, cout, device="cuda") / np.sqrt(9 * cin)
        sc, sh = torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda")
        y = torch.empty(nb, h, w, cout, device="cuda")
        d = L.ConvDesc()
        d.x, d.w, d.y, d.scale, d.shift = x.data_ptr(), wt.data_ptr(), y.data_ptr(), sc.data_ptr(), sh.data_ptr()
        d.nb, d.h, d.w_, d.c, d.oh, d.ow = nb, h, w, cin, h, w
        d.kh, d.kw, d.stride, d.pad_t, d.pad_l, d.n = 3, 3, 1, 1, 1, cout
        d.ldw, d.ldy, d.ld_add, d.act, d.act_cols = cout, cout, cout, 1, 0
        t_dir = timeit(lambda: lib.radnet_conv_fwd(ctx.h, C.byref(d)))
        fl = 2.0 * nb * h * w * cout * 9 * cin
        line = "%-20s direct %6.1f us (%5.1f TF/s)" % (name, t_dir, fl / t_dir / 1e6)
        for form in (2, 4):
            fn = "radnet_winograd4_" if form == 4 else "radnet_winograd_"
            P, T = (form + 2) ** 2, nb * ((h + form - 1) // form) * ((w + form - 1) // form)
            U = torch.empty(P, cin, cout, device="cuda")
            V = torch.empty(P, T, cin, device="cuda")
            M = torch.empty(P, T, cout, device="cuda")
            dU = torch.empty(P, cin, cout, device="cuda")
            dw = torch.empty(9 * cin, cout, device="cuda")
            t_f = timeit(lambda: getattr(lib, fn + "filter")(ctx.h, wt.data_ptr(), cin, cout, cout, U.data_ptr()))
            t_i = timeit(lambda: getattr(lib, fn + "input")(ctx.h, x.data_ptr(), nb, h, w, cin, V.data_ptr()))
            t_g = timeit(lambda: lib.radnet_gemm_batched(ctx.h, V.data_ptr(), U.data_ptr(), M.data_ptr(), P, T, cout, cin))
            t_o = timeit(lambda: getattr(lib, fn + "output")(ctx.h, M.data_ptr(), nb, h, w, cout, sc.data_ptr(), sh.data_ptr(), 1, y.data_ptr(), cout))
            t_dz = timeit(lambda: getattr(lib, fn + "dy")(ctx.h, y.data_ptr(), nb, h, w, cout, cout, None, M.data_ptr()))
            t_wg = timeit(lambda: lib.radnet_wgrad_batched(ctx.h, V.data_ptr(), M.data_ptr(), dU.data_ptr(), P, T, cin, cout, 0))
            t_fg = timeit(lambda: getattr(lib, fn + "filter_grad")(ctx.h, dU.data_ptr(), cin, cout, cout, dw.data_ptr(), 0))
            line += "\n    F(%dx%d): in %5.1f + gemm %6.1f + out %5.1f = %6.1f us (%5.1f TF/s eff.), filter %5.1f us; %d tiles | wgrad: dy %5.1f + gemm %6.1f + filter-grad %5.1f = %6.1f us" % (
                form, form, t_i, t_g, t_o, t_i + t_g + t_o, fl / (t_i + t_g + t_o) / 1e6, t_f, T, t_dz, t_wg, t_fg, t_dz + t_wg + t_fg)
        print(line, flush=True)

if __name__ == "__main__":
    main()
