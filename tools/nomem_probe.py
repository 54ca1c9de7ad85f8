"""Is a launch bound by the latency of its operand loads?  The diagnostic twin of the library (make -C rock-art-radnet_amd/csrc diag;
RADNET_HIP_LIBRARY=.../libradnet_hip_diag.so) answers every operand load with 0 WITHOUT touching memory when RADNET_DIAG_NOMEM is set
(empty buffer descriptors): same instructions, same LDS traffic, same MFMAs, no memory round trips.  The step's shapes with their
shipped launch shapes, each alone on the chip; run once without and once with RADNET_DIAG_NOMEM=1.  usage: nomem_probe.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rock-art-radnet_amd")]
from radnet_hip import lib as L  # noqa: E402
from tile_probe import CONV, BATCHED, timeit  # noqa: E402

SHAPES = {  # name prefix -> (tile_a, tile_b, slices, waves): the shipped batch-1 table
    "res5a_2a": (32, 64, -1, 4), "res5x_2a": (32, 32, 1, 4), "res5x_2b": (32, 64, -3, 4), "res5x_2c": (64, 64, 1, 4), "res5a_sc": (64, 128, -1, 8),
    "res4a_2a": (32, 32, 1, 4), "res4x_2a": (32, 32, 1, 4), "res4x_2c": (64, 64, 1, 4), "res4a_sc": (32, 64, -1, 4), "rpn heads": (32, 32, 1, 4),
    "res3x_2a": (32, 64, 1, 4), "res3x_2c": (64, 64, 1, 8), "stage-4 Winograd": (32, 64, -1, 4), "stage-3 Winograd": (64, 64, -1, 4),
    "rpn_conv1 Winograd": (32, 64, -1, 4),
}


def shape_of(name):
    for k, v in SHAPES.items():
        if name.startswith(k):
            return v
    return (64, 64, 1, 4)


def main():
    ctx = L.Context(0)
    lib = ctx.lib
    ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    ctx.check(lib.radnet_set_workspace(ctx.h, ws.data_ptr(), ws.numel()), "ws")
    print("RADNET_DIAG_NOMEM =", os.environ.get("RADNET_DIAG_NOMEM"), " library:", os.environ.get("RADNET_HIP_LIBRARY"))
    for name, nb, h, w, cin, cout, k, stride, pad in CONV:
        oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
        x = torch.randn(nb, h, w, cin, device="cuda").relu_()
        wt = torch.randn(k * k * cin, cout, device="cuda") / np.sqrt(k * k * cin)
        sc, sh = torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda")
        y = torch.empty(nb, oh, ow, cout, device="cuda")
        d = L.ConvDesc()
        d.x, d.w, d.y, d.scale, d.shift = x.data_ptr(), wt.data_ptr(), y.data_ptr(), sc.data_ptr(), sh.data_ptr()
        d.nb, d.h, d.w_, d.c, d.oh, d.ow = nb, h, w, cin, oh, ow
        d.kh, d.kw, d.stride, d.pad_t, d.pad_l, d.n = k, k, stride, pad, pad, cout
        d.ldw, d.ldy, d.ld_add, d.act, d.act_cols = cout, cout, cout, 1, 0
        a, b, s, wv = shape_of(name)
        lib.radnet_force_config(ctx.h, a, b, s)
        lib.radnet_force_waves(ctx.h, wv)
        t = timeit(lambda: lib.radnet_conv_fwd(ctx.h, C.byref(d)), n=50)
        print("%-36s tile %3dx%-3d s=%2d w=%d  %6.1f us" % (name, a, b, s, wv, t), flush=True)
    for name, batch, T, c, n in BATCHED:
        V = torch.randn(batch, T, c, device="cuda")
        U = torch.randn(batch, c, n, device="cuda")
        M = torch.empty(batch, T, n, device="cuda")
        a, b, s, wv = shape_of(name)
        lib.radnet_force_config(ctx.h, a, b, s)
        lib.radnet_force_waves(ctx.h, wv)
        t = timeit(lambda: lib.radnet_gemm_batched(ctx.h, V.data_ptr(), U.data_ptr(), M.data_ptr(), batch, T, n, c), n=50)
        print("%-36s tile %3dx%-3d s=%2d w=%d  %6.1f us" % (name, a, b, s, wv, t), flush=True)


if __name__ == "__main__":
    main()
