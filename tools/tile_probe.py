"""Every launch shape of the forward GEMM kernel (tile x K slices x unit order x waves) on the step's problem shapes, each alone
on the chip: best time per tile family.  Round 4: do the 32-row tiles (32x64, 32x32) beat 64x64 + K slices where M is small?
usage: tile_probe.py [quick]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rock-art-radnet_amd")]

from radnet_hip import lib as L  # noqa: E402

# name, nb, h, w, cin, cout, k, stride, pad
CONV = [
    ("res5a_2a 1x1/2 1024->512 (20 RoIs)", 20, 14, 14, 1024, 512, 1, 2, 0),
    ("res5x_2a 1x1 2048->512", 20, 7, 7, 2048, 512, 1, 1, 0),
    ("res5x_2b 3x3 512->512", 20, 7, 7, 512, 512, 3, 1, 1),
    ("res5x_2c 1x1 512->2048", 20, 7, 7, 512, 2048, 1, 1, 0),
    ("res5a_sc 1x1/2 1024->2048", 20, 14, 14, 1024, 2048, 1, 2, 0),
    ("res4a_2a 1x1/2 512->256", 1, 75, 125, 512, 256, 1, 2, 0),
    ("res4x_2a 1x1 1024->256", 1, 38, 63, 1024, 256, 1, 1, 0),
    ("res4x_2c 1x1 256->1024", 1, 38, 63, 256, 1024, 1, 1, 0),
    ("res4a_sc 1x1/2 512->1024", 1, 75, 125, 512, 1024, 1, 2, 0),
    ("rpn heads 1x1 512->64", 1, 38, 63, 512, 64, 1, 1, 0),
    ("res3x_2a 1x1 512->128", 1, 75, 125, 512, 128, 1, 1, 0),
    ("res3x_2c 1x1 128->512", 1, 75, 125, 128, 512, 1, 1, 0),
]
# name, batch, T, c, n
BATCHED = [("stage-4 Winograd GEMMs", 36, 160, 256, 256), ("stage-3 Winograd GEMMs", 36, 608, 128, 128), ("rpn_conv1 Winograd GEMMs", 36, 160, 1024, 512)]

TILES = [(128, 128), (128, 64), (64, 128), (64, 64), (32, 64), (32, 32)]
SLICES = [1, 2, 3, 4, 6, 8]


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def sweep(ctx, fn, m, n, k, batched):
    lib = ctx.lib
    best = {}
    for bm, bn in TILES:
        if bn > 64 and n <= 64:
            continue
        for wv in ((4,) if bm < 64 else (4, 8)):
            for s in ([1, -1, 2, -2, 3, -3, 4, -4, 6, -6, 9, -9] if batched else [v * sg for v in SLICES for sg in (1, -1)]):
                if not batched and abs(s) > 1 and (k // 32) // abs(s) < 2:
                    continue
                if lib.radnet_force_config(ctx.h, bm, bn, s) != 0 or lib.radnet_force_waves(ctx.h, wv) != 0:
                    continue
                if fn() != 0:
                    continue
                t = timeit(fn)
                key = "%dx%d" % (bm, bn)
                if key not in best or t < best[key][0]:
                    best[key] = (t, s, wv)
    lib.radnet_force_config(ctx.h, 0, 0, 0)
    lib.radnet_force_waves(ctx.h, 0)
    return best


def main():
    ctx = L.Context(0)
    lib = ctx.lib
    ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    ctx.check(lib.radnet_set_workspace(ctx.h, ws.data_ptr(), ws.numel()), "ws")
    for name, nb, h, w, cin, cout, k, stride, pad in CONV:
        oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
        x = torch.randn(nb, h, w, cin, device="cuda").relu_()
        wt = torch.randn(k * k * cin, cout, device="cuda") / np.sqrt(k * k * cin)
        sc, sh = torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda")
        y = torch.empty(nb, oh, ow, cout, device="cuda")
        add = torch.randn(nb, oh, ow, cout, device="cuda")
        d = L.ConvDesc()
        d.x, d.w, d.y, d.scale, d.shift = x.data_ptr(), wt.data_ptr(), y.data_ptr(), sc.data_ptr(), sh.data_ptr()
        d.nb, d.h, d.w_, d.c, d.oh, d.ow = nb, h, w, cin, oh, ow
        d.kh, d.kw, d.stride, d.pad_t, d.pad_l, d.n = k, k, stride, pad, pad, cout
        d.ldw, d.ldy, d.ld_add, d.act, d.act_cols = cout, cout, cout, 1, 0
        if "2c" in name:
            d.addend = add.data_ptr()
        m = nb * oh * ow
        fl = 2.0 * m * cout * k * k * cin
        best = sweep(ctx, lambda: lib.radnet_conv_fwd(ctx.h, C.byref(d)), m, cout, k * k * cin, False)
        print("%-36s M=%5d N=%4d K=%4d | " % (name, m, cout, k * k * cin) +
              "  ".join("%s %5.1f us (s=%d w=%d, %4.1f TF)" % (kk, v[0], v[1], v[2], fl / v[0] / 1e6) for kk, v in sorted(best.items(), key=lambda kv: kv[1][0])),
              flush=True)
    for name, batch, T, c, n in BATCHED:
        V = torch.randn(batch, T, c, device="cuda")
        U = torch.randn(batch, c, n, device="cuda")
        M = torch.empty(batch, T, n, device="cuda")
        fl = 2.0 * batch * T * c * n
        best = sweep(ctx, lambda: lib.radnet_gemm_batched(ctx.h, V.data_ptr(), U.data_ptr(), M.data_ptr(), batch, T, n, c), T, n, c, True)
        print("%-36s %d x [%d x %d x %d] | " % (name, batch, T, n, c) +
              "  ".join("%s %5.1f us (s=%d w=%d, %4.1f TF)" % (kk, v[0], v[1], v[2], fl / v[0] / 1e6) for kk, v in sorted(best.items(), key=lambda kv: kv[1][0])),
              flush=True)


if __name__ == "__main__":
    main()
