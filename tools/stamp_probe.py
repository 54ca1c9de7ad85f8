"""Where does a conv GEMM launch spend its time?  Per-workgroup s_memtime / s_memrealtime stamps from the diagnostic
twin of the library (make -C rock-art-radnet_amd/csrc diag), summarised per launch configuration.

Stamps per workgroup: start, first tile in LDS, end of the K loop, end of the epilogue (stores drained), the
100 MHz real-time counter at start and end, HW_ID / XCC_ID, K tiles done.  Nothing here is part of the product.

usage: python tools/stamp_probe.py            (runs the built-in list of ResNet50 @ 1000x600 layer shapes)
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rock-art-radnet_amd")]

from radnet_hip import lib as L  # noqa: E402

L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), os.environ.get("RADNET_DIAG_LIB", "libradnet_hip_diag.so"))

# name, nb, h, w, cin, cout, k, stride, pad, residual, (tile_m, tile_n, slices)
SHAPES = [
    ("res4x_2c  1x1 256->1024 +res", 1, 38, 63, 256, 1024, 1, 1, 0, True, (64, 64, 1)),
    ("res4x_2a  1x1 1024->256", 1, 38, 63, 1024, 256, 1, 1, 0, False, (64, 64, -3)),
    ("res4x_2a  1x1 1024->256 unsplit", 1, 38, 63, 1024, 256, 1, 1, 0, False, (64, 64, 1)),
    ("res4x_2b  3x3 256->256", 1, 38, 63, 256, 256, 3, 1, 1, False, (64, 64, 5)),
    ("res4x_2b  3x3 256->256 unsplit", 1, 38, 63, 256, 256, 3, 1, 1, False, (64, 64, 1)),
    ("res3x_2b  3x3 128->128", 1, 75, 125, 128, 128, 3, 1, 1, False, (64, 64, -3)),
    ("res3x_2b  3x3 128->128 unsplit", 1, 75, 125, 128, 128, 3, 1, 1, False, (64, 64, 1)),
    ("res3x_2b  3x3 128->128 128x64", 1, 75, 125, 128, 128, 3, 1, 1, False, (128, 64, 1)),
    ("res2x_2b  3x3 64->64", 1, 150, 250, 64, 64, 3, 1, 1, False, (64, 64, -1)),
    ("rpn_conv1 3x3 1024->512", 1, 38, 63, 1024, 512, 3, 1, 1, False, (64, 64, 5)),
    ("rpn_conv1 3x3 1024->512 128x128 x2", 1, 38, 63, 1024, 512, 3, 1, 1, False, (128, 128, 4)),
]

if os.environ.get("RADNET_PROBE_SET") == "head":          # the classifier's layers: 20 RoIs x 7x7 = 980 rows
    SHAPES = [
        ("res5x_2b  3x3 512->512 (s=-2)", 20, 7, 7, 512, 512, 3, 1, 1, False, (64, 64, -2)),
        ("res5x_2b  3x3 512->512 (s=-6)", 20, 7, 7, 512, 512, 3, 1, 1, False, (64, 64, -6)),
        ("res5x_2b  3x3 512->512 (s=4)", 20, 7, 7, 512, 512, 3, 1, 1, False, (64, 64, 4)),
        ("res5x_2c  1x1 512->2048 +res", 20, 7, 7, 512, 2048, 1, 1, 0, True, (64, 64, 1)),
        ("res5x_2a  1x1 2048->512 (s=2)", 20, 7, 7, 2048, 512, 1, 1, 0, False, (64, 64, 2)),
        ("res5x_2a  1x1 2048->512 (s=4)", 20, 7, 7, 2048, 512, 1, 1, 0, False, (64, 64, 4)),
        ("res5a_1   1x1 1024->2048", 20, 7, 7, 1024, 2048, 1, 1, 0, False, (64, 128, -1)),
    ]

WAVES = int(os.environ.get("RADNET_PROBE_WAVES", "4"))      # waves per workgroup of the probed launches (4 or 8)


def main():
    lib = L.load_library()
    lib.radnet_diag_set_stamps.restype = C.c_int
    lib.radnet_diag_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
    ctx = L.Context(0)
    ws = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    ctx.check(lib.radnet_set_workspace(ctx.h, ws.data_ptr(), ws.numel()), "ws")
    stamps = torch.zeros(1 << 16, 8, dtype=torch.int64, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(1)
    for name, nb, h, w, cin, cout, k, stride, pad, res, (bm, bn, s) in SHAPES:
        oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
        x = torch.randn(nb, h, w, cin, device="cuda", generator=g).relu_()
        wt = torch.randn(k * k * cin, cout, device="cuda", generator=g) / np.sqrt(k * k * cin)
        sc = torch.rand(cout, device="cuda", generator=g) + 0.5
        sh = torch.randn(cout, device="cuda", generator=g)
        add = torch.randn(nb, oh, ow, cout, device="cuda", generator=g) if res else None
        y = torch.empty(nb, oh, ow, cout, device="cuda")
        d = L.ConvDesc()
        d.x, d.w, d.y, d.scale, d.shift = x.data_ptr(), wt.data_ptr(), y.data_ptr(), sc.data_ptr(), sh.data_ptr()
        d.addend = add.data_ptr() if res else None
        d.nb, d.h, d.w_, d.c, d.oh, d.ow = nb, h, w, cin, oh, ow
        d.kh, d.kw, d.stride, d.pad_t, d.pad_l, d.n = k, k, stride, pad, pad, cout
        d.ldw, d.ldy, d.ld_add, d.act, d.act_cols = cout, cout, cout, 1, 0
        ctx.check(lib.radnet_force_config(ctx.h, bm, bn, s), "force")
        ctx.check(lib.radnet_force_waves(ctx.h, WAVES), "force waves")
        ctx.check(lib.radnet_diag_set_stamps(ctx.h, None), "stamps off")
        for _ in range(300):
            ctx.check(lib.radnet_conv_fwd(ctx.h, C.byref(d)), "fwd")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            lib.radnet_conv_fwd(ctx.h, C.byref(d))
        e1.record()
        torch.cuda.synchronize()
        wall = e0.elapsed_time(e1) * 10.0          # us per launch (kernel + fix-up when split)
        stamps.zero_()
        ctx.check(lib.radnet_diag_set_stamps(ctx.h, stamps.data_ptr()), "stamps on")
        for _ in range(20):
            lib.radnet_conv_fwd(ctx.h, C.byref(d))
        torch.cuda.synchronize()
        st = stamps.cpu().numpy().astype(np.uint64)
        st = st[st[:, 3] != 0]
        n = len(st)
        t = st[:, :4].astype(np.float64)
        rt0, rt1 = st[:, 4].astype(np.float64), st[:, 5].astype(np.float64)
        clk = np.median((t[:, 3] - t[:, 0]) / np.maximum(rt1 - rt0, 1.0)) * 100.0      # MHz
        cyc_us = 1.0 / clk
        base = rt0.min()
        start = (rt0 - base) / 100.0
        end = (rt1 - base) / 100.0
        pro, loop, epi = (t[:, 1] - t[:, 0]) * cyc_us, (t[:, 2] - t[:, 1]) * cyc_us, (t[:, 3] - t[:, 2]) * cyc_us
        hw = st[:, 6]
        cu = ((hw >> np.uint64(32)) & np.uint64(0xF)) * np.uint64(256) + ((hw >> np.uint64(8)) & np.uint64(0xFF))
        ncu = len(np.unique(cu))
        per_cu = np.bincount(np.unique(cu, return_inverse=True)[1])
        kt = st[:, 7].astype(np.float64)
        M, K = nb * oh * ow, k * k * cin
        fl = 2.0 * M * cout * K
        q = lambda a: "%.1f/%.1f/%.1f" % (np.percentile(a, 10), np.median(a), np.percentile(a, 90))  # noqa: E731
        print("%s  [%dx%d s=%d waves=%d]  M=%d N=%d K=%d" % (name, bm, bn, s, WAVES, M, cout, K))
        print("   wall %.1f us/launch (%.1f TF/s); %d workgroups on %d CUs (per CU %d..%d); in-kernel clock %.0f MHz" % (
            wall, fl / wall / 1e6, n, ncu, per_cu.min(), per_cu.max(), clk))
        print("   span first start -> last end %.1f us; starts p10/50/90 %s max %.1f; ends p10/50/90 %s" % (
            end.max(), q(start), start.max(), q(end)))
        print("   per workgroup us p10/50/90: prologue %s | K loop %s (%.0f tiles, %.0f cyc/tile median) | epilogue %s" % (
            q(pro), q(loop), np.median(kt), np.median((t[:, 2] - t[:, 1]) / np.maximum(kt, 1)), q(epi)))
        # second-round workgroups: started after some other workgroup had ended
        late = start > end.min()
        print("   workgroups started after the first one finished: %d; MFMA-ideal %.1f us at this clock" % (
            int(late.sum()), fl / (256 * 4 * 64 * clk) ))
    ctx.check(lib.radnet_force_config(ctx.h, 0, 0, 0), "unforce")


if __name__ == "__main__":
    main()
