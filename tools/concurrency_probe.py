"""Do two / three conv GEMM launches on different HIP streams (own contexts) add up?  Aggregate TFLOP/s of the same layer
run on 1, 2, 3 streams at once, for a few layers of the 1000x600 step.  usage: python tools/concurrency_probe.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rock-art-radnet_amd")]

from radnet_hip import lib as L  # noqa: E402

# name, rows (nb,h,w), cin, cout, k
CASES = [("res4 1x1 256->1024", (1, 38, 63), 256, 1024, 1), ("res4 1x1 1024->256", (1, 38, 63), 1024, 256, 1),
         ("res3 1x1 512->128", (1, 75, 125), 512, 128, 1), ("res2 1x1 64->256", (1, 150, 250), 64, 256, 1),
         ("res5 3x3 512 (20 RoIs)", (20, 7, 7), 512, 512, 3), ("res5 1x1 512->2048 (20 RoIs)", (20, 7, 7), 512, 2048, 1)]


def main():
    NS = int(os.environ.get("PROBE_STREAMS", "3"))
    force = os.environ.get("RADNET_FORCE_CONFIG")
    streams = [torch.cuda.Stream() for _ in range(NS)]
    ctxs = []
    keep = []
    for s in streams:
        c = L.Context(0, stream_handle=s.cuda_stream)
        ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
        keep.append(ws)
        c.check(c.lib.radnet_set_workspace(c.h, ws.data_ptr(), ws.numel()), "ws")
        c.check(c.lib.radnet_set_autotune(c.h, 1), "tune")
        if ctxs:
            c.check(c.lib.radnet_share_tuning(c.h, ctxs[0].h), "share")
        if force:
            fa, fb, fs = (int(v) for v in force.split(","))
            c.check(c.lib.radnet_force_config(c.h, fa, fb, fs), "force")
            if os.environ.get("PROBE_FORCE_WAVES"):
                c.check(c.lib.radnet_force_waves(c.h, int(os.environ["PROBE_FORCE_WAVES"])), "force waves")
        ctxs.append(c)
    lib = ctxs[0].lib
    for name, (nb, h, w), cin, cout, k in CASES:
        descs = []
        for i in range(NS):
            x = torch.randn(nb, h, w, cin, device="cuda").relu_()
            wt = torch.randn(k * k * cin, cout, device="cuda") / np.sqrt(k * k * cin)
            sc, sh = torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda")
            y = torch.empty(nb, h, w, cout, device="cuda")
            keep.extend([x, wt, sc, sh, y])
            d = L.ConvDesc()
            d.x, d.w, d.y, d.scale, d.shift = x.data_ptr(), wt.data_ptr(), y.data_ptr(), sc.data_ptr(), sh.data_ptr()
            d.nb, d.h, d.w_, d.c, d.oh, d.ow = nb, h, w, cin, h, w
            d.kh, d.kw, d.stride, d.pad_t, d.pad_l, d.n = k, k, 1, k // 2, k // 2, cout
            d.ldw, d.ldy, d.ld_add, d.act, d.act_cols = cout, cout, cout, 1, 0
            descs.append(d)
        with torch.cuda.stream(streams[0]):
            lib.radnet_conv_fwd(ctxs[0].h, C.byref(descs[0]))     # autotune once (shared table)
        torch.cuda.synchronize()
        fl = 2.0 * nb * h * w * cout * k * k * cin
        out = []
        for ns in sorted(set([1, 2, NS])):
            n = 60
            for rep in range(2):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for s in streams[:ns]:
                    s.wait_event(e0)
                for _ in range(n):
                    for i in range(ns):
                        lib.radnet_conv_fwd(ctxs[i].h, C.byref(descs[i]))
                for s in streams[:ns]:
                    torch.cuda.current_stream().wait_stream(s)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1)
            out.append("%d stream%s: %6.1f us/launch-set, %5.1f TF/s" % (ns, "s" if ns > 1 else " ", ms * 1e3 / n, ns * n * fl / ms / 1e9))
        print("%-30s %s" % (name, " | ".join(out)), flush=True)


if __name__ == "__main__":
    main()
