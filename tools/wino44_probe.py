"""Feasibility probe: the batched GEMMs a Winograd F(4x4,3x3) form would run (36 per layer, tiles/4) against the 16 of
F(2x2,3x3), each autotuned, alone on the chip."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rock-art-radnet_amd")]
from radnet_hip import lib as L  # noqa: E402
from winograd_timing import timeit  # noqa: E402

CASES = [("rpn_conv1", 1, 38, 63, 1024, 512), ("res4x_2b", 1, 38, 63, 256, 256), ("res3x_2b", 1, 75, 125, 128, 128),
         ("rpn_conv1 x2", 2, 38, 63, 1024, 512), ("res4x_2b x2", 2, 38, 63, 256, 256), ("res3x_2b x2", 2, 75, 125, 128, 128)]


def main():
    ctx = L.Context(0)
    lib = ctx.lib
    ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    ctx.check(lib.radnet_set_workspace(ctx.h, ws.data_ptr(), ws.numel()), "ws")
    ctx.check(lib.radnet_set_autotune(ctx.h, 1), "tune")
    for name, nb, h, w, cin, cout in CASES:
        fl = 2.0 * nb * h * w * cout * 9 * cin
        line = "%-14s" % name
        for tag, m, batch in (("F2", 2, 16), ("F4", 4, 36)):
            T = nb * ((h + m - 1) // m) * ((w + m - 1) // m)
            U = torch.randn(batch, cin, cout, device="cuda")
            V = torch.randn(batch, T, cin, device="cuda")
            M = torch.empty(batch, T, cout, device="cuda")
            t = timeit(lambda: ctx.check(lib.radnet_gemm_batched(ctx.h, V.data_ptr(), U.data_ptr(), M.data_ptr(), batch, T, cout, cin), "gemm"))
            ex = 2.0 * batch * T * cin * cout
            line += " | %s: %4d tiles x%d  %6.1f us  executed %5.1f TF/s  (layer-equivalent ceiling %5.1f TF/s)" % (tag, T, batch, t, ex / t / 1e6, fl / t / 1e6)
        print(line, flush=True)


if __name__ == "__main__":
    main()
