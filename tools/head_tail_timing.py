"""Fused classifier-tail kernels (csrc/head_tail.hip) against the five separate launches, alone on the chip."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rock-art-radnet_amd")]
import bench  # noqa: E402
from radnet_hip import lib as L  # noqa: E402

def main():
    ctx = L.Context(0)
    R, nc, nreg, hw, c, ld = int(os.environ.get("R", "20")), 7, 24, 49, 2048, 32
    y5 = torch.randn(R, hw, c, device="cuda").relu_()
    w = torch.randn(c, ld, device="cuda") * 0.02; b = torch.zeros(ld, device="cuda")
    y1 = torch.zeros(R, nc, device="cuda"); y1[:, 0] = 1
    y2 = torch.zeros(R, 2 * nreg, device="cuda"); y2[:, :4] = 1
    z = lambda *s: torch.zeros(*s, device="cuda")
    feat, pc, pr, dz, dw, db, df, gl, Ls = z(R, c), z(R, nc), z(R, nreg), z(R, nc + nreg), z(c, ld), z(ld), z(R, c), z(R, hw, c), z(1, 3)
    scratch = torch.zeros(int(ctx.lib.radnet_head_tail_scratch_bytes(R)), dtype=torch.uint8, device="cuda")
    t = lambda f: bench._time_us(f, n=200, warm=20)
    print("separate: avgpool_fwd %.1f  dense_fwd %.1f  det_loss %.1f  dense_bwd %.1f  avgpool_bwd %.1f us" % (
        t(lambda: ctx.call("radnet_avgpool_fwd", y5, R, hw, c, feat)),
        t(lambda: ctx.call("radnet_dense_heads_fwd", feat, R, c, w, ld, b, nc, nreg, pc, pr)),
        t(lambda: ctx.call("radnet_det_loss", pc, pr, y1, y2, R, nc, nreg, dz, Ls)),
        t(lambda: ctx.call("radnet_dense_heads_bwd", feat, dz, R, c, w, ld, nc + nreg, dw, db, df, 0)),
        t(lambda: ctx.call("radnet_avgpool_bwd_relu", df, y5, R, hw, c, gl))))
    print("fused:    fwd (inference) %.1f  fwd (+losses) %.1f us" % (
        t(lambda: ctx.call("radnet_head_tail_fwd", y5, R, hw, c, w, ld, b, nc, nreg, feat, pc, pr, None, None, None, None, 1, None, scratch)),
        t(lambda: ctx.call("radnet_head_tail_fwd", y5, R, hw, c, w, ld, b, nc, nreg, feat, pc, pr, y1, y2, dz, Ls, 1, None, scratch))))
    print("(back-to-back launches from Python: ~%.1f us of host time per call bounds these from below)" % t(lambda: ctx.call("radnet_fill_zero", Ls, L.C.c_uint64(12))))

if __name__ == "__main__":
    main()
