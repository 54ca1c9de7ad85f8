"""radnet_roi_resize_bwd alone on the chip: ordered form (a gather per feature-map pixel since round 4: 41 us; the row-in-LDS form before it
155-165 us with 1, 2 or 4 waves per row) against the atomics form (50 us), 20 RoIs of the sizes the RPN proposes on the 38x63 map, 14x14 crops,
1024 channels.  usage: python tools/roi_bwd_timing.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rock-art-radnet_amd")]
from radnet_hip import lib as L  # noqa: E402


def main():
    ctx = L.Context(0)
    rs = np.random.RandomState(5)
    R, ps, C, H, W = 20, 14, 1024, 38, 63
    rois = np.zeros((R, 4), np.float32)
    for r in range(R):
        w, h = rs.randint(3, 30), rs.randint(3, 24)
        rois[r] = (rs.randint(0, W - w), rs.randint(0, H - h), w, h)
    dy = torch.from_numpy(rs.standard_normal((R, ps, ps, C)).astype(np.float32)).cuda()
    rd = torch.from_numpy(rois).cuda()
    dF = torch.zeros(1, H, W, C, device="cuda")
    for det in (1, 0):
        ctx.check(ctx.lib.radnet_set_deterministic(ctx.h, det), "det")
        for _ in range(5):
            ctx.call("radnet_roi_resize_bwd", dy, H, W, C, rd, R, ps, dF)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            ctx.call("radnet_roi_resize_bwd", dy, H, W, C, rd, R, ps, dF)
        e1.record()
        torch.cuda.synchronize()
        print("%s: %.1f us per call" % ("ordered (gather per pixel)" if det else "atomics", e0.elapsed_time(e1) * 1e3 / 50), flush=True)


if __name__ == "__main__":
    main()
