#!/usr/bin/env python3
"""Frozen base forward at 1000x600 alone on the chip: the launch-by-launch layer program (hipGraph replay) against the chain
kernel (one persistent launch, include/radnet_hip.h) at several grid widths.  Prints ms per forward (mean of N replays between
two HIP events) and the executed / algorithmic TFLOP/s of the chain."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rock-art-radnet_amd")):
    sys.path.insert(0, p)
from faster_rcnn.config import Config  # noqa: E402
from radnet_hip import synth  # noqa: E402
from radnet_hip.engine import FasterRCNNEngine  # noqa: E402


def time_base(eng, bp, n=60):
    for _ in range(6):
        eng.base_forward(bp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        eng.base_forward(bp)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    H, W = (int(v) for v in (sys.argv[1:3] if len(sys.argv) > 2 else (600, 1000)))
    img = synth.synthetic_panel(1, H, W)
    Wt = synth.synthetic_weights(seed=3)
    C = Config()
    C.img_size = min(H, W)
    eng = FasterRCNNEngine(C)
    eng.set_weights(Wt)
    bp = eng.upload_image(img)
    F0 = eng.base_forward(bp).cpu().numpy()
    print("launch list (hipGraph replay): %.3f ms" % time_base(eng, bp))
    for wgs in (int(v) for v in os.environ.get("CHAIN_WGS_LIST", "128,256,384,512,768,1024").split(",")):
        e = FasterRCNNEngine(C)
        e.use_chain, e.chain_wgs = True, wgs
        e.set_weights(Wt)
        b = e.upload_image(img)
        F1 = e.base_forward(b).cpu().numpy()
        ms = time_base(e, b)
        err, runs, n_items, n_stages, fe, fa = e.chain_status(b)
        # the two launches in front of the chain (stem conv + max-pool) are inside both timings
        print("chain %4d workgroups: %.3f ms  (items %d, stages %d, error %d; stages 2-4: executed %.1f GF, algorithmic %.1f GF)   max |dF| / max |F| = %.2e"
              % (wgs, ms, n_items, n_stages, err, fe / 1e9, fa / 1e9, np.abs(F1 - F0).max() / np.abs(F0).max()))
        del e


if __name__ == "__main__":
    main()
