"""rpn.rpn_to_roi on the device at the 1000x600 step's size (38x63x12 = 28 728 candidates, scores from the synthetic-weight
network): the one-workgroup radix-select / LDS sort / integer NMS path against the former full-sort path (rocPRIM radix sort,
7 launches, + fp64 NMS: the default) -- the former is RADNET_PROPOSALS_SELECT=1.  usage: python tools/proposals_timing.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rock-art-radnet_amd")]

import bench  # noqa: E402
from faster_rcnn.config import Config  # noqa: E402
from radnet_hip import make_engine, synth  # noqa: E402


def main():
    C = Config()
    eng = make_engine(C)
    eng.set_weights(synth.synthetic_weights(seed=3))
    for (h, w) in ((600, 1000), (600, 600), (1000, 1000)):
        img = synth.synthetic_panel(1, h, w)
        bp = eng.upload_image(img)
        eng.base_forward(bp)
        rp = eng.rpn_forward(bp)
        out = []
        for name, env in (("select+sort+nms (1 launch)", "1"), ("rocPRIM sort + fp64 nms", None)):
            if env:
                os.environ["RADNET_PROPOSALS_SELECT"] = env
            else:
                os.environ.pop("RADNET_PROPOSALS_SELECT", None)
            us = bench._time_us(lambda: eng.proposals(rp, 0.7, 300), n=50)
            n = int(rp["Rn"].cpu()[0])
            out.append("%s: %6.1f us (%d RoIs)" % (name, us, n))
        os.environ.pop("RADNET_PROPOSALS_SELECT", None)
        print("%dx%d  fmap %dx%d  %d candidates | %s" % (w, h, rp["fw"], rp["fh"], rp["fw"] * rp["fh"] * eng.A, " | ".join(out)), flush=True)


if __name__ == "__main__":
    main()
