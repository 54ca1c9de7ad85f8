"""Which lane bounds the pipelined train step?  The same loop as bench.py's timed region (one synthetic 600x1000 sample, lookahead 4),
with parts of the GPU work REMOVED after priming -- diagnostic only, results of the variants are not a training step:
  full        the step as benched
  no-base     the frozen base forward of the prefetch lanes is not launched (the feature maps keep their values)
  no-head     the classifier phase (RoI batch, stage 5 forward / backward, Adam #2) is not launched
  no-base-no-head   both: the RPN cycle alone (host RNG order: RoI codes -> sample selection -> anchor subsampling -> RPN backward ...)
usage: python tools/lane_isolation_probe.py [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rock-art-radnet_amd")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import bench  # noqa: E402
from faster_rcnn.config import Config  # noqa: E402
from radnet_hip import make_engine, synth  # noqa: E402
from radnet_hip.trainer import TrainStep  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    C = Config()
    eng = make_engine(C)
    eng.set_weights(synth.synthetic_weights(seed=3))
    ts = TrainStep(eng)
    batch = bench.make_batch(0, 1, 600, 1000)
    look = getattr(ts, "LOOKAHEAD", 3)

    def run(n):
        for k in range(n):
            ts.step(batch, upcoming=[batch] * min(look, n - 1 - k))
        ts.flush()
        torch.cuda.synchronize()

    run(2 * getattr(ts, "NBUF", 6) + 12)
    orig = dict(base_forward=eng.base_forward, head_forward=eng.head_forward, head_backward=eng.head_backward, adam=eng.adam,
                pack_roi_batch=eng.pack_roi_batch, refresh_head_shift=eng.refresh_head_shift)

    def variant(name, no_base=False, no_head=False):
        for k, v in orig.items():
            setattr(eng, k, v)
        if no_base:
            eng.base_forward = lambda bp: bp["F"]
        if no_head:
            eng.head_forward = lambda *a, **k: None
            eng.head_backward = lambda *a, **k: None
            eng.pack_roi_batch = lambda *a, **k: None
            eng.refresh_head_shift = lambda *a, **k: None
            eng.adam = lambda arena, **k: (None if arena is eng.head_arena else orig["adam"](arena, **k))
        run(30)
        best = 1e9
        for rep in range(3):
            t = time.perf_counter()
            run(steps)
            best = min(best, (time.perf_counter() - t) / steps)
        print("%-18s %7.1f us per step  (%.1f steps/s)" % (name, best * 1e6, 1.0 / best), flush=True)

    variant("full")
    variant("no-base", no_base=True)
    variant("no-head", no_head=True)
    variant("no-base-no-head", no_base=True, no_head=True)
    variant("full (again)")


if __name__ == "__main__":
    main()
