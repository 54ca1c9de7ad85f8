"""Adam #2 over the classifier arena alone on the chip: radnet_adam_step_affine against radnet_adam_step_fused (+ the three Winograd filter
transforms in the same pass) against affine + three radnet_winograd4_filter launches.  usage: python tools/adam_timing.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rock-art-radnet_amd")]
from faster_rcnn.config import Config  # noqa: E402
from radnet_hip import make_engine, synth  # noqa: E402


def main():
    eng = make_engine(Config())
    eng.set_weights(synth.synthetic_weights(seed=3))
    eng.head_arena.g.normal_(0, 1e-3)

    def timed(label, fn, n=50):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print("%-64s %.1f us" % (label, e0.elapsed_time(e1) * 1e3 / n), flush=True)

    names = list(eng.INFERENCE_WINOGRAD_LAYERS)
    timed("fused: Adam + shifts + 3 filter transforms, one launch", lambda: eng.adam(eng.head_arena, zero_grad=False))
    eng.head_train_wino = False
    timed("Adam + shifts (radnet_adam_step_affine)", lambda: eng.adam(eng.head_arena, zero_grad=False))
    timed("Adam + shifts, then 3 x radnet_winograd4_filter", lambda: (eng.adam(eng.head_arena, zero_grad=False), eng._refresh_winograd(names)))
    timed("3 x radnet_winograd4_filter", lambda: eng._refresh_winograd(names))


if __name__ == "__main__":
    main()
