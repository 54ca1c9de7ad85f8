"""Is the pipelined step bound by the host/GPU cycle "RoI codes -> sample selection -> anchor subsampling -> RPN backward ->
proposals -> RoI codes"?  Times the step as bench.py runs it with the host subsampling (a) as shipped, (b) replaced by a trivial
stand-in that draws nothing (WRONG labels: timing experiment only), (c) with a busy-wait added.  If the cycle bounds the step,
the step time moves 1:1 with the host time on it.   usage: python tools/cycle_probe.py [steps]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rock-art-radnet_amd")]

import bench  # noqa: E402
from faster_rcnn.config import Config  # noqa: E402
from radnet_hip import make_engine, synth  # noqa: E402
from radnet_hip import engine as E  # noqa: E402
from radnet_hip.trainer import TrainStep  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    eng = make_engine(Config())
    eng.set_weights(synth.synthetic_weights(seed=3))
    ts = TrainStep(eng)
    batch = bench.make_batch(0, 1, 600, 1000)
    look = getattr(ts, "LOOKAHEAD", 3)
    for _ in range(2 * getattr(ts, "NBUF", 6) + 6):
        ts.step(batch, upcoming=[batch] * look)
    torch.cuda.synchronize()
    orig = E.subsample_valid
    sub = [0.0]

    def timed(fn):
        def f(*a, **k):
            t = time.perf_counter()
            r = fn(*a, **k)
            sub[0] += time.perf_counter() - t
            return r
        return f

    def trivial(valid, overlap, max_regions=256):
        vf, of = valid.reshape(-1), overlap.reshape(-1)
        live = vf == 1
        pos = np.flatnonzero(live & (of == 1))
        neg = np.flatnonzero(live & (of == 0))
        half = max_regions // 2
        if len(pos) > half:
            vf[pos[half:]] = 0
        n_pos = min(len(pos), half)
        if len(neg) + n_pos > max_regions:
            vf[neg[max_regions - n_pos:]] = 0
        return n_pos

    def delayed(us):
        def f(*a, **k):
            r = orig(*a, **k)
            t = time.perf_counter() + us * 1e-6
            while time.perf_counter() < t:
                pass
            return r
        return f

    cases = [("as shipped", orig), ("trivial stand-in (no draws)", trivial), ("+100 us busy-wait", delayed(100)), ("+300 us busy-wait", delayed(300)),
             ("as shipped (again)", orig)]
    for name, fn in cases:
        E.subsample_valid = timed(fn)
        for _ in range(30):
            ts.step(batch, upcoming=[batch] * look)
        torch.cuda.synchronize()
        sub[0] = 0.0
        t0 = time.perf_counter()
        for _ in range(steps):
            ts.step(batch, upcoming=[batch] * look)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / steps
        print("%-32s step %7.1f us  (%.1f images/s)   host subsampling %6.1f us" % (name, wall * 1e6, 1.0 / wall, sub[0] / steps * 1e6), flush=True)
    ts.flush()


if __name__ == "__main__":
    main()
