"""Why do back-to-back bench runs differ by up to 8 %?  Per run: the bench value, the GPU time of the same steps
(events), and a fixed GEMM (rpn_conv1 forward, 200 launches) timed right before and after as a clock witness."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rock-art-radnet_amd")]

import bench  # noqa: E402
from faster_rcnn.config import Config  # noqa: E402
from radnet_hip import lib as L, synth  # noqa: E402
from radnet_hip.engine import FasterRCNNEngine  # noqa: E402
from radnet_hip.trainer import TrainStep  # noqa: E402


def witness(eng):
    c = eng.convs["rpn_conv1"]
    x = torch.randn(1, 38, 63, 1024, device="cuda").relu_()
    y = torch.empty(1, 38, 63, 512, device="cuda")
    d, _, _ = eng._desc(c, x, 1, 38, 63, y)
    for _ in range(20):
        eng.lib.radnet_conv_fwd(eng.ctx.h, C.byref(d))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        eng.lib.radnet_conv_fwd(eng.ctx.h, C.byref(d))
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 5.0      # us per launch


def main():
    Cc = Config()
    eng = FasterRCNNEngine(Cc)
    eng.set_weights(synth.synthetic_weights(seed=3))
    ts = TrainStep(eng)
    batch = bench.make_batch(0, 1, 600, 1000)
    np.random.seed(64)
    for _ in range(5):
        ts.step(batch, next_batch=batch)
    torch.cuda.synchronize()
    import gc
    gc_log, gc_t = [], [0.0]

    def on_gc(phase, info):
        if phase == "start":
            gc_t[0] = time.perf_counter()
        else:
            gc_log.append((info["generation"], (time.perf_counter() - gc_t[0]) * 1e3, info["collected"]))

    gc.callbacks.append(on_gc)
    if os.environ.get("PROBE_GC_FREEZE", "0") == "1":
        gc.collect()
        gc.freeze()
    for rep in range(int(os.environ.get("PROBE_REPS", "8"))):
        w0 = witness(eng)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t = time.perf_counter()
        e0.record()
        n = 30
        slow = []
        for k in range(n):
            ts.host_marks = []
            ts.step(batch, next_batch=batch if k + 1 < n else None)
            m = ts.host_marks
            for (l0, t0), (l1, t1) in zip(m[:-1], m[1:]):
                slow.append((t1 - t0, k, l1))
        ts.host_marks = None
        e1.record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t) / n * 1e3
        gpu = e0.elapsed_time(e1) / n
        w1 = witness(eng)
        print("rep %d: %.3f ms/step wall (%.1f img/s), %.3f ms/step between GPU events; witness GEMM %.1f us before, %.1f us after" % (
            rep, wall, 1e3 / wall, gpu, w0, w1), flush=True)
        if gc_log:
            print("      gc: " + "; ".join("gen%d %.2f ms (%d freed)" % g for g in gc_log if g[0] >= 1 or g[1] > 1.0), flush=True)
            del gc_log[:]
        slow.sort(reverse=True)
        print("      longest host intervals: " + "; ".join("%.2f ms step %d -> %s" % (d * 1e3, k, l[:38]) for d, k, l in slow[:4]), flush=True)


if __name__ == "__main__":
    main()
