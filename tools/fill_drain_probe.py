"""Where do the fill and the drain of a short timed run go?  bench.py's protocol (pipeline drained before the clock, K steps,
flush) with a HIP event after every classifier phase and host timestamps at every step() return.
usage: python tools/fill_drain_probe.py [K]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rock-art-radnet_amd")]

import bench  # noqa: E402
from faster_rcnn.config import Config  # noqa: E402
from radnet_hip import make_engine, synth  # noqa: E402
from radnet_hip.trainer import TrainStep  # noqa: E402


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    eng = make_engine(Config())
    eng.set_weights(synth.synthetic_weights(seed=3))
    ts = TrainStep(eng)
    batch = bench.make_batch(0, 1, 600, 1000)
    look = ts.LOOKAHEAD
    for k in range(int(os.environ.get("PROBE_PRIME", 2 * ts.NBUF + 6))):
        ts.step(batch, upcoming=[batch] * look)
    ts.flush()
    torch.cuda.synchronize()
    # fine-grained: which host-side call blocks for milliseconds?
    long_calls = []

    def wrap(obj, name, label):
        orig = getattr(obj, name)

        def f(*a, **k):
            t = time.perf_counter()
            r = orig(*a, **k)
            d = time.perf_counter() - t
            if d > 1e-3:
                long_calls.append((label, round(d * 1e3, 2)))
            return r
        setattr(obj, name, f)

    wrap(torch.cuda.Event, "synchronize", "Event.synchronize")
    wrap(torch.cuda.CUDAGraph, "replay", "CUDAGraph.replay")
    wrap(torch.cuda.Event, "record", "Event.record")
    wrap(torch.cuda.Stream, "wait_event", "Stream.wait_event")
    for nm in ("upload_image", "upload_images", "base_forward", "anchor_targets_launch", "anchor_targets_finish", "roi_targets_finish",
               "pack_roi_batch", "head_forward", "head_backward", "adam", "proposals", "roi_targets_launch", "rpn_loss_backward"):
        if hasattr(eng, nm):
            wrap(eng, nm, "eng." + nm)
    import gc
    gc.collect()
    gc.freeze()
    for rep in range(6):
        ev0 = torch.cuda.Event(enable_timing=True)
        ends, host = [], []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev0.record()
        slow = []
        for k in range(K):
            ts.host_marks = []
            tk = time.perf_counter()
            ts.step(batch, upcoming=[batch] * min(look, K - 1 - k))
            host.append(time.perf_counter() - t0)
            if time.perf_counter() - tk > 2.5e-3:          # a slow call: which phase took it?
                m = ts.host_marks
                slow.append((k, [(l1, round((t1 - a) * 1e3, 2)) for (l0, a), (l1, t1) in zip(m[:-1], m[1:]) if t1 - a > 3e-4]))
            e = torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(eng.head_stream):
                e.record()
            ends.append(e)
        ts.flush()
        torch.cuda.synchronize()
        total = (time.perf_counter() - t0) * 1e3
        gpu = [ev0.elapsed_time(e) for e in ends]
        print("run %d: %d steps in %.2f ms = %.3f ms per step (%.1f images/s)" % (rep, K, total, total / K, 1e3 * K / total))
        print("   head phase k finished at (ms):", " ".join("%.2f" % g for g in gpu))
        print("   deltas:", " ".join("%.2f" % (b - a) for a, b in zip([0.0] + gpu[:-1], gpu)))
        print("   host returned from step k at (ms):", " ".join("%.2f" % (h * 1e3) for h in host))
        for k, phases in slow:
            print("   slow call %d:" % k, phases)
        print("   host calls over 1 ms:", long_calls)
        del long_calls[:]


if __name__ == "__main__":
    main()
