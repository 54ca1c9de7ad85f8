"""bench.py's exact sequence (priming with tapering announcements, W warm-up steps, flush, K timed steps) with every eager run /
graph capture of a layer program logged: is compile work left inside the timed region of a short run?
usage: python tools/bench_sequence_probe.py [W] [K]"""
import gc
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
from radnet_hip.trainer import TrainStep  # noqa: E402


def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    eng = make_engine(Config())
    eng.set_weights(synth.synthetic_weights(seed=3))
    ts = TrainStep(eng)
    batch = bench.make_batch(0, 1, 600, 1000)
    np.random.seed(64)
    phase = ["prime"]
    log = []
    orig_eager = eng._run_eager

    def run_eager(ops):
        capturing = torch.cuda.is_current_stream_capturing()
        t = time.perf_counter()
        r = orig_eager(ops)
        log.append((phase[0], "capture" if capturing else "eager", len(ops), round((time.perf_counter() - t) * 1e3, 2)))
        return r

    eng._run_eager = run_eager
    LOOK = ts.LOOKAHEAD
    n_prime = int(os.environ.get("PROBE_PRIME", 2 * ts.NBUF + 2))
    for k in range(n_prime):
        ts.step(batch, upcoming=[batch] * min(LOOK, n_prime - 1 - k))
    ts.flush()
    torch.cuda.synchronize()
    phase[0] = "warmup"
    for k in range(W):
        ts.step(batch, upcoming=[batch] * min(LOOK, W - 1 - k))
    ts.flush()
    torch.cuda.synchronize()
    gc.collect()
    gc.freeze()
    phase[0] = "timed"
    per = []
    t0 = time.perf_counter()
    for k in range(K):
        tk = time.perf_counter()
        ts.step(batch, upcoming=[batch] * min(LOOK, K - 1 - k))
        per.append(round((time.perf_counter() - tk) * 1e3, 2))
    ts.flush()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print("prime %d, warm-up %d, timed %d steps: %.2f ms = %.3f ms per step (%.1f images/s)" % (n_prime, W, K, el * 1e3, el * 1e3 / K, K / el))
    print("host ms per step() call:", per if K <= 40 else "(%d calls) over 3 ms: %s at steps %s" % (
        K, [x for x in per if x > 3], [i for i, x in enumerate(per) if x > 3]))
    for ph in ("prime", "warmup", "timed"):
        ev = [e for e in log if e[0] == ph]
        print("%-7s eager program runs %3d (%.1f ms), captures %3d (%.1f ms)" % (
            ph, sum(1 for e in ev if e[1] == "eager"), sum(e[3] for e in ev if e[1] == "eager"),
            sum(1 for e in ev if e[1] == "capture"), sum(e[3] for e in ev if e[1] == "capture")))


if __name__ == "__main__":
    main()
