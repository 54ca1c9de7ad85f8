"""bench.py's exact sequence (prime 2*NBUF+2 steps, flush, W warm-up steps, flush, sync, gc.freeze, K timed steps, flush, sync) with a
HIP event after every classifier phase, repeated: is the FIRST timed region slower than a repetition of it, and where?
usage: python tools/bench_sequence_probe2.py [K] [W] [reps]"""
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
from radnet_hip import synth  # noqa: E402
from radnet_hip.engine import FasterRCNNEngine  # noqa: E402
from radnet_hip.trainer import TrainStep  # noqa: E402


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    eng = FasterRCNNEngine(Config(), device_index=0)
    eng.set_weights(synth.synthetic_weights(seed=3))
    ts = TrainStep(eng, world_size=1)
    batch = bench.make_batch(0, 1, 600, 1000)
    np.random.seed(64)
    LOOK = ts.LOOKAHEAD
    n_prime = 2 * ts.NBUF + 2
    for k in range(n_prime):
        ts.step(batch, upcoming=[batch] * min(LOOK, n_prime - 1 - k))
    ts.flush()
    torch.cuda.synchronize()
    for k in range(W):
        ts.step(batch, upcoming=[batch] * min(LOOK, W - 1 - k))
    ts.flush()
    torch.cuda.synchronize()
    gc.collect()
    gc.freeze()
    for rep in range(reps):
        ev0 = torch.cuda.Event(enable_timing=True)
        ends, host = [], []
        t0 = time.perf_counter()
        ev0.record()
        for k in range(K):
            ts.step(batch, upcoming=[batch] * min(LOOK, K - 1 - k))
            host.append((time.perf_counter() - t0) * 1e3)
            e = torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(eng.head_stream):
                e.record()
            ends.append(e)
        ts.flush()
        torch.cuda.synchronize()
        total = (time.perf_counter() - t0) * 1e3
        gpu = [ev0.elapsed_time(e) for e in ends]
        print("timed region %d: %d steps in %.2f ms = %.3f ms per step (%.1f images/s)" % (rep, K, total, total / K, 1e3 * K / total))
        print("   head phase deltas:", " ".join("%.2f" % (b - a) for a, b in zip([0.0] + gpu[:-1], gpu)))
        print("   host step returns:", " ".join("%.2f" % (b - a) for a, b in zip([0.0] + host[:-1], host)))


if __name__ == "__main__":
    main()
