"""Host-side timeline of the train step: where the Python thread spends its time between the phase boundaries
(enqueue cost vs waits for the GPU).  usage: python tools/host_timeline.py [steps]"""
import collections
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
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    C = Config()
    nccl1 = os.environ.get("RADNET_BENCH_REHEARSAL") == "nccl1"          # 1-rank RCCL group: both exchanges issued
    if nccl1:
        import torch.distributed as dist
        from radnet_hip import trainer as _tr
        _tr.FORCE_COLLECTIVES = True
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29534")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    eng = make_engine(C)
    eng.set_weights(synth.synthetic_weights(seed=3))
    ts = TrainStep(eng, defer_head_update=True if nccl1 else None)
    batch = bench.make_batch(0, 1, 600, 1000)
    pipelined = os.environ.get("RADNET_TIMELINE_SERIAL", "0") != "1"       # as bench.py runs it: next batch announced
    look = getattr(ts, "LOOKAHEAD", 3)
    for _ in range(2 * getattr(ts, "NBUF", 6) + 6):          # priming as bench.py: every shape tuned, every program recorded
        ts.step(batch, upcoming=[batch] * look if pipelined else None)
    torch.cuda.synchronize()
    acc = collections.OrderedDict()
    # split the label-map phase: time inside the host subsampling itself
    from radnet_hip import engine as E
    sub = [0.0]
    orig = E.subsample_valid

    def timed(*a, **k):
        t = time.perf_counter()
        r = orig(*a, **k)
        sub[0] += time.perf_counter() - t
        return r

    E.subsample_valid = timed
    ev = [0.0]
    orig_sync = torch.cuda.Event.synchronize

    def timed_sync(self):
        t = time.perf_counter()
        orig_sync(self)
        ev[0] += time.perf_counter() - t

    torch.cuda.Event.synchronize = timed_sync
    t_all = time.perf_counter()
    for _ in range(steps):
        ts.host_marks = []
        ts.step(batch, upcoming=[batch] * look if pipelined else None)
        m = ts.host_marks
        for (l0, t0), (l1, t1) in zip(m[:-1], m[1:]):
            acc[l1] = acc.get(l1, 0.0) + (t1 - t0)
    ts.flush()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t_all) / steps
    print("step wall %.0f us (host thread, %d steps)" % (wall * 1e6, steps))
    for k, v in acc.items():
        print("  %7.0f us  -> %s" % (v / steps * 1e6, k))
    print("  of the label-map phase, %.0f us is the host subsampling (NumPy RNG draws)" % (sub[0] / steps * 1e6))
    print("  Event.synchronize total %.0f us per step" % (ev[0] / steps * 1e6))


if __name__ == "__main__":
    main()
