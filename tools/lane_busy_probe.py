"""Which lane bounds the pipelined step?  HIP-event brackets (no tracer) around every block the step enqueues on a lane: the
frozen base forwards on the prefetch lanes, the classifier phase on the head lane, the RPN phase (loss .. RoI labels) on the
main lane.  A bracket opens when the lane reaches it (its earlier work done) and closes after its last kernel, so its length is
the block's execution time under co-scheduling, dependency waits inside it included.  Prints mean block times, the step period
and each lane's occupied share.  usage: python tools/lane_busy_probe.py [steps]"""
import contextlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rock-art-radnet_amd")]

import bench  # noqa: E402
from faster_rcnn.config import Config  # noqa: E402
from radnet_hip import make_engine, synth  # noqa: E402
from radnet_hip.trainer import TrainStep  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    C = Config()
    nccl1 = os.environ.get("RADNET_BENCH_REHEARSAL") == "nccl1"          # 1-rank RCCL group: the data-parallel structure of the step
    if nccl1:
        import torch.distributed as dist
        from radnet_hip import trainer as _tr
        _tr.FORCE_COLLECTIVES = True
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    eng = make_engine(C)
    eng.set_weights(synth.synthetic_weights(seed=3))
    ts = TrainStep(eng, defer_head_update=True if nccl1 else None)
    batch = bench.make_batch(0, 1, 600, 1000)
    look = ts.LOOKAHEAD
    for _ in range(2 * ts.NBUF + 6):
        ts.step(batch, upcoming=[batch] * look)
    torch.cuda.synchronize()
    rec = {}
    on = [False]
    orig_lane = eng.lane

    @contextlib.contextmanager
    def lane(name):
        with orig_lane(name):
            if not on[0]:
                yield
                return
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            yield
            e1.record()
            rec.setdefault(name, []).append((e0, e1))

    eng.lane = lane
    orig_rpn = ts._rpn_phase

    def rpn_phase(st, ntot, mark):
        if not on[0]:
            return orig_rpn(st, ntot, mark)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig_rpn(st, ntot, mark)
        e1.record()
        rec.setdefault("main: rpn phase", []).append((e0, e1))
        return r

    ts._rpn_phase = rpn_phase
    on[0] = True
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(steps):
        ts.step(batch, upcoming=[batch] * look)
    ts.flush()
    t1.record()
    torch.cuda.synchronize()
    period = t0.elapsed_time(t1) / steps
    print("step period %.3f ms (%d steps, brackets on: %.1f images/s)" % (period, steps, 1e3 / period))
    for name in sorted(rec):
        d = np.array([a.elapsed_time(b) for a, b in rec[name]])
        per_step = d.sum() / steps
        print("  %-18s %4d blocks  mean %.3f ms  p10 %.3f  p90 %.3f   occupied %.3f ms per step = %4.1f %% of the period" % (
            name, len(d), d.mean(), np.percentile(d, 10), np.percentile(d, 90), per_step, 100 * per_step / period))


if __name__ == "__main__":
    main()
