"""What the pipelined step is short of: extra work of one kind is enqueued on the RPN lane (53 % occupied) once per step and the
period's response is read off -- d(period)/d(HBM bytes) with a device-to-device copy, d(period)/d(fp32 matrix flops) with a
GEMM whose operands stay in cache.  usage: python tools/sensitivity_probe.py [steps]"""
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
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    eng = make_engine(Config())
    eng.set_weights(synth.synthetic_weights(seed=3))
    ts = TrainStep(eng)
    batch = bench.make_batch(0, 1, 600, 1000)
    look = getattr(ts, "LOOKAHEAD", 3)
    src = torch.empty(64 << 20, dtype=torch.float32, device="cuda")       # 256 MB
    dst = torch.empty_like(src)
    A = torch.randn(1024, 1024, device="cuda")
    B = torch.randn(1024, 1024, device="cuda")
    Cm = torch.empty(1024, 1024, device="cuda")

    def alone(fn, n=20):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6

    def copy_mb(mb):
        k = (mb << 20) // 8                 # mb of traffic = read + write of mb/2
        return lambda: dst[:k].copy_(src[:k])

    def mm(reps):
        def f():
            for _ in range(reps):
                torch.mm(A, B, out=Cm)      # 2.1 GFLOP each, 12 MB of operands (L2-resident)
        return f

    cases = [("nothing", None), ("+ 100 MB of copy traffic", copy_mb(100)), ("+ 200 MB", copy_mb(200)), ("+ 400 MB", copy_mb(400)),
             ("+ 4.3 GFLOP fp32 GEMM (2 x 1024^3)", mm(2)), ("+ 8.6 GFLOP", mm(4)), ("+ 17.2 GFLOP", mm(8)), ("nothing (again)", None)]
    for _ in range(2 * getattr(ts, "NBUF", 6) + 6):
        ts.step(batch, upcoming=[batch] * look)
    torch.cuda.synchronize()
    base = None
    for name, extra in cases:
        t_alone = alone(extra) if extra else 0.0

        def run(n):
            for _ in range(n):
                if extra:
                    extra()                  # current stream = the RPN lane's
                ts.step(batch, upcoming=[batch] * look)
        run(30)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(steps)
        torch.cuda.synchronize()
        per = (time.perf_counter() - t0) / steps * 1e6
        base = per if base is None else base
        print("%-40s alone %6.1f us   step %7.1f us  (%+6.1f us = %4.2f x its alone time)" % (name, t_alone, per, per - base, (per - base) / t_alone if t_alone else 0.0), flush=True)
    ts.flush()


if __name__ == "__main__":
    main()
