"""In-situ tuning inside a running job on a panel size no shipped table knows (BASELINE cfg 1's 600x800).
Three training jobs over an endless feed of one synthetic sample, each a fresh engine, rate over the job's last 300 steps:
  job 1  run_training(tune=True), empty cache: the walk runs over the job's own first steps, writes the table, training goes on
  job 2  run_training(tune=True), cache present: the table is loaded before the first step
  job 3  run_training(tune=False): autotuned (+ shipped, where shapes coincide) launch shapes only
usage: insitu_job_demo.py [H W] [budget_s]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rock-art-radnet_amd")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch


def main():
    H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (600, 800)
    budget = float(sys.argv[3]) if len(sys.argv) > 3 else 240.0
    from faster_rcnn import data_feed
    from faster_rcnn.config import Config
    from radnet_hip import synth
    from radnet_hip.engine import FasterRCNNEngine
    from radnet_hip.trainer import TrainStep
    cache = tempfile.mkdtemp(prefix="radnet_tuned_")
    meta = synth.synthetic_gt(2, n=8, src_w=2 * W, src_h=2 * H)
    sample = dict(img=synth.synthetic_panel(1, H, W), bboxes=meta["bboxes"], width=2 * W, height=2 * H)
    C = Config()

    def job(tune, label, extra=340):
        np.random.seed(64)
        eng = FasterRCNNEngine(C)
        eng.set_weights(synth.synthetic_weights(seed=3))
        ts = TrainStep(eng)
        marks = []                  # (steps done, host time) after every step; the first entry comes when the walk (if any) is over
        state = {"stop_at": None}

        def on_step(n, _ts):
            marks.append((n, time.perf_counter()))
            if state["stop_at"] is None:
                state["stop_at"] = n + extra

        class Feed:                 # ends a few batches after the target, so that run_training drains its pipeline itself
            def __iter__(self):
                return self

            def __next__(self):
                if state["stop_at"] is not None and marks and marks[-1][0] >= state["stop_at"]:
                    raise StopIteration
                return sample

        t0 = time.perf_counter()
        data_feed.run_training(ts, Feed(), 10 ** 9, lookahead=3, tune=tune, tune_budget_s=budget, tune_cache_dir=cache,
                               log=lambda m: print(m, flush=True), on_step=on_step)
        torch.cuda.synchronize()
        t_end = time.perf_counter()
        n_end = marks[-1][0]
        t_300 = next(t for n, t in marks if n >= n_end - 300)
        us = (t_end - t_300) / 300 * 1e6
        print("%-66s %.1f us per step (%.1f images/s)  [job: %d steps, %.0f s]" % (label, us, 1e6 / us, n_end, t_end - t0), flush=True)
        del ts, eng
        torch.cuda.empty_cache()
        return us

    us_walk = job(True, "%dx%d job 1: in-situ walk over its first steps, then" % (H, W))
    us_cached = job(True, "%dx%d job 2: cached table loaded before the first step" % (H, W))
    us_plain = job(False, "%dx%d job 3: tune=False (autotuned + shipped shapes only)" % (H, W))
    print("gain of the in-situ table on this box: %.1f %% (job 1 after its walk), %.1f %% (job 2)" % (100 * (us_plain / us_walk - 1), 100 * (us_plain / us_cached - 1)))


if __name__ == "__main__":
    main()
