"""How often does a synthetic bench batch hit the reference's 'no RoI kept -> skip the classifier step' path?
usage: python tools/skip_probe.py [ranks] [steps]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rock-art-radnet_amd")]
import bench
from faster_rcnn.config import Config
from radnet_hip import synth
from radnet_hip.engine import FasterRCNNEngine
from radnet_hip.trainer import TrainStep

ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
C = Config()
eng = FasterRCNNEngine(C)
for r in range(ranks):
    eng.set_weights(synth.synthetic_weights(seed=3))
    for ar in (eng.rpn_arena, eng.head_arena):
        ar.m.zero_(); ar.v.zero_(); ar.t = 0
    ts = TrainStep(eng)
    ts.capture = []
    batch = bench.make_batch(r, 1, 600, 1000)
    np.random.seed(64 + r)
    skipped_at = []
    for k in range(steps):
        before = ts.skipped_head_steps
        ts.step(batch)
        if ts.skipped_head_steps != before:
            skipped_at.append(k)
    kept = [int(c["keep"].sum()) for c in ts.capture]
    print("rank-%d batch: skipped %d of %d steps at %s; kept RoIs min %d median %d" % (r, len(skipped_at), steps, skipped_at[:10], min(kept) if kept else -1,
                                                                                      int(np.median(kept)) if kept else -1), flush=True)
