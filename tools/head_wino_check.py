import os, sys
ROOT = "/root/repo" if os.path.isdir("/root/repo/tools") else os.getcwd()
sys.path[:0] = [ROOT, os.path.join(ROOT, "rock-art-radnet_amd")]
import numpy as np, torch
import bench
from faster_rcnn.config import Config
from radnet_hip import make_engine, synth
from radnet_hip.trainer import TrainStep
np.random.seed(7)
eng = make_engine(Config())
eng.set_weights(synth.synthetic_weights(seed=3))
ts = TrainStep(eng)
batch = bench.make_batch(0, 1, 600, 1000)
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    ts.step(batch)
    l = ts.losses()
    print(k, " ".join("%s=%.6f" % (a, l[a]) for a in ("rpn_cls", "rpn_regr", "det_cls", "det_regr", "det_acc")), flush=True)
