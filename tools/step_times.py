import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path[:0] = ["/root/repo", "/root/repo/rock-art-radnet_amd"]
import numpy as np, torch, gc
import bench
from faster_rcnn.config import Config
from radnet_hip import synth
from radnet_hip.engine import FasterRCNNEngine
from radnet_hip.trainer import TrainStep
eng = FasterRCNNEngine(Config()); eng.set_weights(synth.synthetic_weights(seed=3)); ts = TrainStep(eng)
batch = bench.make_batch(0, 1, 600, 1000); np.random.seed(64)
for k in range(40): ts.step(batch, upcoming=[batch] * min(3, 39 - k))
ts.flush(); torch.cuda.synchronize(); gc.collect(); gc.freeze()
K = 30
t = [time.perf_counter()]
for k in range(K):
    ts.step(batch, upcoming=[batch] * min(3, K - 1 - k)); t.append(time.perf_counter())
ts.flush(); torch.cuda.synchronize(); t.append(time.perf_counter())
d = np.diff(t) * 1e3
print("per-call ms:", " ".join("%.2f" % x for x in d[:-1]), "| final drain %.2f" % d[-1])
print("total %.1f ms for %d steps -> %.1f img/s; middle steps mean %.2f ms" % ((t[-1] - t[0]) * 1e3, K, K / (t[-1] - t[0]), d[5:K - 3].mean()))
