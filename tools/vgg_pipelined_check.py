import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path[:0] = ["/root/repo", "/root/repo/rock-art-radnet_amd"]
import numpy as np, torch
from faster_rcnn.config import Config
from radnet_hip import make_engine, synth
from radnet_hip.trainer import TrainStep
C = Config(); C.network = "vgg16"; C.anchor_box_scales = [128, 256, 512]; C.img_size = 600
eng = make_engine(C)
eng.set_weights(synth.synthetic_weights_vgg16(seed=3, n_anchors=eng.A, n_classes=eng.nc))
meta = synth.synthetic_gt(2, n=8, src_w=2000, src_h=1200)
batch = [dict(img=synth.synthetic_panel(1, 600, 1000), bboxes=meta["bboxes"], width=2000, height=1200)]
np.random.seed(64)
ts = TrainStep(eng)
for k in range(16): ts.step(batch, upcoming=[batch] * min(3, 15 - k))
ts.flush(); torch.cuda.synchronize()
t0 = time.perf_counter(); K = 30
for k in range(K): ts.step(batch, upcoming=[batch] * min(3, K - 1 - k))
ts.flush(); torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print("vgg16 pipelined: %.2f ms/step (%.1f img/s), skipped %d, losses %s" % (dt * 1e3, 1 / dt, ts.skipped_head_steps, ts.losses()))
t0 = time.perf_counter()
for k in range(K): ts.step(batch)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print("vgg16 one lane:  %.2f ms/step (%.1f img/s)" % (dt * 1e3, 1 / dt))
