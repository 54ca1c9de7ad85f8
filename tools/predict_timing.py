"""Wall time of the predict.py path (RADNet.predict on one 2048x2048 synthetic tile, BASELINE cfg 3) and where the host
spends it.  usage: python tools/predict_timing.py [img_size]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rock-art-radnet_amd")]

from faster_rcnn import models  # noqa: E402
from faster_rcnn.RADNet import RADNet  # noqa: E402
from faster_rcnn.base_models import resnet50  # noqa: E402
from faster_rcnn.config import Config  # noqa: E402
from radnet_hip import synth  # noqa: E402


def main():
    C = Config()
    if len(sys.argv) > 1:
        C.img_size = int(sys.argv[1])
    m_rpn, m_cls, m_all, m_rpn3, m_det = models.build_models(C)
    m_all._s.eng.set_weights(synth.synthetic_weights(seed=3))
    net = RADNet(C, m_rpn3, m_det, resnet50.preprocess)
    net.bbox_threshold = 0.0          # synthetic weights are never confident: keep every non-background RoI in the tail
    tile = np.random.RandomState(4).randint(0, 256, (2048, 2048, 3)).astype(np.uint8)
    for _ in range(3):
        dets = net.predict([tile])
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = 10
    for _ in range(n):
        dets = net.predict([tile])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    print("RADNet.predict, 2048x2048 tile -> short side %d: %.1f ms per tile (%d detections)" % (C.img_size, dt * 1e3, len(dets)))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        net.predict([tile])
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(18)


if __name__ == "__main__":
    main()
