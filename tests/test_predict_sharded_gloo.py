"""Tile-sharded inference over the ranks of a job (inference: tiles round-robin across ranks, per-tile boxes
gathered, merge tail on every rank) on CPU: world_size 2 over gloo.  RADNet.predict with the reference generator's fake models
must return, on EVERY rank, the detections the reference's own single-process RADNet.predict produced (tests/golden/
predict_fake.npz: plain tiling, and tiling + C.include_full_img = 3 work items over 2 ranks).  The device functions the tile
pass calls (resize, rpn_to_roi, NMS) are bound to the oracle's restatements here -- the suite's CPU half has no GPU; the GPU
half runs the same predict path single-process against the same goldens (tests/test_gpu_radnet.py)."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "rock-art-radnet_amd"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    torch.set_num_threads(1)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from faster_rcnn import RADNet as RM, rpn as prpn
    from faster_rcnn.config import Config
    from oracle import glue, resize as oresize
    from test_gpu_radnet import _FakeDet, _FakeRPN
    RM.resize_cubic = lambda img, w, h, **kw: oresize.resize_bicubic_u8(np.ascontiguousarray(img), w, h)
    prpn.rpn_to_roi = lambda Y1, Y2, C, use_regr=True, max_boxes=300, overlap_thresh=0.9: glue.rpn_to_roi(Y1, Y2, C, use_regr, max_boxes, overlap_thresh)
    prpn.non_max_suppression_fast = lambda b, p, overlap_thresh=0.9, max_boxes=300: glue.greedy_nms(np.asarray(b), np.asarray(p), overlap_thresh, max_boxes)
    g = np.load(os.path.join(root, "tests", "golden", "predict_fake.npz"))
    res = {}
    for tag, full in (("plain", False), ("full", True)):
        C = Config(); C.tile_size = 600; C.tile_overlap = 300; C.img_size = 600; C.include_full_img = full
        rpn_model = _FakeRPN(12, 8)
        calls = []
        orig = rpn_model.predict
        rpn_model.predict = lambda X, _o=orig, _c=calls: (_c.append(1), _o(X))[1]
        net = RM.RADNet(C, rpn_model, _FakeDet(7, 2), lambda x: x - np.float32(100.0))
        if world > 1:
            net.set_distributed()
        dets = net.predict([g["img"]])
        res[tag] = (sorted((d["class"], int(d["x1"]), int(d["y1"]), int(d["x2"]), int(d["y2"]), float(d["prob"])) for d in dets), len(calls))
    out[rank] = res
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_return_the_single_process_detections():
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "predict_fake.npz"))
    ref = {tag: sorted((str(c), int(b[0]), int(b[1]), int(b[2]), int(b[3]), float(p)) for c, b, p in zip(g[pre + "classes"], g[pre + "boxes"], g[pre + "probs"]))
           for tag, pre in (("plain", ""), ("full", "full_"))}
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_run, args=(world, _free_port(), out), nprocs=world, join=True)
    for rank in range(world):
        for tag in ("plain", "full"):
            dets, n_passes = out[rank][tag]
            assert dets == ref[tag], (rank, tag, len(dets), len(ref[tag]))           # every rank: the reference's detections
    # the tiles really were shared out: 2 tiles -> 1 network pass per rank; 2 tiles + the full image -> 2 and 1
    assert sorted(out[r]["plain"][1] for r in range(world)) == [1, 1]
    assert sorted(out[r]["full"][1] for r in range(world)) == [1, 2]


def test_single_process_path_unchanged():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_run, args=(1, 0, out), nprocs=1, join=True)          # own process: _run rebinds module functions
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "predict_fake.npz"))
    ref = sorted((str(c), int(b[0]), int(b[1]), int(b[2]), int(b[3]), float(p)) for c, b, p in zip(g["classes"], g["boxes"], g["probs"]))
    assert out[0]["plain"][0] == ref and out[0]["plain"][1] == 2
