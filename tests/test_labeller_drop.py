"""Labeller failures inside the training step (reference: utils.py:789-797 raises KeyError while building the
subsampling probabilities, utils.py:461-465 swallows it inside the generator and the sample never reaches the model).

TrainStep must drop such an image before any optimizer effect, keep going -- in pipelined mode the failure surfaces for the
ANNOUNCED batch in the middle of step(i), whose own head phase must still run -- and, data-parallel, still join every
collective (zeros) so the peers do not hang.  Exercised with the recording stand-in engine of test_dp_deferred (no
kernels), world 1 in-process and world 2 over gloo."""
import contextlib
import os
import types

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from test_dp_deferred import FakeEngine, _free_port

FAIL_WIDTH = 13


class LaneFakeEngine(FakeEngine):
    """FakeEngine + the lane interface, so TrainStep runs its pipelined schedule; a sample whose source width is FAIL_WIDTH
    makes anchor_targets_finish raise the reference's KeyError."""
    n_side_lanes = 1

    def __init__(self, rank, bucketed=False):
        super().__init__(rank, bucketed)
        self.ctx = types.SimpleNamespace(timing_on=False)

    def lane(self, name):
        return contextlib.nullcontext()

    @staticmethod
    def mark():
        return None

    @staticmethod
    def after(ev):
        pass

    def anchor_targets_launch(self, gt, width, height, W, H, slot=0):
        self.k += 1
        return dict(slot=slot, fail=(width == FAIL_WIDTH))

    def anchor_targets_finish(self, tp):
        if tp["fail"]:
            raise KeyError(3)
        return None, None, 0


def _batches(fail_index, n=4):
    out = []
    for i in range(n):
        out.append([dict(img=np.zeros((4, 4, 3), np.uint8), bboxes=[dict({"class": "fg"}, x1=0, x2=2, y1=0, y2=2)],
                         width=FAIL_WIDTH if i == fail_index else 8, height=8)])
    return out


@pytest.mark.parametrize("pipelined", [True, False])
def test_failed_labelling_drops_the_image_and_training_goes_on(pipelined):
    from radnet_hip.trainer import TrainStep
    np.random.seed(64)
    eng = LaneFakeEngine(0) if pipelined else FakeEngine(0)
    if not pipelined:
        eng.anchor_targets_launch = types.MethodType(LaneFakeEngine.anchor_targets_launch, eng)
        eng.anchor_targets_finish = types.MethodType(LaneFakeEngine.anchor_targets_finish, eng)
    ts = TrainStep(eng, world_size=1)
    drops = []
    ts.on_drop = lambda sample, exc: drops.append((sample["width"], type(exc).__name__))
    bs = _batches(fail_index=2)
    n_head_after = []
    for i, b in enumerate(bs):
        ts.step(b, upcoming=bs[i + 1:i + 4] if pipelined else None)
        n_head_after.append(sum(1 for e in eng.log if e[0] == "head_fwd"))
        if i == 2:
            l = ts.losses()
            assert l["dropped"] == 1 and l["n_head"] == 0 and np.isnan(l["rpn_cls"])
    ts.flush()
    assert drops == [(FAIL_WIDTH, "KeyError")] and ts.dropped_images == 1 and ts.skipped_head_steps == 0
    # the dropped batch trained nothing: three RPN updates, three head updates, and every other batch's head phase ran in
    # its own step (batch 1's too, although batch 2's failure surfaced in the middle of that call when pipelined)
    assert sum(1 for e in eng.log if e[0] == "adam_rpn") == 3
    assert sum(1 for e in eng.log if e[0] == "adam_head") == 3
    assert n_head_after == [1, 2, 2, 3]
    assert np.allclose(eng.rpn_arena.p.numpy(), -3.0)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "rock-art-radnet_amd"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from radnet_hip.trainer import TrainStep
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    np.random.seed(64 + rank)
    eng = LaneFakeEngine(rank, bucketed=True)
    ts = TrainStep(eng, world_size=world, defer_head_update=True)
    ts.on_drop = lambda sample, exc: None
    bs = _batches(fail_index=2 if rank == 1 else -1)
    for i, b in enumerate(bs):
        ts.step(b, upcoming=bs[i + 1:i + 4])
    ts.flush()
    out[rank] = (eng.log, eng.rpn_arena.p.numpy().copy(), eng.head_arena.p.numpy().copy(), ts.dropped_images)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_rank_with_a_dropped_image_still_joins_both_exchanges():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)          # a hang here = a missed collective
    (log0, rp0, hp0, d0), (log1, rp1, hp1, d1) = out[0], out[1]
    assert (d0, d1) == (0, 1)
    assert np.array_equal(rp0, rp1) and np.array_equal(hp0, hp1)                           # replicas stay identical
    for log in (log0, log1):
        assert sum(1 for e in log if e[0] == "adam_rpn") == 4 and sum(1 for e in log if e[0] == "adam_head") == 4
    assert sum(1 for e in log1 if e[0] == "head_fwd") == 3 and sum(1 for e in log0 if e[0] == "head_fwd") == 4
    # RPN arena: batches 0, 1, 3 get (1 + 2) / 2 images; batch 2 only rank 0's gradient over the nominal global batch
    assert np.allclose(rp0, -(1.5 * 3 + 0.5))
