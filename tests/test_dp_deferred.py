"""Deferred head update of the data-parallel train step, world_size 2 over gloo on CPU.

trainer.TrainStep starts the all-reduce of the head gradients asynchronously after the head backward and applies
Adam #2 just before the NEXT step's head forward (the first reader of head weights), so the 60 MB exchange overlaps the
next image's labelling / base forward / RPN phases.  The step logic is exercised here with a recording stand-in for the
HIP engine (no kernels: the engine's numerics are covered by the GPU tests): what is checked is ORDER (every head
forward sees all earlier head updates, none is lost, flush() applies the last one) and that the update uses the SUM over
ranks scaled by 1 / global batch."""
import os
import socket
import types

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Arena:
    def __init__(self, n):
        self.p = torch.zeros(n, dtype=torch.float64)
        self.g = torch.zeros(n, dtype=torch.float64)
        self.n = n


class FakeEngine:
    """Implements exactly what TrainStep.step touches; gradients are rank- and step-dependent constants and 'Adam' is
    p -= grad_scale * g followed by clearing g (the zero_grad contract of the real engine)."""

    def __init__(self, rank, bucketed=False, skip_step=None):
        self.rank = rank
        self.skip_step = skip_step     # this rank finds no RoI to train on in that step (calc_iou -> None in the reference)
        self.bucketed = bucketed
        self.head_bias_off = 6         # arena of 8: three "blocks" of 2 + a tail of 2 (biases, dense heads)
        self.dev = "cpu"
        self.bg = 1
        self.C = types.SimpleNamespace(class_mapping={"fg": 0, "bg": 1}, img_size=600, n_rois=4)
        self.rpn_arena, self.head_arena = _Arena(8), _Arena(8)
        self.log = []
        self.k = -1            # step index, advanced by the first call of a step

    def head_exchange_slices(self):
        return [(4, 6), (2, 4), (0, 2)] if self.bucketed else None

    def upload_gt(self, boxes, isbg, cls):
        return dict(g=len(boxes))

    def anchor_targets_launch(self, gt, width, height, W, H, slot=0):
        self.k += 1            # one image per step here (slot alternates between the two buffer sets)
        return dict(slot=slot)

    def upload_image(self, img, slot=0):
        return dict(fh=2, fw=2, F=None, slot=slot)

    def base_forward(self, bp):
        self.log.append(("base_fwd", self.k))

    def rpn_forward(self, bp):
        return dict(fwd=[], bwd=[], pred=None)

    def anchor_targets_finish(self, tp):
        return None, None, 0

    @staticmethod
    def set_accumulate(ops, flag, prezeroed=False):
        pass

    def rpn_backward(self, rp, ycls, yregr, loss_out=None):
        self.rpn_arena.g += float(self.rank + 1)

    def adam(self, arena, grad_scale=1.0, zero_grad=True):
        name = "head" if arena is self.head_arena else "rpn"
        self.log.append(("adam_" + name, self.k, float(arena.g[0]) * grad_scale))
        arena.p -= grad_scale * arena.g
        arena.g.zero_()

    def refresh_head_shift(self):
        self.log.append(("refresh", self.k))

    def _run(self, ops, overlap=False):
        pass

    def proposals(self, rp, overlap_thresh=0.7, max_boxes=300):
        return None, None

    def roi_targets_launch(self, R, Rn, gt, width, height, rw, rh, slot=0):
        return dict()

    def roi_targets_finish(self, P):
        if self.skip_step is not None and self.k == self.skip_step:
            return P, np.zeros(0, dtype=np.int32), 0
        return P, np.array([0, 1, 0, 1, 1], dtype=np.int32), 5

    def _plan_head(self, R, fh, fw, F):
        return dict(bwd=[], bwd_parts=[([], (4, 6)), ([], (2, 4)), ([], (0, 2))])

    def pack_roi_batch(self, P, sel, hp):
        pass

    def head_forward(self, hp, training=False, loss_out=None, group_live=None):
        self.log.append(("head_fwd", self.k, float(self.head_arena.p[0])))

    def head_backward(self, hp, accumulate=True, loss_out=None, on_part=None):
        self.head_arena.g += float((self.rank + 1) * (self.k + 1))
        if on_part is not None:            # the real engine reports each block's finished kernel-gradient slice
            self.log.append(("bucketed", self.k))
            for _, (lo, hi) in hp["bwd_parts"]:
                on_part(lo, hi)


def _worker(rank, world, port, out, defer, bucketed=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "rock-art-radnet_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from radnet_hip.trainer import TrainStep
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    np.random.seed(64 + rank)
    eng = FakeEngine(rank, bucketed)
    ts = TrainStep(eng, world_size=world, defer_head_update=defer)
    batch = [dict(img=np.zeros((4, 4, 3), np.uint8), bboxes=[dict({"class": "fg"}, x1=0, x2=2, y1=0, y2=2)], width=8, height=8)]
    for _ in range(3):
        ts.step(batch)
    ts.flush()
    out[rank] = (eng.log, eng.head_arena.p.numpy().copy(), eng.rpn_arena.p.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("defer,bucketed", [(True, False), (True, True), (False, False)])
def test_deferred_head_update_order_and_value(defer, bucketed):
    """bucketed: the engine reports each block's finished gradient slice during the backward and the trainer exchanges the
    slices one by one (+ the arena's tail at the end): every element must still be reduced exactly once."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out, defer, bucketed), nprocs=world, join=True)
    (log0, hp0, rp0), (log1, hp1, rp1) = out[0], out[1]
    assert any(e[0] == "bucketed" for e in log0) == bucketed
    assert np.array_equal(hp0, hp1) and np.array_equal(rp0, rp1)          # replicas stay identical
    # head update of step k: sum over ranks of (rank+1)(k+1) = 3(k+1), scaled by 1/global batch (2 images)
    upd = [1.5 * (k + 1) for k in range(3)]
    assert np.allclose(hp0, -sum(upd)) and np.allclose(rp0, -3 * 1.5)
    for log in (log0, log1):
        heads = [e for e in log if e[0] == "adam_head"]
        assert [round(e[2], 12) for e in heads] == upd                    # each step's update applied exactly once, in order
        fwd = [e for e in log if e[0] == "head_fwd"]
        # every head forward reads weights that already contain ALL earlier head updates
        assert np.allclose([e[2] for e in fwd], [0.0, -upd[0], -upd[0] - upd[1]])
        pos = {("adam_head", k): i for i, e in enumerate(log) if e[0] == "adam_head" for k in [heads.index(e)]}
        for k in range(1, 3):
            i_upd = pos[("adam_head", k - 1)]
            i_fwd = log.index(fwd[k])
            i_rpn = [i for i, e in enumerate(log) if e[0] == "adam_rpn" and e[1] == k][0]
            assert i_upd < i_fwd
            # deferred: the update of step k-1 lands AFTER step k's RPN phase (that is the overlap window);
            # immediate: before step k even starts
            assert (i_upd > i_rpn) == defer
        assert any(e[0] == "refresh" for e in log)


def _worker_skip(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "rock-art-radnet_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from radnet_hip.trainer import TrainStep
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    np.random.seed(64 + rank)
    eng = FakeEngine(rank, bucketed=True, skip_step=1 if rank == 1 else None)
    ts = TrainStep(eng, world_size=world, defer_head_update=True)
    batch = [dict(img=np.zeros((4, 4, 3), np.uint8), bboxes=[dict({"class": "fg"}, x1=0, x2=2, y1=0, y2=2)], width=8, height=8)]
    for _ in range(3):
        ts.step(batch)
    ts.flush()
    out[rank] = (eng.head_arena.p.numpy().copy(), ts.skipped_head_steps)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_bucketed_exchange_when_one_rank_skips_its_head_step():
    """Rank 1 has nothing to train the classifier on in step 1: it must still issue the SAME sequence of slice exchanges
    (zeros from its side) or the job hangs; the update of that step is then rank 0's gradient alone over the global batch."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_skip, args=(world, _free_port(), out), nprocs=world, join=True)
    (hp0, sk0), (hp1, sk1) = out[0], out[1]
    assert (sk0, sk1) == (0, 1) and np.array_equal(hp0, hp1)
    # steps 0 and 2: (1 + 2) * (k + 1) / 2 images; step 1: rank 0 only -> 1 * 2 / 2
    assert np.allclose(hp0, -(1.5 * 1 + 1.0 + 1.5 * 3))


class FakeContEngine(FakeEngine):
    """What trainer_cont.ContTrainStep touches beyond FakeEngine: the shared stage-3/4 arena and its two optimizers."""

    def __init__(self, rank):
        super().__init__(rank)
        self.s34_arena = _Arena(8)

    def stem_forward(self, bp):
        self.log.append(("stem_fwd", self.k))

    def s34_forward(self, bp):
        self.log.append(("s34_fwd", self.k, float(self.s34_arena.p[0])))

    def upload_image(self, img, slot=0):
        return dict(fh=2, fw=2, F=None, slot=slot, bwd34=[])

    def s34_backward(self, bp):
        self.s34_arena.g += float(10 * (self.rank + 1))

    def adam_s34(self, which, grad_scale=1.0):
        self.log.append(("adam_s34_%d" % which, self.k, float(self.s34_arena.g[0]) * grad_scale))
        self.s34_arena.p -= grad_scale * self.s34_arena.g
        self.s34_arena.g.zero_()

    def roi_targets(self, R, Rn, gt, width, height, rw, rh):
        return self.roi_targets_finish(self.roi_targets_launch(R, Rn, gt, width, height, rw, rh))


def _cont_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "rock-art-radnet_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from radnet_hip.trainer_cont import ContTrainStep
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    np.random.seed(64 + rank)
    eng = FakeContEngine(rank)
    ts = ContTrainStep(eng, world_size=world)
    batch = [dict(img=np.zeros((4, 4, 3), np.uint8), bboxes=[dict({"class": "fg"}, x1=0, x2=2, y1=0, y2=2)], width=8, height=8)]
    for _ in range(2):
        ts.step(batch)
    out[rank] = (eng.log, eng.s34_arena.p.numpy().copy(), eng.head_arena.p.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_cont_mode_exchanges_the_shared_arena_before_both_optimizers():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_cont_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    (log0, s0, h0), (log1, s1, h1) = out[0], out[1]
    assert np.array_equal(s0, s1) and np.array_equal(h0, h1)
    # per step and per optimizer the shared arena receives (10 + 20) summed over ranks, scaled by 1/2: 15; two optimizers, two steps
    assert np.allclose(s0, -4 * 15.0)
    for log in (log0, log1):
        upd = [e for e in log if e[0].startswith("adam_s34")]
        assert [e[0] for e in upd] == ["adam_s34_0", "adam_s34_1"] * 2 and all(abs(e[2] - 15.0) < 1e-12 for e in upd)
        # stages 3-4 run twice per step: before the RPN phase and again after Adam #1 moved them
        fwd = [e for e in log if e[0] == "s34_fwd"]
        assert np.allclose([e[2] for e in fwd], [0.0, -15.0, -30.0, -45.0])
        assert sum(1 for e in log if e[0] == "stem_fwd") == 2
