"""Composed C-ABI entry points (include/radnet_hip.h, csrc/program.hip; SURVEY.md 8b "minimum exports") against the
scheduler-driven path on the same engine -- the SAME kernels in the same order, so results must agree bit for bit -- and,
through it, against the oracle (tests/test_gpu_engine.py covers the scheduler path against the oracle):

  radnet_rpn_forward     base program + RPN program                          == engine.base_forward + engine.rpn_forward
  radnet_predict_tile    preprocess .. proposals .. classifier outputs        == the NumPy-facing model calls RADNet makes
  radnet_train_step      one reference iteration, RNG steps as host callbacks == trainer.TrainStep on one lane
  radnet_comm_init / radnet_allreduce_grads   1-rank RCCL communicator: identity all-reduce on the context's stream"""
import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def make(C=None, img_size=300):
    from faster_rcnn.config import Config
    from oracle import dense
    from radnet_hip.engine import FasterRCNNEngine
    C = C or Config()
    C.img_size = img_size
    P = dense.init_params(seed=3)
    eng = FasterRCNNEngine(C)
    eng.set_weights(P)
    return C, P, eng


def sample(i=0):
    from radnet_hip import synth
    meta = synth.synthetic_gt(40 + i, n=6, src_w=1000, src_h=600, smin=60, smax=300)
    return dict(img=synth.synthetic_panel(30 + i, 300, 500), bboxes=meta["bboxes"], width=1000, height=600)


def test_rpn_forward_composed_equals_program_by_program():
    from radnet_hip import native
    C, P, eng = make()
    img = sample()["img"]
    bp = eng.upload_image(img)
    eng.base_forward(bp)
    rp = eng.rpn_forward(bp)
    want_F, want = bp["F"].cpu().numpy().copy(), rp["pred"].cpu().numpy().copy()
    bp["F"].zero_(); rp["pred"].zero_()
    rp2 = native.rpn_forward(eng, bp)
    assert rp2 is rp
    assert np.array_equal(bp["F"].cpu().numpy(), want_F) and np.array_equal(rp["pred"].cpu().numpy(), want)


def test_predict_tile_composed_equals_model_calls():
    from faster_rcnn import models as M
    from faster_rcnn import rpn
    from faster_rcnn.base_models import resnet50
    from faster_rcnn.config import Config
    from oracle import dense
    from radnet_hip import native
    C = Config(); C.img_size = 300
    P = dense.init_params(seed=3)
    m_rpn, m_cls, m_all, m_rpn3, m_det = M.build_models(C, weights=copy.deepcopy(P))
    eng = m_all._s.eng
    img = np.random.RandomState(11).randint(0, 256, (300, 420, 3)).astype(np.uint8)
    R, pc, pr = native.predict_tile(eng, torch.from_numpy(img).cuda(), 40)
    X = resnet50.preprocess(img[:, :, (2, 1, 0)].astype(np.float32)[None])
    Y1, Y2, F = m_rpn3.predict(X)
    R_ref = rpn.rpn_to_roi(Y1, Y2, C, overlap_thresh=0.7)
    assert np.array_equal(R, R_ref)
    rois = R_ref[:40].copy(); rois[:, 2] -= rois[:, 0]; rois[:, 3] -= rois[:, 1]
    pc_ref, pr_ref = m_det.predict([F, rois[None]])
    assert np.array_equal(pc, pc_ref[0]) and np.array_equal(pr, pr_ref[0])


def test_train_step_composed_equals_scheduler_one_lane():
    """Three iterations through radnet_train_step == three TrainStep.step calls (one lane): losses, RNG consumption, RoI
    selection, and every trainable weight."""
    from radnet_hip import native
    from radnet_hip.trainer import TrainStep
    batches = [sample(i) for i in range(3)]
    C, P, eng_a = make()
    np.random.seed(64)
    ts = TrainStep(eng_a)
    ts.capture = []
    la = []
    for s in batches:
        ts.step([s])
        la.append(ts.losses())
    rng_a = int(np.random.randint(0, 2 ** 31 - 1))
    wa = eng_a.get_weights()
    tune = "/tmp/radnet_native_tune.txt"
    eng_a.save_tuning(tune)
    C2, P2, eng_b = make()
    eng_b.load_tuning(tune)                          # same launch shapes -> same summation order
    np.random.seed(64)
    nt = native.NativeTrainStep(eng_b)
    nt.capture = []
    lb = []
    for s in batches:
        lb.append(nt.step(s).losses())
    rng_b = int(np.random.randint(0, 2 ** 31 - 1))
    wb = eng_b.get_weights()
    assert rng_a == rng_b
    for a, b, ca, cb in zip(la, lb, ts.capture, nt.capture):
        assert a["n_head"] == b["n_head"] == 1
        assert ca["sel_kept"] == cb["sel_kept"] and np.array_equal(ca["R"], cb["R"])
        for k in ("rpn_cls", "rpn_regr", "det_cls", "det_regr", "det_acc"):
            assert a[k] == b[k], (k, a[k], b[k])
    for name in wa:
        for k in wa[name]:
            assert np.array_equal(wa[name][k], wb[name][k]), (name, k)     # ordered reductions: the native step is the same arithmetic


def _comm_stats(eng):
    import ctypes as C
    calls, elems = C.c_int64(0), C.c_int64(0)
    eng.ctx.check(eng.lib.radnet_comm_stats(eng.ctx.h, C.byref(calls), C.byref(elems)), "comm_stats")
    return calls.value, elems.value


def test_train_step_keeps_the_collectives_symmetric_on_its_early_exits():
    """Data parallel (world > 1): a rank whose image the labeller drops (n_pos < 0), or whose RoI hook keeps nothing, must
    still issue the SAME sequence of gradient exchanges as a peer that trains on its image -- RPN arena, then head arena --
    or the peers block in ncclAllReduce.  Rehearsed on one GPU: the descriptor says world = 2, the communicator has
    one rank, radnet_comm_stats counts the exchanges per path; zero gradients leave first-step weights where they were."""
    from radnet_hip import native
    C, P, eng = make()
    native.comm_init(eng, 1, 0)
    n_rpn, n_head = eng.rpn_arena.n, eng.head_arena.n
    try:
        nt = native.NativeTrainStep(eng, world=2)
        w0 = eng.get_weights()
        # (1) labeller failure
        nt.force_drop = True
        c0, e0 = _comm_stats(eng)
        nt.step(sample(0))
        c1, e1 = _comm_stats(eng)
        assert nt.losses()["dropped"] == 1 and nt.losses()["n_head"] == 0
        assert (c1 - c0, e1 - e0) == (2, n_rpn + n_head)
        assert eng.rpn_arena.t == 1 and eng.head_arena.t == 1                  # the peers' step counters
        w1 = eng.get_weights()
        for name in w0:
            for k in w0[name]:
                assert np.array_equal(w0[name][k], w1[name][k]), (name, k)      # Adam on zeros with zero moments: no move
        # (2) no RoI kept: the RPN phase trains, the head joins with zeros
        nt.force_drop, nt.force_no_rois = False, True
        np.random.seed(64)
        nt.step(sample(1))
        c2, e2 = _comm_stats(eng)
        assert (c2 - c1, e2 - e1) == (2, n_rpn + n_head)
        assert nt.losses()["n_head"] == 0 and nt.skipped_head_steps == 1
        assert eng.rpn_arena.t == 2 and eng.head_arena.t == 2
        w2 = eng.get_weights()
        assert not np.array_equal(w1["rpn_conv1"]["kernel"], w2["rpn_conv1"]["kernel"])
        assert np.array_equal(w1["res5a_branch2a"]["kernel"], w2["res5a_branch2a"]["kernel"])
        # (3) a full step: the same two exchanges
        nt.force_no_rois = False
        nt.step(sample(2))
        c3, e3 = _comm_stats(eng)
        assert (c3 - c2, e3 - e2) == (2, n_rpn + n_head) and nt.losses()["n_head"] == 1
        # world = 1 takes no part in any exchange
        nt1 = native.NativeTrainStep(eng, world=1)
        nt1.force_drop = True
        nt1.step(sample(3))
        assert _comm_stats(eng) == (c3, e3)
    finally:
        eng.ctx.check(eng.lib.radnet_comm_destroy(eng.ctx.h), "comm_destroy")


def test_native_step_marks_inference_filters_stale():
    """After radnet_train_step moved the head weights, an inference head plan (Winograd-transformed copies of the classifier's
    3x3 filters) must re-transform them before its next pass."""
    from radnet_hip import native
    C, P, eng = make()
    eng._inference_filters_stale = False
    np.random.seed(64)
    nt = native.NativeTrainStep(eng)
    nt.step(sample(0))
    assert nt.losses()["n_head"] == 1 and eng._inference_filters_stale is True


def test_allreduce_grads_one_rank_communicator():
    from radnet_hip import native
    C, P, eng = make()
    native.comm_init(eng, 1, 0)
    g = eng.rpn_arena.g
    g.copy_(torch.arange(g.numel(), dtype=torch.float32, device=g.device) % 1000)
    want = g.cpu().numpy().copy()
    native.allreduce(eng, g)
    torch.cuda.synchronize()
    assert np.array_equal(g.cpu().numpy(), want)
    eng.ctx.check(eng.lib.radnet_comm_destroy(eng.ctx.h), "comm_destroy")
    g.zero_()


def test_pipelined_step_with_native_exchanges_equals_the_step_without_them():
    """Round 4: the data-parallel schedule through the library's own RCCL binding (TrainStep.native_comm: AR#1 in line on the
    main lane, AR#2 one call on the head communicator's stream, head update deferred to the next step's head forward),
    rehearsed with 1-rank communicators -- every exchange an identity, so each loss, each weight and the random stream must
    equal the plain pipelined step's bit for bit; any difference is a missing dependency between the lanes and the
    communicator's stream."""
    import os
    import tempfile
    from radnet_hip import trainer as T
    batches = [[sample(i)] for i in range(5)]
    out = {}
    table = os.path.join(tempfile.mkdtemp(), "shapes.txt")       # both engines on ONE table of launch shapes (another K-split count
    for mode in ("plain", "native"):                             # re-associates the sums: bit equality holds for a given table)
        C, P, eng = make()
        if os.path.exists(table):
            eng.load_tuning(table)
        T.FORCE_COLLECTIVES = mode == "native"
        try:
            ts = T.TrainStep(eng, world_size=1, defer_head_update=True if mode == "native" else None)
            assert ts.native_comm == (mode == "native")
            np.random.seed(64)
            losses = []
            for k, b in enumerate(batches):
                ts.step(b, upcoming=batches[k + 1:k + 4])
                losses.append(ts.losses())
            ts.flush()
            torch.cuda.synchronize()
            out[mode] = (losses, eng.get_weights(), int(np.random.randint(0, 2 ** 31 - 1)))
            if mode == "plain":
                eng.save_tuning(table)
            if mode == "native":
                import ctypes
                calls, elems = ctypes.c_int64(), ctypes.c_int64()
                eng.lib.radnet_comm_stats(ts._main_ctx.h, ctypes.byref(calls), ctypes.byref(elems))
                assert calls.value == len(batches) and elems.value == len(batches) * eng.rpn_arena.n
                eng.lib.radnet_comm_stats(ts._comm_ctx.h, ctypes.byref(calls), ctypes.byref(elems))
                assert calls.value == len(batches) and elems.value == len(batches) * eng.head_arena.n
        finally:
            T.FORCE_COLLECTIVES = False
    assert out["plain"][2] == out["native"][2]
    for a, b in zip(out["plain"][0], out["native"][0]):
        assert a == b, (a, b)
    for name, d in out["plain"][1].items():
        for k, v in d.items():
            assert np.array_equal(v, out["native"][1][name][k]), (name, k)
